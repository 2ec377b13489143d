// Dev microbenchmark (round 3): cycles per wave64 instruction per SIMD on gfx950, by instruction class, written in
// inline asm so that nothing can be folded (profiles/r2/valu_rate.txt had rows the compiler had optimised away).
// Each kernel runs NIT trips of a block of 32 instructions of ONE class on 8 independent registers; the grid puts
// W = 1, 2, 4, 8 waves on every SIMD (256 CUs x 4 SIMDs); time = s_memtime (shader clock) of the wave, averaged.
//   cycles per wave-instruction per SIMD = wave_cycles / (NIT * 32 * W)   (what the pipe sustains with W waves issuing)
// Memory classes (LDS, L1-resident global loads) are counted per CU: cycles_per_CU = wave_cycles / (NIT * ninstr * 4 * W) ... printed as both.
// Build: hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
typedef __attribute__((ext_vector_type(4))) float f4;

#define R4(x) x x x x
// one instruction template applied to 8 register sets
#define I8(T) T(0, 1) T(1, 2) T(2, 3) T(3, 4) T(4, 5) T(5, 6) T(6, 7) T(7, 0)

#define VREGS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
              "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7])
#define INS   "v"(c0), "v"(c1), "s"(s0), "s"(s1), "s"(m64)
// operand numbering: %0..%7 = a[], %8..%15 = u[], %16 = c0, %17 = c1, %18 = s0, %19 = s1, %20 = m64 (sgpr pair)

#define T_FMA(i, j)      "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define T_MUL(i, j)      "v_mul_f32 %" #i ", %" #i ", %16\n"
#define T_FMAC_S(i, j)   "v_fmac_f32 %" #i ", %18, %16\n"
#define T_FMA_S(i, j)    "v_fma_f32 %" #i ", %18, %16, %" #i "\n"
#define T_RCP(i, j)      "v_rcp_f32 %" #i ", %" #i "\n"
#define T_FLOOR(i, j)    "v_floor_f32 %" #i ", %" #i "\n"
#define T_FRACT(i, j)    "v_fract_f32 %" #i ", %" #i "\n"
#define T_CVTU(i, j)     "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define T_MED3(i, j)     "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define T_MIN(i, j)      "v_min_f32 %" #i ", %" #i ", %16\n"
#define T_DPPADD(i, j)   "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_DPPMOV(i, j)   "v_mov_b32_dpp %" #i ", %" #j " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_DPPROW(i, j)   "v_mov_b32_dpp %" #i ", %" #j " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_CMPDPP(i, j)   "v_xor_b32_dpp %" #i ", %" #j ", %" #j " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define T_CMPVCC(i, j)   "v_cmp_lt_f32 vcc, %" #i ", %" #j "\n"
#define T_CMPSG(i, j)    "v_cmp_lt_f32_e64 s[30:31], %" #i ", %" #j "\n"
#define T_CNDM(i, j)     "v_cndmask_b32_e64 %" #i ", %" #i ", %" #j ", %20\n"
#define T_DOT2(i, j)     "v_dot2c_f32_bf16 %" #i ", %" #i ", %" #j "\n"
#define T_MBLO(i, j)     "v_mbcnt_lo_u32_b32 %" #i ", %18, %" #i "\n"
#define T_MBHI(i, j)     "v_mbcnt_hi_u32_b32 %" #i ", %19, %" #i "\n"
#define T_RDLANE(i, j)   "v_readlane_b32 s30, %" #i ", 3\n"
#define T_RDFIRST(i, j)  "v_readfirstlane_b32 s30, %" #i "\n"
#define T_MAD24(i, j)    "v_mad_u32_u24 %" #i ", %" #i ", %16, %17\n"
#define T_LSHLADD(i, j)  "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n"
#define T_LSHL(i, j)     "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define T_ADDU(i, j)     "v_add_u32 %" #i ", %" #i ", %16\n"
#define T_SUBF(i, j)     "v_sub_f32 %" #i ", %" #i ", %" #j "\n"
#define T_SAND(i, j)     "s_and_b64 s[30:31], %20, %20\n"
#define T_SBCNT(i, j)    "s_bcnt1_i32_b64 s30, %20\n"
#define T_SBREV(i, j)    "s_brev_b64 s[30:31], %20\n"
#define T_SADDC(i, j)    "s_add_u32 s30, %18, %19\ns_addc_u32 s31, %19, %18\n"
#define T_SMOVX(i, j)    "s_mov_b64 s[32:33], exec\ns_mov_b64 exec, %20\ns_mov_b64 exec, s[32:33]\n"
#define T_MIXVS(i, j)    "v_fma_f32 %" #i ", %" #i ", %16, %17\ns_and_b64 s[30:31], %20, %20\n"
#define T_MIXRCP(i, j)   "v_fma_f32 %" #i ", %" #i ", %16, %17\nv_fma_f32 %" #j ", %" #j ", %16, %17\nv_fma_f32 %" #i ", %" #i ", %16, %17\nv_rcp_f32 %" #i ", %" #i "\n"

template <int OP>
__global__ __launch_bounds__(256) void k_valu(u64* out, int nit, float cf, unsigned s0i, unsigned s1i) {
    float a[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; u[i] = threadIdx.x * 977u + i * 31u; }
    float c0 = cf, c1 = cf * 0.5f;
    if (OP >= 100) { c0 = __uint_as_float(3u); c1 = __uint_as_float(5u); }
    unsigned s0 = s0i, s1 = s1i;
    u64 m64 = ((u64)s1i << 32) | s0i;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
#define BODY(T) asm volatile(R4(I8(T)) : VREGS : INS : "vcc", "scc", "s30", "s31", "s32", "s33");
        if (OP == 0) BODY(T_FMA)
        if (OP == 1) BODY(T_MUL)
        if (OP == 2) BODY(T_FMAC_S)
        if (OP == 3) BODY(T_FMA_S)
        if (OP == 4) BODY(T_RCP)
        if (OP == 5) BODY(T_FLOOR)
        if (OP == 6) BODY(T_FRACT)
        if (OP == 7) BODY(T_CVTU)
        if (OP == 8) BODY(T_MED3)
        if (OP == 9) BODY(T_MIN)
        if (OP == 10) BODY(T_DPPADD)
        if (OP == 11) BODY(T_DPPMOV)
        if (OP == 12) BODY(T_DPPROW)
        if (OP == 13) BODY(T_CMPDPP)
        if (OP == 14) BODY(T_CMPVCC)
        if (OP == 15) BODY(T_CMPSG)
        if (OP == 16) BODY(T_CNDM)
        if (OP == 17) BODY(T_DOT2)
        if (OP == 18) BODY(T_RDLANE)
        if (OP == 19) BODY(T_RDFIRST)
        if (OP == 20) BODY(T_SUBF)
        if (OP == 21) BODY(T_SAND)
        if (OP == 22) BODY(T_SBCNT)
        if (OP == 25) BODY(T_SBREV)
        if (OP == 26) BODY(T_SADDC)
        if (OP == 27) BODY(T_SMOVX)
        if (OP == 23) BODY(T_MIXVS)
        if (OP == 24) BODY(T_MIXRCP)
        if (OP == 100) BODY(T_MBLO)
        if (OP == 101) BODY(T_MBHI)
        if (OP == 102) BODY(T_MAD24)
        if (OP == 103) BODY(T_LSHLADD)
        if (OP == 104) BODY(T_LSHL)
        if (OP == 105) BODY(T_ADDU)
    }
    const u64 t1 = __builtin_readcyclecounter();
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i] + (float)u[i];
    if (r == 12345.678f) out[0] = 1;                       // keep the registers live
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- LDS classes -----------------------------------------------------------------------------------------------------------
// MODE 0: ds_read_b128, all lanes the same address (broadcast); 1: ds_read_b128 lane*16; 2: ds_read_b128 random 16 slots (run-like: lane/16)
// 3: ds_write_b128 all lanes lane*16; 4: ds_write_b128 exec = 4 lanes; 5: ds_write_b128 exec = 16 lanes (one per quad)
// 6: ds_write_b32 lane*4; 7: ds_write_b32 exec = 8 lanes; 8: ds_read_b32 lane*4; 9: ds_read2_b32; 10: ds_read_b64 lane*8
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(u64* out, int nit) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[4 * 4096];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned base = (unsigned)(size_t)(sm) + wv * 4096;       // LDS byte address (low 32 bits of the flat shared pointer are the offset)
    base = (unsigned)(wv * 4096);
    unsigned addr;
    if (MODE == 0) addr = base;
    else if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) addr = base + lane * 16;
    else if (MODE == 2) addr = base + (lane / 16) * 48;
    else if (MODE == 10) addr = base + lane * 8;
    else addr = base + lane * 4;
    u4 d[8]; for (int i = 0; i < 8; ++i) d[i] = u4{(unsigned)lane, 1u, 2u, 3u};
    u64 em = MODE == 4 ? 0x0001000100010001ull : MODE == 5 ? 0x1111111111111111ull : MODE == 7 ? 0x0101010101010101ull : ~0ull;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
        if (MODE <= 2) {
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:64\n"
                "ds_read_b128 %4, %8 offset:128\n ds_read_b128 %5, %8 offset:256\n ds_read_b128 %6, %8 offset:512\n ds_read_b128 %7, %8 offset:1024\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]) : "v"(addr) : "memory");
        } else if (MODE <= 5) {
            asm volatile(
                "s_mov_b64 s[30:31], exec\n s_mov_b64 exec, %9\n"
                "ds_write_b128 %8, %0\n ds_write_b128 %8, %1 offset:1024\n ds_write_b128 %8, %2 offset:2048\n ds_write_b128 %8, %3 offset:3072\n"
                "ds_write_b128 %8, %4\n ds_write_b128 %8, %5 offset:1024\n ds_write_b128 %8, %6 offset:2048\n ds_write_b128 %8, %7 offset:3072\n"
                "s_mov_b64 exec, s[30:31]\n s_waitcnt lgkmcnt(0)\n"
                :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(addr), "s"(em) : "memory", "s30", "s31");
        } else if (MODE <= 7) {
            asm volatile(
                "s_mov_b64 s[30:31], exec\n s_mov_b64 exec, %9\n"
                "ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:256\n ds_write_b32 %8, %2 offset:512\n ds_write_b32 %8, %3 offset:768\n"
                "ds_write_b32 %8, %4\n ds_write_b32 %8, %5 offset:256\n ds_write_b32 %8, %6 offset:512\n ds_write_b32 %8, %7 offset:768\n"
                "s_mov_b64 exec, s[30:31]\n s_waitcnt lgkmcnt(0)\n"
                :: "v"(d[0].x), "v"(d[1].x), "v"(d[2].x), "v"(d[3].x), "v"(d[4].x), "v"(d[5].x), "v"(d[6].x), "v"(d[7].x), "v"(addr), "s"(em) : "memory", "s30", "s31");
        } else if (MODE == 8) {
            asm volatile(
                "ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=v"(d[0].x), "=v"(d[1].x), "=v"(d[2].x), "=v"(d[3].x), "=v"(d[4].x), "=v"(d[5].x), "=v"(d[6].x), "=v"(d[7].x) : "v"(addr) : "memory");
        } else if (MODE == 9) {
            asm volatile(
                "ds_read2_b32 %0, %8 offset0:0 offset1:1\n ds_read2_b32 %1, %8 offset0:8 offset1:9\n ds_read2_b32 %2, %8 offset0:16 offset1:17\n ds_read2_b32 %3, %8 offset0:24 offset1:25\n"
                "ds_read2_b32 %4, %8 offset0:64 offset1:65\n ds_read2_b32 %5, %8 offset0:72 offset1:73\n ds_read2_b32 %6, %8 offset0:80 offset1:81\n ds_read2_b32 %7, %8 offset0:88 offset1:89\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=v"(*(u64*)&d[0]), "=v"(*(u64*)&d[1]), "=v"(*(u64*)&d[2]), "=v"(*(u64*)&d[3]), "=v"(*(u64*)&d[4]), "=v"(*(u64*)&d[5]), "=v"(*(u64*)&d[6]), "=v"(*(u64*)&d[7]) : "v"(addr) : "memory");
        } else {
            asm volatile(
                "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n"
                "s_waitcnt lgkmcnt(0)\n"
                : "=v"(*(u64*)&d[0]), "=v"(*(u64*)&d[1]), "=v"(*(u64*)&d[2]), "=v"(*(u64*)&d[3]), "=v"(*(u64*)&d[4]), "=v"(*(u64*)&d[5]), "=v"(*(u64*)&d[6]), "=v"(*(u64*)&d[7]) : "v"(addr) : "memory");
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    unsigned r = 0; for (int i = 0; i < 8; ++i) r += d[i].x + d[i].w;
    if (r == 0x12345679u) out[0] = sm[r & 4095];
    if (lane == 0) out[1 + blockIdx.x * 4 + wv] = t1 - t0;
}

// ---- L1-resident global loads: what one dwordx4 wave-load costs the CU's vector-memory pipe, by lane pattern -------------------------
// PAT 0: every lane the same 16 B           1: lane*16 (8 full lines)            2: units: 16 lines, 4 lanes x 32 B each (+16 on the second load)
//     3: 4 runs of 16 lanes, one address per run (the (mu,sigma) taps today)     4: exec = one lane per quad (16 lanes), 16 different lines
//     5: exec = lanes 0..15, 16 different lines   6: exec = 4 lanes, 4 lines     7: MFMA A layout: line = lane % 16, chunk = lane / 16
//     8: lanes 0..15 x 32 B contiguous pairs (exec 16 lanes, quad-form gmm window)  9: 2 lanes x 64 B per unit: 32 lines
//    10: dword loads, lane*4                 11: dwordx2 loads lane*8
template <int PAT>
__global__ __launch_bounds__(256) void k_vmem(u64* out, int nit, const unsigned char* gbuf) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned char* base = gbuf + (size_t)(blockIdx.x & 255) * 65536;                  // 16 KB per wave (wave part in the lane offset)
    unsigned off;
    u64 em = ~0ull;
    switch (PAT) {
        case 0: off = 0; break;
        case 1: off = lane * 16; break;
        case 2: off = (lane / 4) * 384 + (lane & 3) * 32; break;
        case 3: off = (lane / 16) * 640; break;
        case 4: off = (lane / 4) * 384; em = 0x1111111111111111ull; break;
        case 5: off = lane * 384; em = 0xffffull; break;
        case 6: off = lane * 96; em = 0x0001000100010001ull; break;
        case 7: off = (lane & 15) * 384 + (lane >> 4) * 16; break;
        case 8: off = lane * 32; em = 0xffffull; break;
        case 9: off = (lane / 2) * 256 + (lane & 1) * 64; break;
        case 10: off = lane * 4; break;
        default: off = lane * 8; break;
    }
    off += wv * 16384;
    u4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = u4{0, 0, 0, 0};
    const unsigned char* p = base;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
        if (PAT == 10) {
            asm volatile(
                "s_mov_b64 s[30:31], exec\n s_mov_b64 exec, %10\n"
                "global_load_dword %0, %8, %9\n global_load_dword %1, %8, %9 offset:256\n global_load_dword %2, %8, %9 offset:512\n global_load_dword %3, %8, %9 offset:768\n"
                "global_load_dword %4, %8, %9 offset:1024\n global_load_dword %5, %8, %9 offset:1280\n global_load_dword %6, %8, %9 offset:1536\n global_load_dword %7, %8, %9 offset:1792\n"
                "s_mov_b64 exec, s[30:31]\n s_waitcnt vmcnt(0)\n"
                : "=v"(d[0].x), "=v"(d[1].x), "=v"(d[2].x), "=v"(d[3].x), "=v"(d[4].x), "=v"(d[5].x), "=v"(d[6].x), "=v"(d[7].x) : "v"(off), "s"(p), "s"(em) : "memory", "s30", "s31");
        } else if (PAT == 11) {
            asm volatile(
                "s_mov_b64 s[30:31], exec\n s_mov_b64 exec, %10\n"
                "global_load_dwordx2 %0, %8, %9\n global_load_dwordx2 %1, %8, %9 offset:512\n global_load_dwordx2 %2, %8, %9 offset:1024\n global_load_dwordx2 %3, %8, %9 offset:1536\n"
                "global_load_dwordx2 %4, %8, %9 offset:2048\n global_load_dwordx2 %5, %8, %9 offset:2560\n global_load_dwordx2 %6, %8, %9 offset:3072\n global_load_dwordx2 %7, %8, %9 offset:3584\n"
                "s_mov_b64 exec, s[30:31]\n s_waitcnt vmcnt(0)\n"
                : "=v"(*(u64*)&d[0]), "=v"(*(u64*)&d[1]), "=v"(*(u64*)&d[2]), "=v"(*(u64*)&d[3]), "=v"(*(u64*)&d[4]), "=v"(*(u64*)&d[5]), "=v"(*(u64*)&d[6]), "=v"(*(u64*)&d[7]) : "v"(off), "s"(p), "s"(em) : "memory", "s30", "s31");
        } else {
            // 8 loads at small immediate offsets inside the wave's 16 KB (all L1-resident after the first trip)
            asm volatile(
                "s_mov_b64 s[30:31], exec\n s_mov_b64 exec, %10\n"
                "global_load_dwordx4 %0, %8, %9\n global_load_dwordx4 %1, %8, %9 offset:16\n global_load_dwordx4 %2, %8, %9 offset:128\n global_load_dwordx4 %3, %8, %9 offset:144\n"
                "global_load_dwordx4 %4, %8, %9 offset:256\n global_load_dwordx4 %5, %8, %9 offset:272\n global_load_dwordx4 %6, %8, %9 offset:1024\n global_load_dwordx4 %7, %8, %9 offset:1040\n"
                "s_mov_b64 exec, s[30:31]\n s_waitcnt vmcnt(0)\n"
                : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]) : "v"(off), "s"(p), "s"(em) : "memory", "s30", "s31");
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    unsigned r = 0; for (int i = 0; i < 8; ++i) r += d[i].x + d[i].w;
    if (r == 0x12345679u) out[0] = r;
    if (lane == 0) out[1 + blockIdx.x * 4 + wv] = t1 - t0;
}

// ---- matrix pipe next to the vector pipe: MFMA 16x16x32 bf16 alone, and interleaved 1:4 with v_fma -----------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
template <int MODE>
__global__ __launch_bounds__(256) void k_mfma(u64* out, int nit) {
    f4 acc[4]; for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    u4 av = u4{threadIdx.x, 2, 3, 4}, bv = u4{5, 6, 7, threadIdx.x};
    float a[8]; for (int i = 0; i < 8; ++i) a[i] = 1.0f + i;
    float c0 = 1.0001f, c1 = 0.5f;
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nit; ++it) {
        if (MODE == 0) {
            asm volatile(R4("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n")
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(av), "v"(bv));
        } else {
#define MF(i) "v_mfma_f32_16x16x32_bf16 %" #i ", %12, %13, %" #i "\n"
#define FM(i) "v_fma_f32 %" #i ", %" #i ", %14, %15\n"
            asm volatile(R4(MF(0) FM(4) FM(5) FM(6) FM(7) MF(1) FM(8) FM(9) FM(10) FM(11) MF(2) FM(4) FM(5) FM(6) FM(7) MF(3) FM(8) FM(9) FM(10) FM(11))
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                         : "v"(av), "v"(bv), "v"(c0), "v"(c1));
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    float r = 0; for (int i = 0; i < 4; ++i) r += acc[i].x + acc[i].w; for (int i = 0; i < 8; ++i) r += a[i];
    if (r == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static u64* d_out; static u64* h_out; static unsigned char* d_buf;
static const int WS[4] = {1, 2, 4, 8};

template <typename F> static void run(const char* name, int ninstr_per_trip, int nit, F launch) {
    printf("%-46s", name);
    for (int wi = 0; wi < 4; ++wi) {
        const int W = WS[wi], blocks = 256 * W;
        hipMemset(d_out, 0, (1 + blocks * 4) * sizeof(u64));
        launch(blocks, nit);                                     // warm
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); launch(blocks, nit); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h_out, d_out, (1 + blocks * 4) * sizeof(u64), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h_out[1 + i];
        const double wave_cyc = s / (blocks * 4);
        const double per_simd = wave_cyc / ((double)nit * ninstr_per_trip * W);
        // s_memtime ticks at a fixed 100 MHz-class reference on some parts: print the wall-clock figure too (2.4 GHz nominal)
        const double wall_per_simd = ms * 1e-3 * 2.4e9 / ((double)nit * ninstr_per_trip * W);
        printf("  W=%d %6.2f (wall %6.2f)", W, per_simd, wall_per_simd);
    }
    printf("\n");
}

#define RUN_VALU(OP, NAME, N) run(NAME, N, 2000, [](int blocks, int nit) { hipLaunchKernelGGL(k_valu<OP>, dim3(blocks), dim3(256), 0, 0, d_out, nit, 1.0001f, 0x0f0f3355u, 0x00ff1248u); })
#define RUN_LDS(MODE, NAME) run(NAME, 8, 2000, [](int blocks, int nit) { hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, nit); })
#define RUN_VMEM(PAT, NAME) run(NAME, 8, 1000, [](int blocks, int nit) { hipLaunchKernelGGL(k_vmem<PAT>, dim3(blocks), dim3(256), 0, 0, d_out, nit, d_buf); })

int main() {
    hipMalloc(&d_out, (1 + 256 * 8 * 4) * sizeof(u64)); h_out = (u64*)malloc((1 + 256 * 8 * 4) * sizeof(u64));
    hipMalloc(&d_buf, (size_t)1024 * 16384 + 65536); hipMemset(d_buf, 0, (size_t)1024 * 16384 + 65536);
    printf("# cycles per wave-instruction per SIMD (s_memtime wave cycles / instructions / W) and the same from wall time at 2.4 GHz; W waves per SIMD\n");
    RUN_VALU(0, "v_fma_f32", 32); RUN_VALU(1, "v_mul_f32", 32); RUN_VALU(2, "v_fmac_f32 v, s, v", 32); RUN_VALU(3, "v_fma_f32 v, s, v, v (VOP3 + sgpr)", 32);
    RUN_VALU(20, "v_sub_f32", 32);
    RUN_VALU(4, "v_rcp_f32", 32); RUN_VALU(5, "v_floor_f32", 32); RUN_VALU(6, "v_fract_f32", 32); RUN_VALU(7, "v_cvt_u32_f32", 32);
    RUN_VALU(8, "v_med3_f32", 32); RUN_VALU(9, "v_min_f32", 32);
    RUN_VALU(10, "v_add_f32_dpp quad_perm", 32); RUN_VALU(11, "v_mov_b32_dpp wave_shr:1", 32); RUN_VALU(12, "v_mov_b32_dpp row_shr:1", 32);
    RUN_VALU(13, "v_xor_b32_dpp wave_shr:1", 32);
    RUN_VALU(14, "v_cmp_lt_f32 vcc", 32); RUN_VALU(15, "v_cmp_lt_f32_e64 sgpr", 32); RUN_VALU(16, "v_cndmask_b32_e64 (sgpr mask)", 32);
    RUN_VALU(17, "v_dot2c_f32_bf16", 32);
    RUN_VALU(100, "v_mbcnt_lo_u32_b32", 32); RUN_VALU(101, "v_mbcnt_hi_u32_b32", 32);
    RUN_VALU(18, "v_readlane_b32", 32); RUN_VALU(19, "v_readfirstlane_b32", 32);
    RUN_VALU(102, "v_mad_u32_u24", 32); RUN_VALU(103, "v_lshl_add_u32", 32); RUN_VALU(104, "v_lshlrev_b32", 32); RUN_VALU(105, "v_add_u32", 32);
    RUN_VALU(21, "s_and_b64", 32); RUN_VALU(22, "s_bcnt1_i32_b64", 32); RUN_VALU(25, "s_brev_b64", 32); RUN_VALU(26, "s_add_u32 + s_addc_u32 (per pair)", 32);
    RUN_VALU(27, "s_mov exec save / set / restore (per triple)", 32);
    RUN_VALU(23, "v_fma + s_and alternating (per pair)", 32); RUN_VALU(24, "3 v_fma + 1 v_rcp (per 4)", 32);
    printf("# LDS: cycles per wave-instruction per SIMD; x1/4 = LDS-pipe cycles per instruction per CU when all 4 SIMDs stream\n");
    RUN_LDS(0, "ds_read_b128 broadcast"); RUN_LDS(1, "ds_read_b128 lane*16"); RUN_LDS(2, "ds_read_b128 4 runs of 16 lanes");
    RUN_LDS(3, "ds_write_b128 lane*16"); RUN_LDS(4, "ds_write_b128 exec = 4 lanes"); RUN_LDS(5, "ds_write_b128 exec = 16 lanes");
    RUN_LDS(6, "ds_write_b32 lane*4"); RUN_LDS(7, "ds_write_b32 exec = 8 lanes"); RUN_LDS(8, "ds_read_b32 lane*4"); RUN_LDS(9, "ds_read2_b32"); RUN_LDS(10, "ds_read_b64 lane*8");
    printf("# L1-resident global loads: cycles per wave-load per SIMD (x1/4 = vector-memory-pipe cycles per load per CU)\n");
    RUN_VMEM(0, "dwordx4 all lanes one address"); RUN_VMEM(1, "dwordx4 lane*16 (8 lines)"); RUN_VMEM(2, "dwordx4 units 4 lanes x 32 B, 16 lines");
    RUN_VMEM(3, "dwordx4 4 runs of 16 lanes"); RUN_VMEM(4, "dwordx4 exec 1 lane/quad (16), 16 lines"); RUN_VMEM(5, "dwordx4 exec lanes 0..15, 16 lines");
    RUN_VMEM(6, "dwordx4 exec 4 lanes"); RUN_VMEM(7, "dwordx4 MFMA A layout (16 lines x 4 chunks)"); RUN_VMEM(8, "dwordx4 exec lanes 0..15 x 32 B contiguous");
    RUN_VMEM(9, "dwordx4 units 2 lanes x 64 B, 32 lines"); RUN_VMEM(10, "dword lane*4"); RUN_VMEM(11, "dwordx2 lane*8");
    printf("# matrix pipe\n");
    run("v_mfma_f32_16x16x32_bf16 (4 accumulators)", 16, 2000, [](int blocks, int nit) { hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(256), 0, 0, d_out, nit); });
    run("1 mfma + 4 v_fma interleaved (per group of 5)", 16, 2000, [](int blocks, int nit) { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, d_out, nit); });
    return 0;
}
