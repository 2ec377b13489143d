// cost_volume_v3.hip — production matcher for D > 32, round 3 (one reference pixel per wave iteration, lane = candidate).
//
// What the round-2 kernel (cost_volume_fast64.hip) was bound by, measured (profiles/r3/issue_rate.txt, profiles/r2/pmc_kernels.txt):
//   * the vector ALU: on gfx950 only v_add/sub/mul_f32 (2.6 cycles per wave64 instruction), v_add_u32 and v_fma_f32 (3.3)
//     run at the full rate; compares, selects, floor / cvt, shifts, DPP moves, v_mbcnt, v_dot2c cost 4.3 - 4.5, v_rcp_f32
//     8.2.  Its 86 instructions per (pixel, view) were ~320 cycles of the 459 a SIMD had per (pixel, view): 70 % busy.
//   * the vector-memory pipe: a dwordx4 (or dwordx2) wave-load occupies it for >= 16 cycles WHATEVER the number of active
//     lanes (one cycle per (lane quad, cache line) pair, minimum 16), i.e. cost is per instruction: 4.7 loads per (pixel,
//     view) = 75 cycles of the CU's ~115 per (pixel, view): 65 - 75 % busy (TA_BUSY).  2 of the 4.7 were the (mu, sigma) taps
//     that every candidate lane fetched for itself (2 KB through the pipe for ~112 distinct bytes).
// Both at ~70 % is where queueing sets in; this kernel cuts both:
//   1. RUNS.  The 64 candidates of a pixel are sorted along the epipolar segment, so lanes on the same source quad are
//      contiguous runs (~4.5 per view).  One detection per view (DPP compare with the previous lane + ballot + v_mbcnt)
//      numbers the in-window runs of all VG views of the group; everything per-quad is then stored ONCE per run in a
//      wave-private LDS slot and read by the run's lanes at the same address `raddr = run * 16`:
//        - the (mu, sigma) quad: `n_runs` LOADER lanes fetch the quads of ALL views of the group with one pair of
//          loads (0.5 instead of 2 load instructions per view);
//        - the four tap correlations of the quad.
//      Which runs have an open gate is decided on the scalar unit (bit-reversed carry trick over the gate and leader
//      ballots), flagged in the run's slot and compacted by the loader lanes once per group: no second DPP / v_mbcnt pass.
//   2. QUAD FORM.  The source (mu, sigma) map is stored per quad origin as {v00, v10-v00, v01-v00, v11-v10-v01+v00}
//      (magnet_pack_gmm_quad): bilinear interpolation = 3 fma per channel, no weights (was 2 sub + 4 mul + 8 fma).
//   3. Float index arithmetic (full-rate fma instead of cvt / min / mad chains), scalar view bases, reference vectors of
//      the wave's pixels staged once in LDS (one load instruction per 8 pixels instead of two per pixel).
// Arithmetic and tolerance contract: as cost_volume_fast.hip (fma-contracted geometry, one v_rcp_f32, padded-map texel
// coordinates, fp32 view sum); additionally the (mu, sigma) and correlation interpolations use the quad form /
// difference-form weights, which changes results by fp32 rounding only (homography.py:150-152,155-159).
#include "cv_fast_common.hpp"

namespace magnet {

typedef __attribute__((address_space(3))) unsigned char v3_lds_u8;
// wave-private LDS is addressed by 32-bit byte addresses (the run slot address `raddr` is a per-lane value); clang vector
// types, because HIP's float4 / uint4 classes have no address-space-3 assignment operators
typedef float v3_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t v3_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t v3_u32x2 __attribute__((ext_vector_type(2)));
#define V3_LDS(T, a) (*reinterpret_cast<__attribute__((address_space(3))) T*>(a))
__device__ __forceinline__ float4 v3_ld_f4(uint32_t a) { const v3_f32x4 v = V3_LDS(const v3_f32x4, a); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 v3_ld_u4(uint32_t a) { const v3_u32x4 v = V3_LDS(const v3_u32x4, a); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 v3_ld_u2(uint32_t a) { const v3_u32x2 v = V3_LDS(const v3_u32x2, a); return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint32_t v3_ld_u1(uint32_t a) { return V3_LDS(const uint32_t, a); }
__device__ __forceinline__ void v3_st_f4(uint32_t a, float4 v) { V3_LDS(v3_f32x4, a) = v3_f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void v3_st_u4(uint32_t a, uint4 v) { V3_LDS(v3_u32x4, a) = v3_u32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void v3_st_u2(uint32_t a, uint2 v) { V3_LDS(v3_u32x2, a) = v3_u32x2{v.x, v.y}; }
__device__ __forceinline__ void v3_st_u1(uint32_t a, uint32_t v) { V3_LDS(uint32_t, a) = v; }
__device__ __forceinline__ void v3_st_f1(uint32_t a, float v) { V3_LDS(float, a) = v; }

// Masked stores take the lane mask as a 64-bit SCALAR operand (no per-lane predicate has to be materialised): v3_st1_mask / v3_st2_mask below.

// select by a 64-bit scalar lane mask (bit set -> t)
__device__ __forceinline__ float v3_sel_f(uint64_t mask, float t, float f) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask));
    return r;
}
__device__ __forceinline__ uint32_t v3_sel_u(uint64_t mask, uint32_t t, uint32_t f) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(mask));
    return r;
}

// sum over aligned groups of 4 lanes in two DPP adds (the generic helper costs a third instruction); the s_nop covers the
// 2 wait states a DPP read needs after a VALU write of the same register (the assembler does not insert them in inline asm)
__device__ __forceinline__ float v3_reduce4(float v) {
    float t, r;
    // volatile: a cross-lane operation must not be sunk into the divergent `if (sub == 0)` that consumes its result
    asm volatile("s_nop 3\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(v));   // (the producer is a v_dot2c: hipcc itself leaves 3 wait states)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(t));
    return r;
}

// Leaders (first lanes of runs, ballot L) whose run holds at least one open gate (ballot G; every G lane belongs to a run).
// In bit-reversed order a run is [.., leader] with the leader on top: adding the run's gate bits to the all-ones run body
// carries into the leader's (zero) position exactly when the body holds a gate bit; the leader's own gate bit is OR-ed in.
__device__ __forceinline__ uint64_t v3_open_leaders(uint64_t G, uint64_t L) {
    const uint64_t Gr = __builtin_bitreverse64(G), Lr = __builtin_bitreverse64(L);
    const uint64_t Z = ~Lr;
    const uint64_t S = (Gr & Z) + Z;
    return __builtin_bitreverse64((S & Lr) | (Gr & Lr));
}

// fixed offsets inside a wave's LDS region
constexpr int V3_CAP = 64;                         // runs of one view group
constexpr int V3_NPX = 8;                          // reference pixels per wave (raster-order row segments: cost_volume_fast64.hip)
constexpr int V3_CT = 0;                           // [CAP + 1] x 16 B: per run its quad key, later its 4 tap correlations; slot CAP = dump
constexpr int V3_G0 = 65 * 16;                     // [CAP] x 16 B: mu quad form of the run's quad
constexpr int V3_G1 = V3_G0 + 64 * 16;             // [CAP] x 16 B: sigma quad form
constexpr int V3_IT = V3_G1 + 64 * 16;             // [CAP + 1] x 8 B: open runs {feature byte offset, LDS address of the run's slot}
constexpr int V3_MS = (V3_IT + 65 * 8 + 15) / 16 * 16;   // [NPX] x 8 B: (mu, sigma) of the wave's reference pixels
constexpr int V3_FIX = V3_MS + V3_NPX * 8;         // then: view table [Vr] x 8 B, projection table [Vr][NPX] x 32 B, reference vectors, output stage

// s_and_saveexec form of the masked stores (2 scalar instructions around the store instead of 3)
template <int OFF>
__device__ __forceinline__ void v3_st1_mask(uint64_t mask, uint32_t addr, uint32_t val) {
    uint64_t save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3 offset:%4\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(addr), "v"(val), "n"(OFF) : "memory", "scc");
}
__device__ __forceinline__ void v3_st2_mask(uint64_t mask, uint32_t addr, uint32_t lo, uint32_t hi) {
    uint64_t save;
    const uint64_t val = ((uint64_t)hi << 32) | lo;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(addr), "v"(val) : "memory", "scc");
}

// CPL / FULL / LPU: VALU correlation units of LPU lanes x CPL 16-byte chunks (as cv_fast_kernel); VG = views per group;
// OPT bit 0: write the gate bits (debug / parity tests); bits 1.. (dev builds): timing ablations
template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG, int OPT>
__global__ __launch_bounds__(256, MINW) void cv_v3_kernel(const CvParams p) {
    constexpr bool GBITS = (OPT & 1) != 0;
    constexpr bool NO_CORR = (OPT & 2) != 0;              // dev: no feature loads / dot products
    constexpr bool NO_GMM = (OPT & 4) != 0;               // dev: no (mu, sigma) quad loads
    constexpr bool NO_FLOAD = (OPT & 8) != 0;             // dev: dot products on registers, no feature loads
    constexpr bool NO_DOTS = (OPT & 16) != 0;             // dev: feature loads waited for, no dot products
    constexpr int NPX = V3_NPX;
    constexpr int IPP = 64 / (4 * LPU);                   // items per correlation pass
    constexpr int NPASS_D = 16 / IPP > 4 ? 4 : 16 / IPP;  // passes fetched together (default: 16 items)
    constexpr int NPASS = ((OPT >> 8) & 15) ? ((OPT >> 8) & 15) : NPASS_D;
    constexpr int CSTR = LPU * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block -> (frame b, pixel row y, segment tx): the XCD-aware remap of cv_common.hpp with the two runtime divisions done by
    // multiplication with host-computed reciprocals (all scalar: blockIdx is wave-uniform; the generic integer division is ~25
    // vector instructions each)
    int tile, b;
    {
        const unsigned n = gridDim.x, bid = blockIdx.x;
        const unsigned qn = n / NUM_XCD, rn = n % NUM_XCD;
        const unsigned xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
        const unsigned start = (xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        const unsigned logical = start + idx;
        b = p.magic_tiles ? (int)__umulhi(logical, p.magic_tiles) : (int)logical;        // magic 0 = division by 1
        tile = (int)(logical - (unsigned)b * (unsigned)(p.tiles_x * p.tiles_y));
    }
    const int y = p.magic_tiles_x ? (int)__umulhi((unsigned)tile, p.magic_tiles_x) : tile;   // raster order: see cost_volume_fast64.hip's launcher
    const int tx = tile - y * p.tiles_x;
    const int yc = min(y, p.h - 1);
    const int x_base = (tx * 4 + wv) * NPX;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const int JB = (p.D + 63) / 64;
    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    const int nchunk = (int)(texel_bytes / 16);
    const uint32_t map_texels = (uint32_t)(Hp * Wp);
    const uint32_t vstride = (uint32_t)p.B * map_texels;                                    // texels between the views of a frame (view-major)

    // ---- valid views (homography.py:97), compacted: the loop below never sees an invalid view ----
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    const int nval = __popcll(vmask);
    const int Vr = (p.V + VG - 1) / VG * VG;                                                // table rows: views rounded up to whole groups

    // ---- wave-private LDS ----
    const int vt_bytes = Vr * 8, pv_bytes = Vr * NPX * 32, rf_bytes = NPX * (int)texel_bytes;
    const int out_bytes = p.cost_hi ? 0 : NPX * 64 * 4;
    const int wave_bytes = V3_FIX + (vt_bytes + 15) / 16 * 16 + pv_bytes + rf_bytes + out_bytes;
    const uint32_t wb = (uint32_t)(uintptr_t)(v3_lds_u8*)smem + (uint32_t)(wv * wave_bytes);
    const uint32_t vtb = wb + V3_FIX, pvb = vtb + (vt_bytes + 15) / 16 * 16, rfb = pvb + pv_bytes, outb = rfb + rf_bytes;

    // ---- per (compact view c, pixel q): depth-linear projection terms; per compact view: {texel offset of the view, view index} ----
    for (int e = lane; e < NPX * Vr; e += 64) {
        const int q = e % NPX, c = min(e / NPX, max(nval - 1, 0));                          // padding rows repeat the last valid view (masked in the loop)
        // view index of the c-th valid view: c-th set bit of vmask
        int v = 0;
        {
            unsigned long long m = vmask;
            for (int i = 0; i < c; ++i) m &= m - 1;
            v = m ? __builtin_ctzll(m) : 0;
        }
        const int xc = min(x_base + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, yc, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        v3_st_f4(pvb + e * 32, make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz));
        v3_st_f4(pvb + e * 32 + 16, make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz));
        if (q == 0) v3_st_u2(vtb + (e / NPX) * 8, make_uint2((uint32_t)v * vstride, (uint32_t)v));
    }
    // ---- reference vectors and (mu, sigma) of the wave's pixels (contiguous in their rows): staged once ----
    {
        const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) +
            ((size_t)b * hw + (size_t)yc * p.w) * texel_bytes;
        // the NPX pixels of the segment are contiguous in the channel-last row: a plain copy of NPX * texel_bytes bytes, clamped
        // at the row's end (pixels past the row are never processed)
        const uint32_t row_bytes = (uint32_t)p.w * texel_bytes, seg0 = (uint32_t)x_base * texel_bytes;
        for (uint32_t e = (uint32_t)lane * 16u; e < (uint32_t)NPX * texel_bytes; e += 1024u)
            v3_st_u4(rfb + e, *reinterpret_cast<const uint4*>(ref_row + min(seg0 + e, row_bytes - 16u)));
        if (lane < NPX) {
            const size_t pixr = (size_t)yc * p.w + min(x_base + lane, p.w - 1);
            v3_st_u2(wb + V3_MS + lane * 8, make_uint2(__float_as_uint(p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr]),
                                                      __float_as_uint(p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr])));
        }
    }
    fwave_lds_fence();

    const uint32_t xlim = __float_as_uint((float)(p.w + 1)), ylim = __float_as_uint((float)(p.h + 1));   // see cost_volume_fast.hip
    const float Wpf = (float)Wp;
    const int sub = lane & (LPU - 1), tap = (lane / LPU) & 3, upair = lane / (4 * LPU);          // correlation roles
    const uint32_t lane_src_off = (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes + (uint32_t)sub * 16u;
    const uint32_t lane16 = wb + (uint32_t)lane * 16u;                                       // loader lane -> its run slot
    const uint32_t tap4 = (uint32_t)tap * 4u;
    const float invV = 1.0f / (float)p.V;
    const uint32_t kmax = (uint32_t)(p.V - 1) * vstride + map_texels - 1u;                  // last quad index relative to frame b, view 0
    const unsigned char* const src_b = reinterpret_cast<const unsigned char*>(p.src_feat) + (size_t)b * map_texels * texel_bytes;
    const unsigned char* const gq_b = reinterpret_cast<const unsigned char*>(p.src_gmq) + (size_t)b * map_texels * 32;
    const float kappa = p.kappa;

    const int npix = min(NPX, p.w - x_base);                                                // pixels of the segment inside the row (may be <= 0)
    const uint32_t it_lane = wb + V3_IT + (uint32_t)upair * 8u;                               // correlation unit -> its item entry of pass 0
    const uint32_t rf_lane = rfb + (uint32_t)sub * 16u;
    // NCHW fp32 output: lane -> (candidate jj = it * 8 + lane / 8, pixel qq = lane % 8) of flush iteration `it`
    const uint32_t out_ld_lane = outb + (uint32_t)(((lane & 7) * 64 + (lane >> 3)) * 4);
    const uint32_t out_g_lane = (uint32_t)(((size_t)(lane >> 3) * hw + (size_t)(lane & 7)) * 4);

    for (int jb = 0; jb < JB; ++jb) {
        const int j = jb * 64 + lane;
        const float kj = p.k[min(j, p.D - 1)];
        const unsigned long long jmask = __builtin_amdgcn_ballot_w64(j < p.D);               // lanes that hold a candidate
        for (int q = 0; q < npix; ++q) {
            const int x = x_base + q;
            uint4 rvp[CPL];                                                               // this lane's chunk(s) of the pixel's reference vector
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc)
                rvp[cc] = (FULL || (sub + LPU * cc < nchunk)) ? v3_ld_u4(rf_lane + q * texel_bytes + cc * CSTR) : make_uint4(0, 0, 0, 0);
            float d;
            {
                const uint2 ms = v3_ld_u2(wb + V3_MS + q * 8);
                d = __builtin_fmaf(__uint_as_float(ms.y), kj, __uint_as_float(ms.x));     // MAGNET.py:155
            }
            d = v3_sel_f(jmask, d, __builtin_nanf(""));                                    // lane without a candidate -> out of the window below
            float acc = 0.f;

            // (item, tap) dot products of the n open runs listed in the item table -> the runs' slots
            auto correlate = [&](const int n) {
                for (int ps = 0; ps < n; ps += IPP * NPASS) {
                    uint4 sv[NPASS][CPL];
                    uint2 ent[NPASS];
                    const uint32_t ita = it_lane + (uint32_t)ps * 8u;
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) ent[a] = v3_ld_u2(ita + a * (IPP * 8));             // past the list: pad entries (dump slot)
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {
                        if (a > 0 && ps + IPP * a >= n) break;                            // wave-uniform: this pass holds no item
                        const unsigned char* sp = src_b + (ent[a].x + lane_src_off);
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc)
                            sv[a][cc] = NO_FLOAD ? make_uint4(ent[a].x, ent[a].y, rvp[cc].x, rvp[cc].y)
                                      : (FULL || (sub + LPU * cc < nchunk)) ? *reinterpret_cast<const uint4*>(sp + cc * CSTR) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {
                        if (a > 0 && ps + IPP * a >= n) break;
                        float part = 0.f;
                        if (NO_DOTS) {
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) part += __uint_as_float(sv[a][cc].x ^ sv[a][cc].w);
                        } else {
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) part = fdot_chunk(rvp[cc], sv[a][cc], part, FeatT());
                            part = LPU == 8 ? freduce8(part) : v3_reduce4(part);
                        }
                        if (sub == 0) v3_st_f1(ent[a].y + tap4, part);                     // pad units write the dump slot
                    }
                }
            };

            int gstep = VG;                                                               // views per group (1 after an overflow)
            for (int g0 = 0; g0 < nval;) {
                const int nact = min(gstep, nval - g0);
                // ---------------- phase A: geometry and runs of equal quads, VG independent chains ----------------
                // Lane predicates are kept as 64-bit scalar masks (ballots) throughout: compares write SGPR pairs, the logic
                // runs on the scalar unit, and masked stores / selects take the mask as an operand.
                float bx[VG], by[VG], fxy[VG], zw[VG];
                uint32_t keyf[VG], raddr[VG];
                unsigned long long Wb[VG], Lb[VG], Gb[VG];                                 // in window; run leaders; open gates
                int nrun[VG];
                const uint32_t pva = pvb + (uint32_t)((g0 * NPX + q) * 32);
                const uint32_t vta = vtb + (uint32_t)(g0 * 8);
                {
                    float4 pa[VG], pb[VG];
                    uint2 vt[VG];
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        pa[u] = v3_ld_f4(pva + u * (NPX * 32)); pb[u] = v3_ld_f4(pva + u * (NPX * 32) + 16);
                        vt[u] = v3_ld_u2(vta + u * 8);
                    }
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        const float Px = __builtin_fmaf(pa[u].x, d, pb[u].x);            // homography.py:132
                        const float Py = __builtin_fmaf(pa[u].y, d, pb[u].y);
                        const float Pz = __builtin_fmaf(pa[u].z, d, pb[u].z);
                        zw[u] = __builtin_fmaf(pa[u].w, d, pb[u].w);                     // homography.py:137-138
                        const float rz = __builtin_amdgcn_rcpf(Pz);                      // homography.py:133
                        const float ixs = __builtin_fmaf(Px, rz, 0.5f);                  // = (u - 0.5) + 1: padded-map texel coordinate
                        const float iys = __builtin_fmaf(Py, rz, 0.5f);
                        const float x0f = __builtin_floorf(ixs), y0f = __builtin_floorf(iys);
                        bx[u] = ixs - x0f; by[u] = iys - y0f;
                        fxy[u] = bx[u] * by[u];
                        const unsigned long long wx = __builtin_amdgcn_ballot_w64(__float_as_uint(ixs) < xlim);
                        const unsigned long long wy = __builtin_amdgcn_ballot_w64(__float_as_uint(iys) < ylim);
                        Wb[u] = (u < nact) ? (wx & wy) : 0ull;                            // views past the group's end (tail group, fallback) are off
                        const uint32_t key = (uint32_t)__builtin_fmaf(y0f, Wpf, x0f);    // quad origin in the padded map (exact inside the window)
                        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x138, 0xf, 0xf, true);   // wave_shr:1, lane 0 <- 0
                        // leader = in window and (lane 0, or another quad than the previous lane, or the previous lane is outside the window: its
                        // key is garbage that may alias this quad's index)
                        Lb[u] = (__builtin_amdgcn_ballot_w64(key != prev) | 1ull | ~(Wb[u] << 1)) & Wb[u];
                        nrun[u] = __popcll(Lb[u] >> 1) + 1;                               // slot 0 of the view: lanes in front of its first leader
                        keyf[u] = key + vt[u].x;                                          // + the view's offset: quad index relative to (frame b, view 0)
                        if (GBITS) raddr[u] = vt[u].y;                                    // (debug output only: the view index, parked until the gate)
                    }
                }
                int n_tot = 0;
#pragma unroll
                for (int u = 0; u < VG; ++u) n_tot += nrun[u];
                if (n_tot > V3_CAP) { gstep = 1; continue; }                              // (only with nact > 1) redo view by view
                uint32_t vidx[VG];
                {
                    uint32_t sbase = wb;                                                  // LDS address of the view's first slot
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        if (GBITS) vidx[u] = raddr[u];
                        const unsigned long long Ls = Lb[u] >> 1;
                        const uint32_t cnt = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ls, 0u));
                        raddr[u] = (cnt << 4) + sbase;
                        v3_st1_mask<V3_CT>(Lb[u], raddr[u], keyf[u]);
                        sbase += (uint32_t)nrun[u] * 16u;
                    }
                }
                fwave_lds_fence();
                // ---------------- loader lanes: the (mu, sigma) quads of all runs of the group, one pair of loads straight into LDS ----------------
                if (!NO_GMM) {
                    if (lane < n_tot) {
                        const uint32_t k = min(v3_ld_u1(lane16 + V3_CT), kmax);           // slots no leader wrote hold stale keys: clamped
                        const unsigned char* gp = gq_b + (k << 5);
                        v3_st_f4(lane16 + V3_G0, *reinterpret_cast<const float4*>(gp));
                        v3_st_f4(lane16 + V3_G1, *reinterpret_cast<const float4*>(gp + 16));
                    }
                    fwave_lds_fence();
                }
                // ---------------- gate; the leaders of runs with an open gate enter the item table ----------------
                int n_items = 0;
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    const float4 m = v3_ld_f4(raddr[u] + V3_G0), s = v3_ld_f4(raddr[u] + V3_G1);
                    const float mu_w = __builtin_fmaf(fxy[u], m.w, __builtin_fmaf(by[u], m.z, __builtin_fmaf(bx[u], m.y, m.x)));   // homography.py:151
                    const float sg_w = __builtin_fmaf(fxy[u], s.w, __builtin_fmaf(by[u], s.z, __builtin_fmaf(bx[u], s.y, s.x)));   // homography.py:152
                    Gb[u] = __builtin_amdgcn_ballot_w64(__builtin_fabsf(zw[u] - mu_w) < sg_w * kappa) & Wb[u];   // homography.py:157-158
                    const unsigned long long Lo = v3_open_leaders(Gb[u], Lb[u]);
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(Lo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Lo, 0u));
                    v3_st2_mask(Lo, (rank << 3) + (wb + V3_IT + (uint32_t)n_items * 8u), __umul24(keyf[u], texel_bytes), raddr[u] + V3_CT);
                    n_items += __popcll(Lo);
                    if (GBITS) {
                        if (j < p.D && u < nact)
                            p.gate_bits[(((size_t)b * p.V + vidx[u]) * p.D + j) * hw + (size_t)y * p.w + x] = (uint8_t)v3_sel_u(Gb[u], 1u, 0u);
                    }
                }
                if (n_items == 0) { g0 += nact; continue; }                               // wave-uniform: nothing open in this group
                if (IPP > 1)                                                              // pad entries behind the list: the last pass's idle units -> dump slot
                    v3_st2_mask((1ull << (IPP - 1)) - 1ull, wb + V3_IT + ((uint32_t)n_items + (uint32_t)lane) * 8u, 0u, wb + V3_CT + V3_CAP * 16);
                fwave_lds_fence();
                if (!NO_CORR) correlate(n_items);
                fwave_lds_fence();
                // ---------------- bilinear combine + view accumulation ----------------
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    const float4 c4 = v3_ld_f4(raddr[u] + V3_CT);
                    const float w10 = bx[u] - fxy[u], w01 = by[u] - fxy[u];
                    const float w00 = (1.0f - bx[u]) - w01;
                    float c = c4.x * w00;                                                  // homography.py:150,155
                    c = __builtin_fmaf(c4.y, w10, c);
                    c = __builtin_fmaf(c4.z, w01, c);
                    c = __builtin_fmaf(c4.w, fxy[u], c);
                    acc += v3_sel_f(Gb[u], c, 0.f);                                        // homography.py:159,116 (fp32 here)
                }
                fwave_lds_fence();                                                        // the slots are rewritten by the next group
                g0 += nact;
            }
            const float cval = acc * invV;                                                // homography.py:118,120
            if (p.cost_hi) {
                // split-bf16 channel-last output for the conv kernel: lanes = 64 consecutive channels of one padded-grid row
                const size_t e0 = (((size_t)b * Hp + (y + 1)) * Wp + (x + 1)) * (size_t)p.cost_ld + (size_t)(jb * 64);   // scalar
                if (j < p.D) {
                    const uint16_t hi = f32_to_bf16_rne(cval);
                    const uint16_t lo = f32_to_bf16_rne(cval - bf16_to_f32(hi));
                    (p.cost_hi + e0)[lane] = hi; (p.cost_lo + e0)[lane] = lo;
                }
                continue;
            }
            v3_st_f1(outb + (uint32_t)(q * 64 + lane) * 4u, cval);
            if (q == npix - 1) {
                // ---- npix px x 64 results: LDS -> coalesced row segments of cost[b, j, y, :] ----
                fwave_lds_fence();
                const unsigned char* gbase = reinterpret_cast<const unsigned char*>(p.cost + (size_t)b * p.cost_bstride + (size_t)(jb * 64) * hw + (size_t)y * p.w + x_base);
                const bool px_ok = (lane & 7) < npix;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float v = __uint_as_float(v3_ld_u1(out_ld_lane + it * 32));
                    if (px_ok && jb * 64 + it * 8 + (lane >> 3) < p.D)
                        *reinterpret_cast<float*>(const_cast<unsigned char*>(gbase) + (size_t)it * 8 * hw * 4 + out_g_lane) = v;
                }
                fwave_lds_fence();
            }
        }
    }
}

static size_t v3_lds_bytes(const CvParams& p, int vg) {
    const size_t esz = p.feat_bf16 ? 2 : 4;
    const int Vr = (p.V + vg - 1) / vg * vg;
    return (size_t)4 * (V3_FIX + (Vr * 8 + 15) / 16 * 16 + Vr * V3_NPX * 32 + V3_NPX * p.F * esz + (p.cost_hi ? 0 : V3_NPX * 64 * 4));
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG>
static hipError_t launch_v3_v(const CvParams& p0, hipStream_t stream) {
    CvParams p = p0;
    p.npx = V3_NPX;
    p.tiles_x = (p.w + 4 * p.npx - 1) / (4 * p.npx);
    p.tiles_y = p.h;
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    // n / d = umulhi(n, ceil(2^32 / d)) exactly while n * d < 2^32 (n < grid size; checked by launch_cv_v3)
    const uint64_t nt = (uint64_t)p.tiles_x * p.tiles_y;
    p.magic_tiles = nt > 1 ? (uint32_t)((((uint64_t)1 << 32) + nt - 1) / nt) : 0u;                       // 0 = division by 1
    p.magic_tiles_x = p.tiles_x > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)p.tiles_x - 1) / (uint64_t)p.tiles_x) : 0u;
    const size_t lds = v3_lds_bytes(p, VG);
#ifdef MAGNET_DEV
    if (p.ablate & 0x4000) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 0x200>), grid, block, lds, stream, p); return hipGetLastError(); }   // 2 passes in flight
    if (p.ablate & 0x200) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 2>), grid, block, lds, stream, p); return hipGetLastError(); }
    if (p.ablate & 0x10000) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 8>), grid, block, lds, stream, p); return hipGetLastError(); }
    if (p.ablate & 0x20000) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 16>), grid, block, lds, stream, p); return hipGetLastError(); }
    if (p.ablate & 0x400) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 4>), grid, block, lds, stream, p); return hipGetLastError(); }
#endif
    if (p.gate_bits) hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 1>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 0>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU>
static hipError_t launch_v3(const CvParams& p, hipStream_t stream) {
    int vg = p.V >= 4 ? 4 : p.V;
    if (p.V > 4 && p.V % 4 != 0 && (p.V % 3 == 0 || p.V % 4 < p.V % 3)) vg = 3;
#ifdef MAGNET_DEV
    if ((p.ablate & 0x8000) && vg > 2) vg = 2;                                        // dev: two views per group
#endif
    switch (vg) {
        case 1: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 1>(p, stream);
        case 2: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 2>(p, stream);
        case 3: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 3>(p, stream);
        default: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 4>(p, stream);
    }
}

// D > 32, candidates sampled in the kernel, quad-form (mu, sigma) map given; called by launch_cv_fast, which has checked the rest.
hipError_t launch_cv_v3(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if (p.D <= 32 || !p.src_gmq) return hipSuccess;
    if ((size_t)p.V * p.B * (size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets over all views
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 24)) return hipSuccess;                                    // quad keys exact in fp32
    const int nchunk = (int)(p.F * esz / 16);
    if (v3_lds_bytes(p, 4) > 64 * 1024) return hipSuccess;
    {   // the scalar block -> tile divisions by reciprocal multiplication are exact while grid * tiles < 2^32
        const uint64_t tiles = (uint64_t)((p.w + 4 * V3_NPX - 1) / (4 * V3_NPX)) * (uint64_t)p.h;
        if (tiles * (uint64_t)p.B * tiles >= ((uint64_t)1 << 32)) return hipSuccess;
    }
    *handled = true;
    if (p.feat_bf16) {
#ifdef MAGNET_DEV
        if (nchunk == 8 && (p.ablate & 0x800))  return launch_v3<uint16_t, 2, true, 6, 4>(p, stream);    // dev: occupancy A/B
        if (nchunk == 8 && (p.ablate & 0x1000)) return launch_v3<uint16_t, 2, true, 8, 4>(p, stream);
        if (nchunk == 8 && (p.ablate & 0x2000)) return launch_v3<uint16_t, 2, true, 5, 4>(p, stream);
#endif
        if (nchunk == 8)  return launch_v3<uint16_t, 2, true, 4, 4>(p, stream);          // F = 64: 4 lanes x 32 B per (item, tap) unit; 4 waves per SIMD (no spills; 5 measured equal)
        if (nchunk <= 8)  return launch_v3<uint16_t, 1, false, 5, 8>(p, stream);
        if (nchunk <= 16) return launch_v3<uint16_t, 2, false, 5, 8>(p, stream);
    } else {
        if (nchunk == 16) return launch_v3<float, 2, true, 5, 8>(p, stream);             // F = 64
        if (nchunk <= 8)  return launch_v3<float, 1, false, 5, 8>(p, stream);
        if (nchunk <= 16) return launch_v3<float, 2, false, 5, 8>(p, stream);
        if (nchunk <= 32) return launch_v3<float, 4, false, 4, 8>(p, stream);
    }
    *handled = false;
    return hipSuccess;
}

}  // namespace magnet
