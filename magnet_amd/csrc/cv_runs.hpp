// cv_runs.hpp — helpers shared by the run-based production matchers (cost_volume_v3.hip, cost_volume_v4.hip): wave-private LDS
// access by 32-bit byte address, stores / selects under 64-bit scalar lane masks, the 4-lane DPP reduction, run bookkeeping.
#pragma once
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
// float -> uint32 with the hardware's saturation (v_cvt_u32_f32: negative / NaN -> 0, >= 2^32 -> 0xffffffff).  A C cast of an
// out-of-range float is undefined; the projected coordinate of an out-of-window candidate can be anything.
__device__ __forceinline__ uint32_t v3_cvt_u32_sat(float x) {
    uint32_t r;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
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

// The four tap correlations of an item sit in four consecutive 4-lane banks of one 16-lane row (c00, c10, c01, c11).  Turn them
// into the quad form {c00, c10 - c00, c01 - c00, (c11 - c01) - (c10 - c00)} in place with two bank-masked DPP subtractions (a
// bank the mask leaves out keeps its value), so that the bilinear combine of a candidate is three fma instead of four weights
// and four multiply-adds: c = c00 + bx * dx + by * dy + (bx * by) * dxy.
__device__ __forceinline__ float v3_quadform16(float x) {
    asm volatile("s_nop 1\n\tv_subrev_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"      // banks 1, 3: x -= x[lane - 4]
                 "s_nop 1\n\tv_subrev_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc"            // banks 2, 3: x -= x[lane - 8]
                 : "+v"(x));
    return x;
}


// s_and_saveexec form of a masked store (2 scalar instructions around the store instead of 3)
__device__ __forceinline__ void v3_st2_mask(uint64_t mask, uint32_t addr, uint32_t lo, uint32_t hi) {
    uint64_t save;
    const uint64_t val = ((uint64_t)hi << 32) | lo;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(addr), "v"(val) : "memory", "scc");
}


// ---- round 4: helpers shared by cost_volume_v3.hip / _v4.hip / _v5.hip --------------------------------------------------------------
// 16-byte loads through a global-address-space pointer (HIP's uint4 / float4 classes cannot be dereferenced through address_space(1))
typedef const __attribute__((address_space(1))) unsigned char* cvr_gptr;
__device__ __forceinline__ uint4 v3_gld_u4(cvr_gptr q) {
    const v3_u32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) v3_u32x4*>(q);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 v3_gld_f4(cvr_gptr q) {
    const v3_f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) v3_f32x4*>(q);
    return make_float4(v.x, v.y, v.z, v.w);
}
// a wave-uniform pointer the compiler could not prove uniform (64-bit multiplies run on the vector unit): pin it into SGPRs, or a
// buffer descriptor built from it lives in VGPRs (every use wrapped in a readfirstlane waterfall loop) and a plain vector load adds
// it per lane (v_lshl_add_u64) instead of using the scalar-base addressing mode
__device__ __forceinline__ const void* v4_uniform_ptr(const void* q) {
    const unsigned long long a = (unsigned long long)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
typedef __attribute__((address_space(1))) unsigned char v4_gu8;        // global address space (an integer-built generic pointer compiles to flat_ accesses)
typedef __attribute__((address_space(1))) uint16_t v4_gu16;
typedef __attribute__((address_space(1))) float v4_gf32;
__device__ __forceinline__ v4_gu8* v4_uniform_gptr(const void* q) { return (v4_gu8*)(unsigned long long)v4_uniform_ptr(q); }
// 32-bit population count of a 64-bit scalar mask (clang keeps __builtin_popcountll in 64 bits and then compares it on the vector unit)
__device__ __forceinline__ int v4_popc(uint64_t m) {
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}
__device__ __forceinline__ void v4_st1_mask(uint64_t mask, uint32_t addr, uint32_t val) {
    uint64_t save;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(mask), "v"(addr), "v"(val) : "memory", "scc");
}
typedef void __attribute__((address_space(3)))* v4_lptr_t;
#define V4_LPTR(a) reinterpret_cast<v4_lptr_t>(a)


}  // namespace magnet
