// cv_fast_common.hpp — helpers shared by the production matcher kernels (cost_volume_fast.hip, cost_volume_fast64.hip).
#pragma once
#include "cv_common.hpp"

namespace magnet {

typedef __attribute__((ext_vector_type(2))) __bf16 fbf16x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 fbf16x8_t;
typedef __attribute__((ext_vector_type(4))) float ff32x4_t;

__device__ __forceinline__ float fdot_chunk(const uint4 a, const uint4 b, float acc, uint16_t) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fbf16x2_t, a.x), __builtin_bit_cast(fbf16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fbf16x2_t, a.y), __builtin_bit_cast(fbf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fbf16x2_t, a.z), __builtin_bit_cast(fbf16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(fbf16x2_t, a.w), __builtin_bit_cast(fbf16x2_t, b.w), acc, false);
    return acc;
}
__device__ __forceinline__ float fdot_chunk(const uint4 a, const uint4 b, float acc, float) {
    acc = __builtin_fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
    acc = __builtin_fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
    acc = __builtin_fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
    acc = __builtin_fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
    return acc;
}

__device__ __forceinline__ float freduce8(float v) {    // sum over aligned groups of 8 lanes
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}
__device__ __forceinline__ float freduce4(float v) {    // sum over aligned groups of 4 lanes
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));
    return v;
}

__device__ __forceinline__ void fwave_lds_fence() {
    // LDS operations of one wave execute in order; only the compiler must not reorder across this.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// key of a lane without an open quad.  Valid keys are below 2^24 (launchers check), and FKEY_CLOSED +- a step of the padded map is
// none either, so "previous lane's key + step == my key" needs no separate test for a closed previous lane.
constexpr uint32_t FKEY_CLOSED = 0x40000000u;

// Direction in which the candidates of one (pixel, view) travel through the source map as the depth grows (round 5, texel-pair
// items).  P(d) = t + r d (homography.py:132), so d(P_x / P_z)/dd = (r_x t_z - t_x r_z) / P_z^2: the sign and the dominant axis do
// not depend on d.  bit 0: the segment runs along y (pairs are rows); bit 1: towards smaller coordinates.  Speed only — the mode
// decides how many texel pairs consecutive quads share, never the result.
__device__ __forceinline__ uint32_t travel_mode(const PixelView& pv) {
    const float nx = pv.rpx * pv.kt2 - pv.kt0 * pv.rpz, ny = pv.rpy * pv.kt2 - pv.kt1 * pv.rpz;
    const bool rowm = __builtin_fabsf(ny) > __builtin_fabsf(nx);
    return (rowm ? 1u : 0u) | (((rowm ? ny : nx) < 0.f) ? 2u : 0u);
}

}  // namespace magnet
