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

constexpr uint32_t FKEY_CLOSED = 0xffffffffu;

}  // namespace magnet
