// warp_math.hpp — per-sample projection / bilinear-tap arithmetic shared by the cost-volume kernels.
//
// The arithmetic mirrors, operation for operation, what the reference's ATen/BLAS CPU path
// rounds (and therefore what oracle/cost_volume_oracle.c restates):
//   * 3-term dot products (K*R, K*t, (K*R)*ray, R*ray) = a0*b0, then two fused accumulations
//     (reference: models/submodules/homography.py:99-102, sgemm)
//   * P = t_pix + r_pix*d, z_warp = t_z + r_z*d : separate multiply and add (homography.py:132,137)
//   * P / (P_z + 1e-10) and (P - c)/c : IEEE divisions (homography.py:133,143-146)
//   * clamp to [-10,10] (homography.py:147-148); unnormalise = fma(g+1, size/2, -0.5)
//     (ATen grid_sampler, align_corners=False); weights (x1-ix)*(y1-iy) ... (homography.py:150-152)
// This translation unit must be compiled with -ffp-contract=off: fused operations are spelled
// __builtin_fmaf explicitly, everything else must stay unfused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magnet {

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}

// Per (reference pixel, source view): the depth-linear projection P(d) = t + r*d.
struct PixelView {
    float rpx, rpy, rpz;   // (K R) ray
    float rcz;             // (R ray)_z
    float kt0, kt1, kt2;   // K t
    float tz;              // t_z
};

// K: (3,3) row-major intrinsics; T: (4,4) row-major [R|t]; ray: (rx, ry, rz).
__device__ __forceinline__ PixelView make_pixel_view(const float* __restrict__ K,
                                                     const float* __restrict__ T,
                                                     float r0, float r1, float r2) {
    const float R00 = T[0], R01 = T[1], R02 = T[2],  t0 = T[3];
    const float R10 = T[4], R11 = T[5], R12 = T[6],  t1 = T[7];
    const float R20 = T[8], R21 = T[9], R22 = T[10], t2 = T[11];
    PixelView pv;
    float kr[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float k0 = K[i * 3 + 0], k1 = K[i * 3 + 1], k2 = K[i * 3 + 2];
        kr[i * 3 + 0] = dot3(k0, k1, k2, R00, R10, R20);
        kr[i * 3 + 1] = dot3(k0, k1, k2, R01, R11, R21);
        kr[i * 3 + 2] = dot3(k0, k1, k2, R02, R12, R22);
        const float kt = dot3(k0, k1, k2, t0, t1, t2);
        if (i == 0) pv.kt0 = kt; else if (i == 1) pv.kt1 = kt; else pv.kt2 = kt;
    }
    pv.rpx = dot3(kr[0], kr[1], kr[2], r0, r1, r2);
    pv.rpy = dot3(kr[3], kr[4], kr[5], r0, r1, r2);
    pv.rpz = dot3(kr[6], kr[7], kr[8], r0, r1, r2);
    pv.rcz = dot3(R20, R21, R22, r0, r1, r2);
    pv.tz = t2;
    return pv;
}

// Grid constants shared by every sample of a launch.
struct GridConst {
    float cw, ch;     // w/2, h/2 (the reference's u_center, v_center)
    float sw, sh;     // ATen scaling_factor = size/2
};

struct Taps {
    float nw, ne, sw, se;  // bilinear weights, ATen order
    int   x0, y0;          // integer origin of the 2x2 quad (far out of range when not finite)
};

__device__ __forceinline__ float clamp10(float g) {
    // reference: coords[coords > 10] = 10; coords[coords < -10] = -10  (NaN passes through)
    g = (g > 10.0f) ? 10.0f : g;
    g = (g < -10.0f) ? -10.0f : g;
    return g;
}

// Sample position in texel units and the warped depth for candidate depth d.
__device__ __forceinline__ void project(const PixelView& pv, const GridConst& gc, float d,
                                        float& ix, float& iy, float& zw) {
    float Px = pv.kt0 + pv.rpx * d;
    float Py = pv.kt1 + pv.rpy * d;
    const float Pz = pv.kt2 + pv.rpz * d;
    const float zz = Pz + 1e-10f;
    Px = Px / zz;
    Py = Py / zz;
    zw = pv.tz + pv.rcz * d;
    const float gx = clamp10((Px - gc.cw) / gc.cw);
    const float gy = clamp10((Py - gc.ch) / gc.ch);
    ix = __builtin_fmaf(gx + 1.0f, gc.sw, -0.5f);
    iy = __builtin_fmaf(gy + 1.0f, gc.sh, -0.5f);
}

__device__ __forceinline__ Taps make_taps(float ix, float iy) {
    Taps t;
    const float x0 = __builtin_floorf(ix), y0 = __builtin_floorf(iy);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    t.nw = (x1 - ix) * (y1 - iy);
    t.ne = (ix - x0) * (y1 - iy);
    t.sw = (x1 - ix) * (iy - y0);
    t.se = (ix - x0) * (iy - y0);
    // the +-10 clamp bounds |ix|,|iy| by 5.5*size; anything else is NaN/garbage -> out of range
    const bool finite = (__builtin_fabsf(ix) < 1e9f) && (__builtin_fabsf(iy) < 1e9f);
    t.x0 = finite ? (int)x0 : -100000;
    t.y0 = finite ? (int)y0 : -100000;
    return t;
}

// nw_val*nw, then fused accumulation of ne, sw, se (ATen CPU grid_sampler's rounding).
__device__ __forceinline__ float bilerp(float a, float b, float c, float d, const Taps& t) {
    float v = a * t.nw;
    v = __builtin_fmaf(b, t.ne, v);
    v = __builtin_fmaf(c, t.sw, v);
    v = __builtin_fmaf(d, t.se, v);
    return v;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    return __uint_as_float(((uint32_t)h) << 16);
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace magnet
