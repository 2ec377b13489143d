// warp_math.hpp — per-sample projection / bilinear-tap arithmetic shared by the cost-volume kernels.
//
// The arithmetic mirrors, operation for operation, what the reference's ATen/BLAS CPU path
// rounds (and therefore what oracle/cost_volume_oracle.c restates):
//   * 3-term dot products of matrix-matrix products (K*R, (K*R)*ray, R*ray) = a0*b0, then two fused
//     accumulations (reference: models/submodules/homography.py:99-102, sgemm); the matrix-vector product K*t
//     (sgemv) = plain products summed left to right, unfused
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
        // K*t is a matrix-VECTOR product in the reference (sgemv): plain products, left to right, unfused
        const float kt = (k0 * t0 + k1 * t1) + k2 * t2;
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
    float rcw, rch;   // RN(1/cw), RN(1/ch) for the exact constant division below
    float sw, sh;     // ATen scaling_factor = size/2
};

// x / c for a launch constant c, correctly rounded (== IEEE division) in 3 operations:
// q = x*rc; r = x - q*c (exact, fused); q' = q + r*rc (fused)   [Markstein; rc = RN(1/c)].
__device__ __forceinline__ float div_const(float x, float c, float rc) {
    const float q = x * rc;
    const float r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, rc, q);
}

struct Taps {
    float nw, ne, sw, se;  // bilinear weights, ATen order
};

__device__ __forceinline__ float clamp10(float g) {
    // reference: coords[coords > 10] = 10; coords[coords < -10] = -10 (homography.py:147-148).
    // One v_med3_f32.  For finite g it is exactly that clamp.  The reference lets a NaN pass through
    // (-> a NaN sample position, which grid_sample treats as out of range); v_med3 maps NaN to one
    // of the bounds instead, i.e. to a position 4.5 images outside: the same all-zero sample.
    return __builtin_amdgcn_fmed3f(g, -10.0f, 10.0f);
}

// Sample position in texel units and the warped depth for candidate depth d.
__device__ __forceinline__ void project(const PixelView& pv, const GridConst& gc, float d,
                                        float& ix, float& iy, float& zw) {
    float Px = pv.kt0 + pv.rpx * d;
    float Py = pv.kt1 + pv.rpy * d;
    const float Pz = pv.kt2 + pv.rpz * d;
    const float zz = Pz + 1e-10f;
    // Px/zz and Py/zz, correctly rounded (== the reference's IEEE divisions) from ONE hardware
    // reciprocal: r1 = Newton-refined v_rcp_f32 (1 ulp -> ~RN(1/zz)), then per quotient
    // q = P*r1; rem = P - q*zz (exact, fused); q' = q + rem*r1 (fused).  9 operations instead of the
    // ~22 of two full division expansions; bit-equality with IEEE division is asserted by the
    // generic kernel's bitwise parity tests (and was checked on 2e7 random operands on the host).
    // A zero/denormal/overflowing denominator yields NaN/inf here and +-inf there: both end up
    // out of the image (make_taps), i.e. the same zero contribution.
    const float r0 = __builtin_amdgcn_rcpf(zz);
    const float e  = __builtin_fmaf(-zz, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    const float qx = Px * r1, qy = Py * r1;
    Px = __builtin_fmaf(__builtin_fmaf(-qx, zz, Px), r1, qx);
    Py = __builtin_fmaf(__builtin_fmaf(-qy, zz, Py), r1, qy);
    zw = pv.tz + pv.rcz * d;
    const float gx = clamp10(div_const(Px - gc.cw, gc.cw, gc.rcw));
    const float gy = clamp10(div_const(Py - gc.ch, gc.ch, gc.rch));
    ix = __builtin_fmaf(gx + 1.0f, gc.sw, -0.5f);
    iy = __builtin_fmaf(gy + 1.0f, gc.sh, -0.5f);
}

// Weights + quad origin; inwin: at least one tap may lie inside the image (x0 in [-1,w-1], y0 in [-1,h-1]).
__device__ __forceinline__ Taps make_taps(float ix, float iy, float fw, float fh, int& qx0, int& qy0, bool& inwin) {
    Taps t;
    const float x0 = __builtin_floorf(ix), y0 = __builtin_floorf(iy);
    const float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    t.nw = (x1 - ix) * (y1 - iy);
    t.ne = (ix - x0) * (y1 - iy);
    t.sw = (x1 - ix) * (iy - y0);
    t.se = (ix - x0) * (iy - y0);
    // floor(ix) in [-1, w-1]  <=>  -1 <= ix < w ; comparisons are false for NaN, and the +-10 clamp
    // bounds |ix|,|iy| by 5.5*size, so the int conversions below cannot overflow when inwin holds
    inwin = (ix >= -1.0f) && (ix < fw) && (iy >= -1.0f) && (iy < fh);
    qx0 = (int)x0;
    qy0 = (int)y0;
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

// round-to-nearest-even fp32 -> bf16 (NaN stays a quiet NaN): gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32,
// one instruction for two values) — the integer add / shift / NaN-select sequence it replaces was ~6 VALU per value
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
typedef __bf16 mg_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float mg_f32x2_t __attribute__((ext_vector_type(2)));
// {bf16(a) | bf16(b) << 16}
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_rne(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(mg_f32x2_t{a, b}, mg_bf16x2_t));
}
// x = hi + lo for two values at once: packed hi words and packed lo words
__device__ __forceinline__ void split_bf16x2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = f32x2_to_bf16x2_rne(a, b);
    lo = f32x2_to_bf16x2_rne(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

}  // namespace magnet
