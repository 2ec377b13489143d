// elementwise.hip — the HBM-bound helpers around the matcher:
//   pack_features  : F-Net's NCHW fp32 -> channel-last fp32/bf16 (the cost-volume kernels' storage)
//   gaussian_update: GNET.forward's tail (models/MAGNET.py:60-69)
//   upsample_depth : upsample_depth_via_mask (models/MAGNET.py:15-27)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "warp_math.hpp"

namespace magnet {

// ---- pack: (N,F,hw) fp32 -> (N,hw,F) OutT, 64 pixels x FT channels per block through LDS ------
// Reads are 256-B rows along the pixel axis, writes are 16-B vectors along the channel axis.
constexpr int PK_PIX = 64;
template <typename OutT, int FT>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ in, OutT* __restrict__ out,
                                                    int F, int hw, int blocks_per_img, int w, int pad) {
    __shared__ float tile[FT][PK_PIX + 1];
    const int n = blockIdx.x / blocks_per_img;
    const int p0 = (blockIdx.x % blocks_per_img) * PK_PIX;
    const int f0 = blockIdx.y * FT;
    const int tid = threadIdx.x, px = tid & 63, cs = tid >> 6;
    const int p = p0 + px;
    for (int c = cs; c < FT; c += 4) {
        const int f = f0 + c;
        tile[c][px] = (p < hw && f < F) ? in[((size_t)n * F + f) * hw + p] : 0.f;
    }
    __syncthreads();
    constexpr int VEC = 16 / sizeof(OutT);           // channels per 16-byte store
    constexpr int VPP = FT / VEC;                    // vectors per pixel
    for (int i = tid; i < PK_PIX * VPP; i += 256) {
        const int q = i / VPP, vc = (i % VPP) * VEC;
        const int pq = p0 + q;
        if (pq >= hw || f0 + vc >= F) continue;
        // destination texel inside the (h+2*pad) x (w+2*pad) image
        const int h_ = hw / w, py = pq / w, px_ = pq % w;
        const size_t tex = (size_t)n * (h_ + 2 * pad) * (w + 2 * pad) + (size_t)(py + pad) * (w + 2 * pad) + (px_ + pad);
        OutT* dst = out + tex * F + f0 + vc;
        if constexpr (sizeof(OutT) == 4) {
            float4 o = make_float4(tile[vc][q], tile[vc + 1][q], tile[vc + 2][q], tile[vc + 3][q]);
            *reinterpret_cast<float4*>(dst) = o;
        } else {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w[k] = f32x2_to_bf16x2_rne(tile[vc + 2 * k][q], tile[vc + 2 * k + 1][q]);
            *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// Wide variant (hw % 4 == 0, F % 8 == 0): 128 pixels x 64 channels per workgroup, 8 independent 16-byte loads per thread, tile
// rows of 132 floats with the pixel index XOR-swizzled in 4-word blocks by the channel octet (float4 tile writes and the
// transposed reads are conflict-free).
constexpr int PKW_PIX = 128, PKW_S = 132;
__device__ __forceinline__ int pkw_idx(int c, int q) { return c * PKW_S + (q ^ (((c >> 3) & 7) << 2)); }
template <typename OutT>
__global__ __launch_bounds__(256) void pack_wide_kernel(const float* __restrict__ in, OutT* __restrict__ out,
                                                         int F, int hw, int blocks_per_img, int w, int pad) {
    __shared__ __attribute__((aligned(16))) float tile[64 * PKW_S];
    const int n = blockIdx.x / blocks_per_img;
    const int p0 = (blockIdx.x % blocks_per_img) * PKW_PIX;
    const int f0 = blockIdx.y * 64;
    const int tid = threadIdx.x, l4 = (tid & 31) * 4, cr = tid >> 5;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int f = f0 + cr + 8 * k, pp = p0 + l4;
        v[k] = (pp < hw && f < F) ? *reinterpret_cast<const float4*>(in + ((size_t)n * F + f) * hw + pp) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(&tile[pkw_idx(cr + 8 * k, l4)]) = v[k];
    __syncthreads();
    const int h_ = hw / w;
    constexpr int VEC = 16 / sizeof(OutT);           // channels per 16-byte store
    constexpr int VPP = 64 / VEC;                    // vectors per pixel (8 or 16)
#pragma unroll
    for (int it = 0; it < PKW_PIX * VPP / 256; ++it) {
        const int i = tid + it * 256;
        const int q = i / VPP, vc = (i % VPP) * VEC, pq = p0 + q;
        if (pq >= hw || f0 + vc >= F) continue;
        const int py = pq / w, px_ = pq - py * w;
        const size_t tex = (size_t)n * (h_ + 2 * pad) * (w + 2 * pad) + (size_t)(py + pad) * (w + 2 * pad) + (px_ + pad);
        OutT* dst = out + tex * F + f0 + vc;
        if constexpr (sizeof(OutT) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(tile[pkw_idx(vc, q)], tile[pkw_idx(vc + 1, q)], tile[pkw_idx(vc + 2, q)], tile[pkw_idx(vc + 3, q)]);
        } else {
            uint32_t wd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                wd[k] = f32x2_to_bf16x2_rne(tile[pkw_idx(vc + 2 * k, q)], tile[pkw_idx(vc + 2 * k + 1, q)]);
            *reinterpret_cast<uint4*>(dst) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        }
    }
}

// zero the one-texel border of (N, h+2, w+2, F) images: 16-byte vectors
__global__ __launch_bounds__(256) void zero_border_kernel(uint4* __restrict__ out, int N, int h, int w, int vec_per_tex) {
    const int Wp = w + 2, Hp = h + 2;
    const int per_img = 2 * Wp + 2 * h;                      // border texels of one image
    const size_t total = (size_t)N * per_img * vec_per_tex;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % vec_per_tex);
        const size_t t = i / vec_per_tex;
        const int n = (int)(t / per_img), k = (int)(t % per_img);
        int y, x;
        if (k < Wp) { y = 0; x = k; }
        else if (k < 2 * Wp) { y = Hp - 1; x = k - Wp; }
        else { const int m = k - 2 * Wp; y = 1 + (m >> 1); x = (m & 1) ? Wp - 1 : 0; }
        out[(((size_t)n * Hp + y) * Wp + x) * vec_per_tex + c] = make_uint4(0, 0, 0, 0);
    }
}

// (N,2,h,w) planar [mu,sigma] -> (N,h+2,w+2,2) interleaved, zero border (one thread per padded texel)
__global__ __launch_bounds__(256) void pack_gmm_kernel(const float* __restrict__ in, float2* __restrict__ out,
                                                        int N, int h, int w) {
    const int Wp = w + 2, Hp = h + 2;
    const size_t total = (size_t)N * Hp * Wp;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wp) - 1, y = (int)((i / Wp) % Hp) - 1;
    const size_t n = i / ((size_t)Wp * Hp);
    float2 v = make_float2(0.f, 0.f);
    if (x >= 0 && x < w && y >= 0 && y < h) {
        const size_t hw = (size_t)h * w, o = (size_t)y * w + x;
        v = make_float2(in[(n * 2 + 0) * hw + o], in[(n * 2 + 1) * hw + o]);
    }
    out[i] = v;
}

hipError_t launch_pack_gmm(const float* in, float* out, int N, int h, int w, hipStream_t s) {
    const size_t total = (size_t)N * (h + 2) * (w + 2);
    hipLaunchKernelGGL(pack_gmm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in,
                       reinterpret_cast<float2*>(out), N, h, w);
    return hipGetLastError();
}

// (N,2,h,w) planar [mu,sigma] -> (N,h+2,w+2,8): the zero-bordered map per quad origin in quad form (one thread per origin):
// v(bx,by) = v00 + bx*(v10-v00) + by*(v01-v00) + bx*by*((v11-v01)-(v10-v00))  — 3 fma per bilinear sample in the matcher
__global__ __launch_bounds__(256) void pack_gmm_quad_kernel(const float* __restrict__ in, float4* __restrict__ out,
                                                             int N, int h, int w) {
    const int Wp = w + 2, Hp = h + 2;
    const size_t total = (size_t)N * Hp * Wp;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x0 = (int)(i % Wp) - 1, y0 = (int)((i / Wp) % Hp) - 1;       // unpadded coordinates of the quad's top-left texel
    const size_t n = i / ((size_t)Wp * Hp);
    const size_t hw = (size_t)h * w;
    float m[4], s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int x = x0 + (t & 1), y = y0 + (t >> 1);
        const bool in_img = x >= 0 && x < w && y >= 0 && y < h;
        const size_t o = (size_t)(in_img ? y : 0) * w + (in_img ? x : 0);
        m[t] = in_img ? in[(n * 2 + 0) * hw + o] : 0.f;
        s[t] = in_img ? in[(n * 2 + 1) * hw + o] : 0.f;
    }
    out[i * 2 + 0] = make_float4(m[0], m[1] - m[0], m[2] - m[0], (m[3] - m[2]) - (m[1] - m[0]));
    out[i * 2 + 1] = make_float4(s[0], s[1] - s[0], s[2] - s[0], (s[3] - s[2]) - (s[1] - s[0]));
}

hipError_t launch_pack_gmm_quad(const float* in, float* out, int N, int h, int w, hipStream_t s) {
    const size_t total = (size_t)N * (h + 2) * (w + 2);
    hipLaunchKernelGGL(pack_gmm_quad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in,
                       reinterpret_cast<float4*>(out), N, h, w);
    return hipGetLastError();
}

hipError_t launch_pack(const float* in, void* out, int N, int F, int h, int w, bool bf16, int pad, hipStream_t s) {
    const int hw = h * w;
    const int bpi = (hw + PK_PIX - 1) / PK_PIX;
    constexpr int FT = 64;
    const dim3 grid((unsigned)(N * bpi), (unsigned)((F + FT - 1) / FT)), block(256);
    if (pad) {
        const int vec_per_tex = F * (bf16 ? 2 : 4) / 16;
        const size_t total = (size_t)N * (2 * (w + 2) + 2 * h) * vec_per_tex;
        const unsigned nb = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(zero_border_kernel, dim3(nb), dim3(256), 0, s, reinterpret_cast<uint4*>(out), N, h, w, vec_per_tex);
    }
#ifdef MAGNET_DEV
    static const bool narrow = getenv("MAGNET_PACK_NARROW") != nullptr;      // dev A/B: the 64-pixel kernel
#else
    constexpr bool narrow = false;
#endif
    if (!narrow && hw % 4 == 0 && F % 8 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const int bpw = (hw + PKW_PIX - 1) / PKW_PIX;
        const dim3 gw((unsigned)(N * bpw), (unsigned)((F + 63) / 64));
        if (bf16) hipLaunchKernelGGL((pack_wide_kernel<uint16_t>), gw, block, 0, s, in, (uint16_t*)out, F, hw, bpw, w, pad);
        else      hipLaunchKernelGGL((pack_wide_kernel<float>),    gw, block, 0, s, in, (float*)out, F, hw, bpw, w, pad);
        return hipGetLastError();
    }
    if (bf16) hipLaunchKernelGGL((pack_kernel<uint16_t, FT>), grid, block, 0, s, in, (uint16_t*)out, F, hw, bpi, w, pad);
    else      hipLaunchKernelGGL((pack_kernel<float, FT>),    grid, block, 0, s, in, (float*)out, F, hw, bpi, w, pad);
    return hipGetLastError();
}

// ---- gaussian update ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_update_kernel(const float* __restrict__ o,
                                                               const float* __restrict__ g,
                                                               float* __restrict__ out, int B, int hw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * hw) return;
    const size_t b = i / hw, p = i % hw;
    const float mu0 = g[(b * 2 + 0) * hw + p], sg0 = g[(b * 2 + 1) * hw + p];
    const float o0 = o[(b * 2 + 0) * hw + p], o1 = o[(b * 2 + 1) * hw + p];
    const float mu1 = mu0 + (o0 * sg0);                                 // MAGNET.py:67
    const float e = (o1 > 0.f) ? o1 : expm1f(o1);                       // F.elu
    const float sg1 = ((e + 1.0f) + 1e-10f) * sg0;                      // MAGNET.py:68
    out[(b * 2 + 0) * hw + p] = mu1;
    out[(b * 2 + 1) * hw + p] = sg1;
}

hipError_t launch_gaussian_update(const float* o, const float* g, float* out, int B, int hw, hipStream_t s) {
    const size_t n = (size_t)B * hw;
    hipLaunchKernelGGL(gaussian_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o, g, out, B, hw);
    return hipGetLastError();
}

// gaussian update reading the conv kernel's padded channel-last fp32 output (B, h+2, w+2, ld)
__global__ __launch_bounds__(256) void gaussian_update_cl_kernel(const float* __restrict__ o, int ld,
                                                                  const float* __restrict__ g, float* __restrict__ out,
                                                                  int B, int h, int w) {
    const size_t hw = (size_t)h * w;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * hw) return;
    const size_t b = i / hw, pp = i % hw;
    const int y = (int)(pp / w), x = (int)(pp % w);
    const float2 ov = *reinterpret_cast<const float2*>(o + (((size_t)b * (h + 2) + (y + 1)) * (w + 2) + (x + 1)) * ld);
    const float mu0 = g[(b * 2 + 0) * hw + pp], sg0 = g[(b * 2 + 1) * hw + pp];
    const float mu1 = mu0 + (ov.x * sg0);
    const float e = (ov.y > 0.f) ? ov.y : expm1f(ov.y);
    out[(b * 2 + 0) * hw + pp] = mu1;
    out[(b * 2 + 1) * hw + pp] = ((e + 1.0f) + 1e-10f) * sg0;
}

hipError_t launch_gaussian_update_cl(const float* o, int ld, const float* g, float* out, int B, int h, int w, hipStream_t s) {
    const size_t n = (size_t)B * h * w;
    hipLaunchKernelGGL(gaussian_update_cl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o, ld, g, out, B, h, w);
    return hipGetLastError();
}

// convex x4 upsampling with the mask in padded channel-last fp32 (B, h+2, w+2, ld): thread = (pixel, sub-row i);
// the 4 sub-columns j of one neighbour n are 16 contiguous bytes (channel n*16 + i*4 + j, MAGNET.py:19)
// n_pred stacked (mu, sigma) maps (n_pred, B, 2, h, w) share one read of the mask and one softmax: the refinement loop returns
// every iteration's prediction upsampled (MAGNET.py:173), and the mask (B, 144, h, w as fp32) is 4.6x the bytes of one output
__global__ __launch_bounds__(256) void upsample_cl_kernel(const float* __restrict__ depth, const float* __restrict__ mask,
                                                           int ld, float* __restrict__ out, int B, int h, int w, int n_pred) {
    const size_t hw = (size_t)h * w;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)B * hw * 4) return;
    const int i = (int)(t & 3);
    const size_t pi = t >> 2, b = pi / hw, pp = pi % hw;
    const int y = (int)(pp / w), x = (int)(pp % w);
    const float* m = mask + (((size_t)b * (h + 2) + (y + 1)) * (w + 2) + (x + 1)) * ld + i * 4;
    float4 mv[9];
    float4 mx = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        mv[n] = *reinterpret_cast<const float4*>(m + n * 16);
        mx.x = fmaxf(mx.x, mv[n].x); mx.y = fmaxf(mx.y, mv[n].y); mx.z = fmaxf(mx.z, mv[n].z); mx.w = fmaxf(mx.w, mv[n].w);
    }
    float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < 9; ++n) {
        mv[n].x = __expf(mv[n].x - mx.x); mv[n].y = __expf(mv[n].y - mx.y);
        mv[n].z = __expf(mv[n].z - mx.z); mv[n].w = __expf(mv[n].w - mx.w);
        den.x += mv[n].x; den.y += mv[n].y; den.z += mv[n].z; den.w += mv[n].w;
    }
    const float4 inv = make_float4(1.0f / den.x, 1.0f / den.y, 1.0f / den.z, 1.0f / den.w);
#pragma unroll
    for (int n = 0; n < 9; ++n) { mv[n].x *= inv.x; mv[n].y *= inv.y; mv[n].z *= inv.z; mv[n].w *= inv.w; }
    for (int pi = 0; pi < n_pred; ++pi) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const size_t plane = ((size_t)pi * B + b) * 2 + c;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                const int yy = y + n / 3 - 1, xx = x + n % 3 - 1;
                const float dv = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? depth[plane * hw + (size_t)yy * w + xx] : 0.f;
                a.x += mv[n].x * dv; a.y += mv[n].y * dv;
                a.z += mv[n].z * dv; a.w += mv[n].w * dv;
            }
            *reinterpret_cast<float4*>(out + (plane * h * 4 + (size_t)y * 4 + i) * ((size_t)w * 4) + (size_t)x * 4) = a;
        }
    }
}

hipError_t launch_upsample_cl(const float* d, const float* m, int ld, float* o, int B, int h, int w, int n_pred, hipStream_t s) {
    const size_t n = (size_t)B * h * w * 4;
    hipLaunchKernelGGL(upsample_cl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, m, ld, o, B, h, w, n_pred);
    return hipGetLastError();
}

// ---- depth metrics (utils.compute_depth_errors, utils/utils.py:106-144 + the masking of test_MaGNet.py:43,58-79) ----
// Per frame: sums over valid pixels (min < gt < max) of the terms each metric averages; fp64 accumulation
// (thread -> wave shuffle tree -> workgroup LDS -> per-workgroup partials -> fixed-order final sum; no atomics).
// sums[b][0..15] = n, |d|, |d|/gt, d^2/gt, d^2, (ln gt - ln p)^2, (ln p - ln gt), |log10 gt - log10 p|,
//                  (1/gt - 1/p)^2, [t<1.25], [t<1.25^2], [t<1.25^3], nll term, 0, 0, 0      with d = gt - p, t = max(gt/p, p/gt)
constexpr int MET_N = 16;
constexpr int MET_PARTS = 64;      // workgroups per frame in the first stage (fixed: the summation order never depends on the launch)
// Two stages, both in a fixed order, so results are run-to-run deterministic without atomics: stage 1 = MET_PARTS workgroups
// of 256 threads per frame, each over a fixed interleaved slice of the frame (per-thread strided sums -> wave shuffle tree
// -> 4 wave partials added in order) writing 13 partial sums; stage 2 = one wave per frame adding the MET_PARTS partials in
// index order.  Optional evaluation window [cy0,cy1) x [cx0,cx1) = the reference's rectangular garg / eigen crops
// (test_MaGNet.py:63-71); cy1 < 0 = whole frame.
__global__ __launch_bounds__(256) void depth_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                             double* __restrict__ part, int HW, int W, float dmin, float dmax,
                                                             int cy0, int cy1, int cx0, int cx1) {
    const int b = blockIdx.y;
    double acc[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) acc[i] = 0.0;
    const float* mu = pred + (size_t)b * 2 * HW;
    const float* sg = mu + HW;
    const float* g = gt + (size_t)b * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += MET_PARTS * 256) {
        float gv = g[i];
        if (gv > dmax) gv = 0.f;                                       // test_MaGNet.py:43
        if (!(gv > dmin && gv < dmax)) continue;                       // test_MaGNet.py:58
        if (cy1 >= 0) {                                                // test_MaGNet.py:63-71
            const int yy = i / W, xx = i - yy * W;
            if (yy < cy0 || yy >= cy1 || xx < cx0 || xx >= cx1) continue;
        }
        float pv = mu[i];
        pv = pv < dmin ? dmin : pv; pv = pv > dmax ? dmax : pv;        // test_MaGNet.py:72-75 (inf -> max, nan -> min)
        if (isinf(pv)) pv = dmax;
        if (isnan(pv)) pv = dmin;
        const double G = gv, P = pv, d = G - P;
        const double lg = log(G), lp = log(P);
        const double t = fmax(G / P, P / G);
        double var = (double)sg[i] * (double)sg[i];
        var = var < 1e-6 ? 1e-6 : var;                                 // utils.py:134
        acc[0] += 1.0; acc[1] += fabs(d); acc[2] += fabs(d) / G; acc[3] += d * d / G; acc[4] += d * d;
        acc[5] += (lg - lp) * (lg - lp); acc[6] += (lp - lg); acc[7] += fabs(log10(G) - log10(P));
        acc[8] += (1.0 / G - 1.0 / P) * (1.0 / G - 1.0 / P);
        acc[9] += (t < 1.25) ? 1.0 : 0.0; acc[10] += (t < 1.25 * 1.25) ? 1.0 : 0.0; acc[11] += (t < 1.25 * 1.25 * 1.25) ? 1.0 : 0.0;
        acc[12] += 0.5 * (log(var) + 1.8378770664093453 + d * d / var);    // ln(2 pi)
    }
    __shared__ double red[4][13];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 13)
        part[((size_t)b * MET_PARTS + blockIdx.x) * 13 + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void depth_metrics_final_kernel(const double* __restrict__ part, double* __restrict__ sums) {
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= MET_N) return;
    double v = 0.0;
    if (k < 13)
        for (int q = 0; q < MET_PARTS; ++q) v += part[((size_t)b * MET_PARTS + q) * 13 + k];
    sums[(size_t)b * MET_N + k] = v;
}

// the partial sums live in a stream-ordered allocation (hipMallocAsync / hipFreeAsync): no state kept between calls, safe
// with concurrent calls on different streams
hipError_t launch_depth_metrics(const float* pred, const float* gt, double* sums, int B, int HW, int W, float dmin, float dmax,
                                int cy0, int cy1, int cx0, int cx1, hipStream_t s) {
    double* part = nullptr;
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&part), (size_t)B * MET_PARTS * 13 * sizeof(double), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(depth_metrics_kernel, dim3(MET_PARTS, (unsigned)B), dim3(256), 0, s, pred, gt, part, HW, W, dmin, dmax,
                       cy0, cy1, cx0, cx1);
    hipLaunchKernelGGL(depth_metrics_final_kernel, dim3((unsigned)B), dim3(64), 0, s, part, sums);
    e = hipGetLastError();
    const hipError_t e2 = hipFreeAsync(part, s);
    return e != hipSuccess ? e : e2;
}

// ---- learned convex upsampling -------------------------------------------------------------------
// One thread per coarse pixel; lanes run along x so every mask-plane read is coalesced; the k
// sub-pixels of one output row are stored as one contiguous run per lane (16 B for k = 4).
template <int K, int C>
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ depth,
                                                        const float* __restrict__ mask,
                                                        float* __restrict__ out, int B, int h, int w) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (x >= w || y >= h) return;
    const size_t hw = (size_t)h * w;
    float nb[C][9];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            const int yy = y + n / 3 - 1, xx = x + n % 3 - 1;          // F.unfold(depth,[3,3],padding=1)
            nb[c][n] = (yy >= 0 && yy < h && xx >= 0 && xx < w)
                           ? depth[((size_t)b * C + c) * hw + (size_t)yy * w + xx] : 0.f;
        }
    const float* m = mask + (size_t)b * 9 * K * K * hw + (size_t)y * w + x;
    const size_t W2 = (size_t)w * K;
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
        float o[C][K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float mv[9], mx = -3.4e38f;
#pragma unroll
            for (int n = 0; n < 9; ++n) { mv[n] = m[(size_t)(n * K * K + i * K + j) * hw]; mx = fmaxf(mx, mv[n]); }
            float den = 0.f;
#pragma unroll
            for (int n = 0; n < 9; ++n) { mv[n] = __expf(mv[n] - mx); den += mv[n]; }
            const float inv = 1.0f / den;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float a = 0.f;
#pragma unroll
                for (int n = 0; n < 9; ++n) a += (mv[n] * inv) * nb[c][n];
                o[c][j] = a;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float* dst = out + (((size_t)b * C + c) * h * K + (size_t)y * K + i) * W2 + (size_t)x * K;
            if constexpr (K == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
            else {
#pragma unroll
                for (int j = 0; j < K; ++j) dst[j] = o[c][j];
            }
        }
    }
}

template <int K>
static hipError_t launch_up_k(const float* d, const float* m, float* o, int B, int C, int h, int w, hipStream_t s) {
    const dim3 grid((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)B), block(256);
    if (C == 2)      hipLaunchKernelGGL((upsample_kernel<K, 2>), grid, block, 0, s, d, m, o, B, h, w);
    else if (C == 1) hipLaunchKernelGGL((upsample_kernel<K, 1>), grid, block, 0, s, d, m, o, B, h, w);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_upsample(const float* d, const float* m, float* o, int B, int C, int h, int w, int k, hipStream_t s) {
    switch (k) {
        case 1: return launch_up_k<1>(d, m, o, B, C, h, w, s);
        case 2: return launch_up_k<2>(d, m, o, B, C, h, w, s);
        case 4: return launch_up_k<4>(d, m, o, B, C, h, w, s);
        case 8: return launch_up_k<8>(d, m, o, B, C, h, w, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- camera intrinsics / relative poses on the device (row N4 of SURVEY.md §8f) ----------------------------------------------
// unit_ray_array_2D (B,3,h*w) from (fx, fy, cx, cy, sx, sy, left, top) per frame: the loaders' float64 expression
// ((x + 0.5) * sx - cx + left) / fx cast to fp32 once (dataloader_scannet.py:139-147, dataloader_kitti.py:113-118,
// dataloader_7scenes.py:100-108): bit-identical to the host table.
__global__ __launch_bounds__(256) void make_rays_kernel(const double* __restrict__ prm, float* __restrict__ rays, int B, int h, int w) {
    const size_t hw = (size_t)h * w;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= hw * (size_t)B) return;
    const int b = (int)(i / hw);
    const size_t pix = i % hw;
    const int y = (int)(pix / w), x = (int)(pix % w);
    const double* q = prm + (size_t)b * 8;
    rays[((size_t)b * 3 + 0) * hw + pix] = (float)(((((double)x + 0.5) * q[4]) - q[2] + q[6]) / q[0]);
    rays[((size_t)b * 3 + 1) * hw + pix] = (float)(((((double)y + 0.5) * q[5]) - q[3] + q[7]) / q[1]);
    rays[((size_t)b * 3 + 2) * hw + pix] = 1.0f;
}

hipError_t launch_make_rays(const double* prm, float* rays, int B, int h, int w, hipStream_t s) {
    const size_t n = (size_t)B * h * w;
    hipLaunchKernelGGL(make_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prm, rays, B, h, w);
    return hipGetLastError();
}

// utils.data_preprocess (utils/utils.py:72-98): pose[b,v] = ext_nghbr[b,v] @ inv(ext_ref[b]) in float64, stored fp32; a NaN anywhere
// in either matrix (ScanNet's lost poses) or a singular reference -> is_valid = 0 and a zero pose.  One thread per (b, v):
// Gauss-Jordan with partial pivoting on the 4x4 reference (float64; agrees with LAPACK's inverse to a few ulp of float64,
// i.e. identical after the cast to fp32 except for rare 1-ulp cases — test tolerance 1e-6).
__global__ void relative_poses_kernel(const double* __restrict__ ext_ref, const double* __restrict__ ext_ngh,
                                      float* __restrict__ poses, int32_t* __restrict__ valid, int B, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * V) return;
    const int b = i / V;
    double a[4][8];
    bool bad = false;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = ext_ref[(size_t)b * 16 + r * 4 + c];
            a[r][4 + c] = (r == c) ? 1.0 : 0.0;
            bad |= (a[r][c] != a[r][c]);
        }
    double n[16];
    for (int e = 0; e < 16; ++e) { n[e] = ext_ngh[(size_t)i * 16 + e]; bad |= (n[e] != n[e]); }
    if (!bad) {
        for (int c = 0; c < 4 && !bad; ++c) {
            int piv = c;
            for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
            if (!(fabs(a[piv][c]) > 0.0)) { bad = true; break; }
            if (piv != c) for (int k = 0; k < 8; ++k) { const double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
            const double inv = 1.0 / a[c][c];
            for (int k = 0; k < 8; ++k) a[c][k] *= inv;
            for (int r = 0; r < 4; ++r) {
                if (r == c) continue;
                const double f = a[r][c];
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
            }
        }
    }
    float out[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += n[r * 4 + k] * a[k][4 + c];
            bad |= (acc != acc);
            out[r * 4 + c] = (float)acc;
        }
    for (int e = 0; e < 16; ++e) poses[(size_t)i * 16 + e] = bad ? 0.0f : out[e];
    valid[i] = bad ? 0 : 1;
}

hipError_t launch_relative_poses(const double* ext_ref, const double* ext_ngh, float* poses, int32_t* valid, int B, int V, hipStream_t s) {
    hipLaunchKernelGGL(relative_poses_kernel, dim3((unsigned)((B * V + 63) / 64)), dim3(64), 0, s, ext_ref, ext_ngh, poses, valid, B, V);
    return hipGetLastError();
}

}  // namespace magnet
