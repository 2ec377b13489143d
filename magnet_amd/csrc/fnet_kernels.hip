// fnet_kernels.hip — the small layers of the F-Net (PSMNet feature extractor, reference
// models/submodules/F_psmnet.py:37-124) that are not matrix-core convolutions (row N3 of SURVEY.md §8f):
//   stem          3 -> 32 channels, 3x3 stride 2 (+ folded BatchNorm, ReLU), reads the NCHW fp32 image   (:40)
//   space_to_depth  (H/2, 32 ch) -> (H/4, 4 x 32 ch): turns the stride-2 3x3 of layer2 into a 2x2-window
//                 stride-1 implicit GEMM for conv_mfma_kernel (taps = 4)                                  (:45,89-93)
//   avgpool       AvgPool2d(k, k) of the 128-channel map for the four SPP branches                        (:50-64)
//   upsample      bilinear, align_corners=True, back to H/4 into a channel slice of the 320-ch concat     (:108-122)
// Activations are the conv kernel's format: zero-bordered channel-last grids stored as two bf16 planes
// (hi = bf16(x), lo = bf16(x - hi)).  All four are HBM-bound element-wise kernels with 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magnet {

namespace {

__device__ __forceinline__ uint16_t fk_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void fk_split8(const float* v, uint4& hi, uint4& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint16_t h0 = fk_bf16(v[2 * i]), h1 = fk_bf16(v[2 * i + 1]);
        const uint16_t l0 = fk_bf16(v[2 * i] - __uint_as_float((uint32_t)h0 << 16));
        const uint16_t l1 = fk_bf16(v[2 * i + 1] - __uint_as_float((uint32_t)h1 << 16));
        h[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void fk_join8(const uint4 hi, const uint4 lo, float* v) {
    const uint32_t h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i]     = __uint_as_float(h[i] << 16) + __uint_as_float(l[i] << 16);
        v[2 * i + 1] = __uint_as_float(h[i] & 0xffff0000u) + __uint_as_float(l[i] & 0xffff0000u);
    }
}

}  // namespace

// ---- stem: out[n, y, x, co] = relu(b[co] + sum_{ci,dy,dx} W[co][ci][dy][dx] * img[n, ci, 2y+dy-1, 2x+dx-1]) ----
// one thread per output pixel, 32 accumulators; the 864 weights are read with wave-uniform addresses (scalar loads)
__global__ __launch_bounds__(256) void fnet_stem_kernel(const float* __restrict__ img, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, uint16_t* __restrict__ out_hi,
                                                        uint16_t* __restrict__ out_lo, int N, int H, int W, int H2, int W2) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * H2 * W2) return;
    const int x = (int)(idx % W2), y = (int)((idx / W2) % H2), n = (int)(idx / ((long long)W2 * H2));
    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int iy = 2 * y + dy - 1, ix = 2 * x + dx - 1;
                const bool ok = (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
                in[ci * 9 + dy * 3 + dx] = ok ? img[(((size_t)n * 3 + ci) * H + iy) * W + ix] : 0.f;
            }
    const size_t row = ((size_t)n * (H2 + 2) + (y + 1)) * (W2 + 2) + (x + 1);
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int co = c8 * 8 + i;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 27; ++k) acc = __builtin_fmaf(wgt[co * 27 + k], in[k], acc);
            acc += bias[co];
            v[i] = acc < 0.f ? 0.f : acc;
        }
        uint4 hi, lo;
        fk_split8(v, hi, lo);
        *reinterpret_cast<uint4*>(out_hi + row * 32 + c8 * 8) = hi;
        *reinterpret_cast<uint4*>(out_lo + row * 32 + c8 * 8) = lo;
    }
}

// ---- space to depth: out[n, y, x, (py*2+px)*C + c] = in[n, 2y+py, 2x+px, c]; in border 1, out border `opad` ----
__global__ __launch_bounds__(256) void space_to_depth_kernel(const uint16_t* __restrict__ in_hi, const uint16_t* __restrict__ in_lo,
                                                             uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo,
                                                             int N, int C, int H2, int W2, int H4, int W4, int opad) {
    const int cpp = 4 * C / 8;                                  // 16-byte chunks per output pixel
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * H4 * W4 * cpp) return;
    const int cc = (int)(idx % cpp);
    const long long pix = idx / cpp;
    const int x = (int)(pix % W4), y = (int)((pix / W4) % H4), n = (int)(pix / ((long long)W4 * H4));
    const int phase = (cc * 8) / C, c0 = (cc * 8) % C, py = phase >> 1, px = phase & 1;
    const int iy = 2 * y + py, ix = 2 * x + px;
    uint4 hi = make_uint4(0, 0, 0, 0), lo = make_uint4(0, 0, 0, 0);
    if (iy < H2 && ix < W2) {
        const size_t irow = ((size_t)n * (H2 + 2) + (iy + 1)) * (W2 + 2) + (ix + 1);
        hi = *reinterpret_cast<const uint4*>(in_hi + irow * C + c0);
        lo = *reinterpret_cast<const uint4*>(in_lo + irow * C + c0);
    }
    const size_t orow = ((size_t)n * (H4 + 2 * opad) + (y + opad)) * (W4 + 2 * opad) + (x + opad);
    *reinterpret_cast<uint4*>(out_hi + orow * (4 * C) + cc * 8) = hi;
    *reinterpret_cast<uint4*>(out_lo + orow * (4 * C) + cc * 8) = lo;
}

// ---- AvgPool2d(k, stride k), floor mode: in = channel slice (ld elements per row) of a grid with border `pad` ----
// block = one pooled cell x 8 channel groups... : 256 threads = 16 channel chunks (C = 128) x 16 window slices
__global__ __launch_bounds__(256) void avgpool_cl_kernel(const uint16_t* __restrict__ in_hi, const uint16_t* __restrict__ in_lo,
                                                         int ld, int h, int w, int pad, int k, int ph, int pw, int C,
                                                         uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo) {
    __shared__ float part[16][129];
    const int cell = blockIdx.x;
    const int px = cell % pw, py = (cell / pw) % ph, n = cell / (pw * ph);
    const int chunk = threadIdx.x & 15, slice = threadIdx.x >> 4;           // C/8 <= 16 chunks
    const int Wp = w + 2 * pad, Hp = h + 2 * pad;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (chunk * 8 < C) {
        for (int e = slice; e < k * k; e += 16) {
            const int yy = py * k + e / k, xx = px * k + e % k;
            const size_t row = ((size_t)n * Hp + (yy + pad)) * Wp + (xx + pad);
            float v[8];
            fk_join8(*reinterpret_cast<const uint4*>(in_hi + row * ld + chunk * 8),
                     *reinterpret_cast<const uint4*>(in_lo + row * ld + chunk * 8), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[slice][chunk * 8 + i] = acc[i];
    __syncthreads();
    if (threadIdx.x < C / 8) {
        float v[8];
        const float inv = 1.0f / (float)(k * k);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = 0.f;
            for (int sl = 0; sl < 16; ++sl) s += part[sl][threadIdx.x * 8 + i];
            v[i] = s * inv;
        }
        uint4 hi, lo;
        fk_split8(v, hi, lo);
        *reinterpret_cast<uint4*>(out_hi + (size_t)cell * C + threadIdx.x * 8) = hi;
        *reinterpret_cast<uint4*>(out_lo + (size_t)cell * C + threadIdx.x * 8) = lo;
    }
}

// ---- bilinear upsampling, align_corners=True (ATen upsample_bilinear2d): in fp32 (N*ph*pw, in_ld) -> split planes
// ---- into channels [0, C) at `out_*` (pre-offset to the slice), row pitch out_ld, grid border `pad` ----
__global__ __launch_bounds__(256) void upsample_bilinear_cl_kernel(const float* __restrict__ in, int in_ld, int ph, int pw, int C,
                                                                   uint16_t* __restrict__ out_hi, uint16_t* __restrict__ out_lo,
                                                                   int out_ld, int N, int h, int w, int pad) {
    const int cpp = C / 8;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * h * w * cpp) return;
    const int cc = (int)(idx % cpp);
    const long long pix = idx / cpp;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), n = (int)(pix / ((long long)w * h));
    // ATen: scale = (in - 1) / (out - 1) (0 when out == 1); src = scale * dst; i0 = int(src); lambda1 = src - i0
    const float sy = (h > 1) ? (float)(ph - 1) / (float)(h - 1) : 0.f;
    const float sx = (w > 1) ? (float)(pw - 1) / (float)(w - 1) : 0.f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < ph - 1) ? 1 : 0), x1 = x0 + ((x0 < pw - 1) ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1, lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
    const float* b = in + (size_t)n * ph * pw * in_ld + cc * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v00 = b[((size_t)y0 * pw + x0) * in_ld + i], v01 = b[((size_t)y0 * pw + x1) * in_ld + i];
        const float v10 = b[((size_t)y1 * pw + x0) * in_ld + i], v11 = b[((size_t)y1 * pw + x1) * in_ld + i];
        v[i] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
    uint4 hi, lo;
    fk_split8(v, hi, lo);
    const size_t row = ((size_t)n * (h + 2 * pad) + (y + pad)) * (w + 2 * pad) + (x + pad);
    *reinterpret_cast<uint4*>(out_hi + row * out_ld + cc * 8) = hi;
    *reinterpret_cast<uint4*>(out_lo + row * out_ld + cc * 8) = lo;
}

hipError_t launch_fnet_stem(const float* img, const float* wgt, const float* bias, uint16_t* out_hi, uint16_t* out_lo,
                            int N, int H, int W, hipStream_t s) {
    const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1;
    const long long n = (long long)N * H2 * W2;
    hipLaunchKernelGGL(fnet_stem_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, wgt, bias, out_hi, out_lo, N, H, W, H2, W2);
    return hipGetLastError();
}

hipError_t launch_space_to_depth(const uint16_t* in_hi, const uint16_t* in_lo, uint16_t* out_hi, uint16_t* out_lo, int N, int C,
                                 int H2, int W2, int opad, hipStream_t s) {
    const int H4 = (H2 - 1) / 2 + 1, W4 = (W2 - 1) / 2 + 1;
    const long long n = (long long)N * H4 * W4 * (4 * C / 8);
    hipLaunchKernelGGL(space_to_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in_hi, in_lo, out_hi, out_lo, N, C,
                       H2, W2, H4, W4, opad);
    return hipGetLastError();
}

hipError_t launch_avgpool_cl(const uint16_t* in_hi, const uint16_t* in_lo, int ld, int N, int h, int w, int pad, int k, int C,
                             uint16_t* out_hi, uint16_t* out_lo, hipStream_t s) {
    const int ph = h / k, pw = w / k;
    hipLaunchKernelGGL(avgpool_cl_kernel, dim3((unsigned)(N * ph * pw)), dim3(256), 0, s, in_hi, in_lo, ld, h, w, pad, k, ph, pw, C,
                       out_hi, out_lo);
    return hipGetLastError();
}

hipError_t launch_upsample_bilinear_cl(const float* in, int in_ld, int ph, int pw, int C, uint16_t* out_hi, uint16_t* out_lo,
                                       int out_ld, int N, int h, int w, int pad, hipStream_t s) {
    const long long n = (long long)N * h * w * (C / 8);
    hipLaunchKernelGGL(upsample_bilinear_cl_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, in_ld, ph, pw, C, out_hi,
                       out_lo, out_ld, N, h, w, pad);
    return hipGetLastError();
}

}  // namespace magnet
