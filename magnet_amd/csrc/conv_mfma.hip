// conv_mfma.hip — G-Net / mask-head convolutions as implicit GEMM on the bf16 matrix cores with
// bf16x3 operand splitting (row N1 of SURVEY.md §8f; reference: models/MAGNET.py:47-70,111-118).
//
// The reference runs these layers in fp32.  gfx950 has no reduced-precision fp32 matrix path (no xf32)
// and fp32 MFMA runs at the vector rate (157 TF), 1/16 of the bf16 matrix rate.  So every fp32 operand
// is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits kept) and
//      x*w  ~=  hi_x*hi_w + hi_x*lo_w + lo_x*hi_w          (dropped terms < 2^-16 |x w|)
// is accumulated in fp32 by three v_mfma_f32_16x16x32_bf16 per tile: fp32-grade results (measured
// relative error ~1e-5 on these layers, tests/test_gpu_conv.py) at up to 1/3 of the bf16 matrix peak.
//
// GEMM view:  out[p, n] = bias[n] + sum_{tap, c} act[p + off(tap), c] * W[tap][n][c]
//   p   = row index in the ZERO-BORDERED channel-last activation (B, h+2, w+2, C) flattened over
//         (B, h+2, w+2): with the border in place a 3x3 tap is a plain row offset off = dy*(w+2)+dx and
//         the kernel needs no boundary logic at all.  Border rows are computed too (3 % extra work at
//         120x160) and hold garbage that no interior output ever reads; callers read interior rows only.
//   act = two bf16 planes (hi, lo), W = two bf16 planes of [tap][Cout_pad][Cin] (Cin contiguous), so both
//         MFMA operands are "row-major with K contiguous": one 16-byte LDS read per fragment per lane.
// Tiling: workgroup = 128 rows x BN = NF*16 output channels, 4 waves stacked along M (32 rows each, all
// channels), K step 32; global -> registers -> LDS staging with the next step's loads in flight during the
// MFMAs; LDS rows padded to 80 B (conflict-free ds_read_b128); epilogue through LDS: bias, ReLU, then
// either re-split to bf16 hi/lo planes (input of the next layer) or fp32 rows (last layer).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/magnet_hip.h"

namespace magnet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct ConvParams {
    const uint16_t* in_hi;  const uint16_t* in_lo;    // activations, row 0 of the flattened padded grid
    const uint16_t* w_hi;   const uint16_t* w_lo;     // [taps][cout_pad][cin]
    const float*    bias;                             // [cout_pad]
    uint16_t* out_hi; uint16_t* out_lo;               // OUT mode 0: bf16 planes [rows][cout_pad]
    float*    out_f32;                                // OUT mode 1: fp32 [rows][cout_pad]
    long long rows;                                   // B*(h+2)*(w+2)
    int cin, cout_pad, taps, wp, relu, out_mode;
    int in_ld;                                        // elements between consecutive input rows (>= cin)
};

constexpr int CV_BM = 128, CV_BK = 32;
constexpr int CV_LDS_ROW = 80;                        // bytes per staged row: 64 data + 16 pad

__device__ __forceinline__ uint16_t bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float((uint32_t)hi << 16));
}

// NF = 16-column fragments per wave = BN/16 (8: 128 channels, 9: 144, 1: 16)
template <int NF>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    constexpr int BN = NF * 16;
    constexpr int A_BYTES = CV_BM * CV_LDS_ROW, B_BYTES = BN * CV_LDS_ROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* a_hi = smem;
    unsigned char* a_lo = smem + A_BYTES;
    unsigned char* b_hi = smem + 2 * A_BYTES;
    unsigned char* b_lo = smem + 2 * A_BYTES + B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long row0 = (long long)blockIdx.x * CV_BM;
    const int n0 = blockIdx.y * BN;

    // staging roles: a 64-byte K-slice of one row = 4 x 16 B; thread -> (row, quarter)
    constexpr int A_VEC = CV_BM * 4, B_VEC = BN * 4;          // 16-byte vectors per plane
    constexpr int A_PT = A_VEC / 256;                          // = 2
    constexpr int B_PT = (B_VEC + 255) / 256;                  // 2 (BN=128), 3 (BN=144), 1 (BN=16)
    static_assert(A_PT == 2 && B_PT <= 3, "staging registers below are spelled out for these sizes");
    // explicit scalars: arrays here end up in scratch memory (the compiler does not promote them)
    uint4 ah0, ah1, al0, al1, bh0, bh1, bh2, bl0, bl1, bl2;
    bh1 = bh2 = bl1 = bl2 = make_uint4(0, 0, 0, 0);

    const int ksteps_per_tap = p.cin / CV_BK;
    const int nsteps = p.taps * ksteps_per_tap;
    const int st_r = tid >> 2, st_q = tid & 3;                 // staging role: row (+64, +128 for later vectors), quarter

    auto a_elem = [&](int s, int i) -> size_t {
        const int tap = s / ksteps_per_tap, k0 = (s % ksteps_per_tap) * CV_BK;
        const int off = (p.taps == 9) ? ((tap / 3 - 1) * p.wp + (tap % 3 - 1)) : 0;
        long long row = row0 + st_r + i * 64 + off;
        row = row < 0 ? 0 : (row >= p.rows ? p.rows - 1 : row);        // guard rows only feed border outputs
        return (size_t)row * p.in_ld + k0 + st_q * 8;
    };
    auto b_elem = [&](int s, int i) -> size_t {
        const int tap = s / ksteps_per_tap, k0 = (s % ksteps_per_tap) * CV_BK;
        const int r = st_r + i * 64;
        return ((size_t)tap * p.cout_pad + n0 + (r < BN ? r : 0)) * p.cin + k0 + st_q * 8;
    };
#define CV_LD(ptr, e) (*reinterpret_cast<const uint4*>((ptr) + (e)))
#define CV_ISSUE(S)                                                                       \
    {                                                                                     \
        const size_t ea0 = a_elem((S), 0), ea1 = a_elem((S), 1);                          \
        ah0 = CV_LD(p.in_hi, ea0); al0 = CV_LD(p.in_lo, ea0);                             \
        ah1 = CV_LD(p.in_hi, ea1); al1 = CV_LD(p.in_lo, ea1);                             \
        const size_t eb0 = b_elem((S), 0);                                                \
        bh0 = CV_LD(p.w_hi, eb0); bl0 = CV_LD(p.w_lo, eb0);                               \
        if constexpr (B_PT >= 2) { const size_t eb1 = b_elem((S), 1); bh1 = CV_LD(p.w_hi, eb1); bl1 = CV_LD(p.w_lo, eb1); } \
        if constexpr (B_PT >= 3) { const size_t eb2 = b_elem((S), 2); bh2 = CV_LD(p.w_hi, eb2); bl2 = CV_LD(p.w_lo, eb2); } \
    }
#define CV_ST(base, r, v) (*reinterpret_cast<uint4*>((base) + (r) * CV_LDS_ROW + st_q * 16) = (v))
#define CV_COMMIT()                                                                       \
    {                                                                                     \
        CV_ST(a_hi, st_r, ah0); CV_ST(a_lo, st_r, al0);                                   \
        CV_ST(a_hi, st_r + 64, ah1); CV_ST(a_lo, st_r + 64, al1);                         \
        if (st_r < BN) { CV_ST(b_hi, st_r, bh0); CV_ST(b_lo, st_r, bl0); }                \
        if constexpr (B_PT >= 2) if (st_r + 64 < BN) { CV_ST(b_hi, st_r + 64, bh1); CV_ST(b_lo, st_r + 64, bl1); }   \
        if constexpr (B_PT >= 3) if (st_r + 128 < BN) { CV_ST(b_hi, st_r + 128, bh2); CV_ST(b_lo, st_r + 128, bl2); } \
    }

    f32x4_t acc[2][NF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addressing: lane -> (row/col = lane & 15, k group = lane >> 4): 16 B at byte (lane>>4)*16
    const int frow = lane & 15, fk = (lane >> 4) * 16;
    const unsigned char* a_base_hi = a_hi + (wv * 32 + frow) * CV_LDS_ROW + fk;
    const unsigned char* a_base_lo = a_lo + (wv * 32 + frow) * CV_LDS_ROW + fk;
    const unsigned char* b_base_hi = b_hi + frow * CV_LDS_ROW + fk;
    const unsigned char* b_base_lo = b_lo + frow * CV_LDS_ROW + fk;

    CV_ISSUE(0)
    CV_COMMIT()
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) CV_ISSUE(s + 1)                   // next step's global loads fly during the MFMAs
        bf16x8_t ah[2], al[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ah[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_base_hi + m * 16 * CV_LDS_ROW));
            al[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_base_lo + m * 16 * CV_LDS_ROW));
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b_base_hi + n * 16 * CV_LDS_ROW));
            const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(b_base_lo + n * 16 * CV_LDS_ROW));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh, acc[m][n], 0, 0, 0);   // small terms first
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh, acc[m][n], 0, 0, 0);
            }
        }
        __syncthreads();                                      // everyone is done reading this step's tiles
        if (s + 1 < nsteps) {
            CV_COMMIT()
            __syncthreads();
        }
    }

    // ---- epilogue through LDS, one 16-row fragment per wave at a time: [16 rows][BN] fp32 per wave ----
    float* stage = reinterpret_cast<float*>(smem) + wv * (16 * (BN + 4));   // +4 floats row pad
    constexpr int SROW = BN + 4;
#pragma unroll
    for (int m = 0; m < 2; ++m) {                             // fully unrolled: acc indices stay static
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NF; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                stage[((lane >> 4) * 4 + r) * SROW + n * 16 + (lane & 15)] = acc[m][n][r];   // C: col = lane&15, row = (lane>>4)*4+r
        __syncthreads();
        // 16 rows x BN channels, 8 channels (one 16-byte bf16 vector / two fp32 vectors) per work item
        for (int it = lane; it < 16 * (BN / 8); it += 64) {
            const int r = it / (BN / 8), c8 = (it % (BN / 8)) * 8;
            const long long row = row0 + wv * 32 + m * 16 + r;
            if (row >= p.rows) continue;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = stage[r * SROW + c8 + i] + p.bias[n0 + c8 + i];
                v[i] = (p.relu && x < 0.f) ? 0.f : x;
            }
            const size_t e = (size_t)row * p.cout_pad + n0 + c8;
            if (p.out_mode == 0) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint16_t h0, l0, h1, l1;
                    split_bf16(v[2 * i], h0, l0); split_bf16(v[2 * i + 1], h1, l1);
                    h[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                    l[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
                }
                *reinterpret_cast<uint4*>(p.out_hi + e) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4*>(p.out_lo + e) = make_uint4(l[0], l[1], l[2], l[3]);
            } else {
                *reinterpret_cast<float4*>(p.out_f32 + e) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.out_f32 + e + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
}

template <int NF>
static size_t conv_lds_bytes() {
    const size_t tiles = 2 * (size_t)CV_BM * CV_LDS_ROW + 2 * (size_t)(NF * 16) * CV_LDS_ROW;
    const size_t stage = (size_t)4 * 16 * (NF * 16 + 4) * 4;
    return tiles > stage ? tiles : stage;
}

template <int NF>
static hipError_t launch_conv_nf(const ConvParams& p, hipStream_t s) {
    const dim3 grid((unsigned)((p.rows + CV_BM - 1) / CV_BM), (unsigned)(p.cout_pad / (NF * 16))), block(256);
    hipLaunchKernelGGL((conv_mfma_kernel<NF>), grid, block, conv_lds_bytes<NF>(), s, p);
    return hipGetLastError();
}

hipError_t launch_conv_mfma(const ConvParams& p, hipStream_t s) {
    if (p.cout_pad % 128 == 0) return launch_conv_nf<8>(p, s);
    if (p.cout_pad == 144)     return launch_conv_nf<9>(p, s);
    if (p.cout_pad == 16)      return launch_conv_nf<1>(p, s);
    return hipErrorInvalidValue;
}

// ---- fp32 NCHW (N, C, h, w) -> split bf16 planes of the padded channel-last buffer (N, h+2, w+2, Ctot),
// ---- channels [c_off, c_off + C); 64 pixels x 64 channels per block through LDS; interior only
// ---- (the zero border is written once by magnet_zero_fill / hipMemsetAsync when the buffer is created)
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ in, uint16_t* __restrict__ out_hi,
                                                          uint16_t* __restrict__ out_lo, int C, int h, int w,
                                                          int ctot, int c_off, long long in_img_stride) {
    __shared__ float tile[64][65];
    const int hw = h * w;
    const int bpi = (hw + 63) / 64;
    const int n = blockIdx.x / bpi, p0 = (blockIdx.x % bpi) * 64, f0 = blockIdx.y * 64;
    const int tid = threadIdx.x, px = tid & 63, cs = tid >> 6;
    for (int c = cs; c < 64; c += 4) {
        const int f = f0 + c, pp = p0 + px;
        tile[c][px] = (pp < hw && f < C) ? in[(size_t)n * in_img_stride + (size_t)f * hw + pp] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 8; i += 256) {                 // 64 pixels x 8 vectors of 8 channels
        const int q = i >> 3, vc = (i & 7) * 8, pq = p0 + q;
        if (pq >= hw || f0 + vc >= C) continue;
        const int y = pq / w, x = pq % w;
        const size_t row = ((size_t)n * (h + 2) + (y + 1)) * (w + 2) + (x + 1);
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint16_t h0, l0, h1, l1;
            split_bf16(tile[vc + 2 * k][q], h0, l0); split_bf16(tile[vc + 2 * k + 1][q], h1, l1);
            hh[k] = (uint32_t)h0 | ((uint32_t)h1 << 16);
            ll[k] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
        const size_t e = row * ctot + c_off + f0 + vc;
        *reinterpret_cast<uint4*>(out_hi + e) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(out_lo + e) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    }
}

hipError_t launch_pack_split(const float* in, uint16_t* out_hi, uint16_t* out_lo, int N, int C, int h, int w,
                             int ctot, int c_off, long long in_img_stride, hipStream_t s) {
    const int bpi = (h * w + 63) / 64;
    const dim3 grid((unsigned)(N * bpi), (unsigned)((C + 63) / 64)), block(256);
    hipLaunchKernelGGL(pack_split_kernel, grid, block, 0, s, in, out_hi, out_lo, C, h, w, ctot, c_off, in_img_stride);
    return hipGetLastError();
}

#undef CV_ISSUE
#undef CV_COMMIT
#undef CV_LD
#undef CV_ST

}  // namespace magnet
