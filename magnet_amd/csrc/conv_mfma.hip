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
// Tiling: workgroup = 128 rows x BN = NF*16 output channels, 2x2 waves of 64x64 (BN = 128) or 4 waves stacked
// along M (BN = 144, 16), K step 32; global -> registers -> LDS staging with the next step's loads in flight during the
// MFMAs into the other half of a double-buffered, XOR-swizzled (conflict-free) LDS image: one barrier per K
// step; epilogue through LDS: bias, ReLU, then either re-split to bf16 hi/lo planes (input of the next layer)
// or fp32 rows (last layer).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/magnet_hip.h"
#include "conv_common.hpp"
#include "warp_math.hpp"

namespace magnet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

[[maybe_unused]] constexpr int CV_BK = 32;
constexpr int CV_ROW = 64;                            // bytes per staged row (32 bf16), unpadded

// LDS image of a [rows][32 bf16] tile: 64-byte rows, the four 16-byte slots of a row XOR-swizzled with
// ((row >> 1) & 3).  Conflict-free for both access patterns (checked against the ds_read_b128 /
// ds_write_b128 lane groups of MI355X_MICROARCH.md §LDS): MFMA fragment reads (lane -> row = lane&15,
// slot = lane>>4) and the staging writes (thread -> row = t>>2, slot = t&3).  The first version used
// 80-byte padded rows: SQ_LDS_BANK_CONFLICT was 50 % of SQ_LDS_IDX_ACTIVE.
__device__ __forceinline__ int cv_swz(int row, int slot) { return row * CV_ROW + ((slot ^ ((row >> 1) & 3)) << 4); }

// The same image for the 32x32x16 fragment pattern (lane -> row = lane & 31, K slot = 2 * half + (lane >> 5)): the ds_read_b128 lane
// groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (and their upper-half twins) then hold, per residue of row mod 4, four rows that differ
// in bits 2..3 (0/12/20/24, 4/8/16/28, ... — also after a shift by a tap offset), so XOR-ing the slot with (row >> 2) & 3 spreads them
// over the four 16-byte slots of the 64-bank window; (row >> 1) & 3 above would put rows 0 and 8 on the same banks.
__device__ __forceinline__ int cv_swz32(int row, int slot) { return row * CV_ROW + ((slot ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ uint16_t bf16_rne(float f) { return f32_to_bf16_rne(f); }     // v_cvt_pk_bf16_f32 (warp_math.hpp)
__device__ __forceinline__ void split_bf16(float x, uint16_t& hi, uint16_t& lo) {
    hi = bf16_rne(x);
    lo = bf16_rne(x - __uint_as_float((uint32_t)hi << 16));
}

// NF = 16-column fragments of the workgroup tile (BN = NF*16: 128, 144 or 16 channels);
// WN = waves along N (2: 2x2 waves; 1: 4x1 waves); CV_BM = rows of the workgroup tile (128 or 256)
// SPB = K stages per barrier interval (the LDS ring holds 2*SPB stages)
__device__ __forceinline__ int act_swz(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }

// One 1x1 layer of the fused tail: this wave's 32 tile rows x (NF*16) output channels over K = 128, activations read
// from the LDS tile (act_swz layout), weight fragments straight from global memory (64 KB per layer, L2-resident; no LDS
// staging, so no barrier: the rows are wave-private).  Operands are swapped (weights = MFMA A operand): the accumulator is
// C^T — a lane holds 4 CONSECUTIVE output channels of one row.
// Arguments of the fused convex upsampling (ConvParams::up_*), passed by value to the row-owned last tail layer
struct UpArgs { const float* depth; float* out; int npred, h, w, B; };   // Gaussian update: depth = (mu, sigma) in, out = (mu, sigma) out, npred = -1

template <int NF, bool LAST, int MT = 2>
__device__ __forceinline__ void tail_layer(const uint16_t* __restrict__ w_hi, const uint16_t* __restrict__ w_lo,
                                           const float* __restrict__ bias, unsigned char* act_hi, unsigned char* act_lo,
                                           float* __restrict__ out, int out_ld, long long row0, long long rows, int lane, int wv) {
    f32x4_t acc[NF][MT];
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, kslot = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8_t xh[MT], xl[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = wv * (MT * 16) + m * 16 + frow;
            xh[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_hi + act_swz(row, kk * 4 + kslot)));
            xl[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_lo + act_swz(row, kk * 4 + kslot)));
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const size_t e = (size_t)(n * 16 + frow) * 128 + kk * 32 + kslot * 8;
            const bf16x8_t wh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w_hi + e));
            const bf16x8_t wl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w_lo + e));
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[m], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[m], acc[n][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[m], acc[n][m], 0, 0, 0);
        }
    }
    // all of this wave's reads of its rows are done (same wave, in-order LDS); publish the layer's output in place
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ch = n * 16 + (lane >> 4) * 4;
            const int trow = wv * (MT * 16) + m * 16 + (lane & 15);
            const float4 b4 = *reinterpret_cast<const float4*>(bias + ch);
            float v[4] = {acc[n][m][0] + b4.x, acc[n][m][1] + b4.y, acc[n][m][2] + b4.z, acc[n][m][3] + b4.w};
            if constexpr (!LAST) {
                uint32_t h01, l01, h23, l23;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
                split_bf16x2(v[0], v[1], h01, l01); split_bf16x2(v[2], v[3], h23, l23);
                const int off = act_swz(trow, ch >> 3) + (ch & 7) * 2;          // 8 bytes: channels ch..ch+3
                *reinterpret_cast<uint2*>(act_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(act_lo + off) = make_uint2(l01, l23);
            } else {
                const long long row = row0 + trow;
                if (row < rows) *reinterpret_cast<float4*>(out + (size_t)row * out_ld + ch) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Column-owned form of the same layer (default): wave w computes output fragments n = w, w + 4 (32 of the 128 output channels)
// for ALL rows of the tile, so it fetches a quarter of the layer's weights (16 KB instead of 64 KB per wave: with row ownership the
// two resident workgroups pull 128 KB per K chunk through the CU's 64 B/clk vector-memory path for 1 536 cycles of MFMA work — the
// tails ran at half the K loop's efficiency) and reads every row's activations from LDS (256 B/clk, cheap).  A remainder
// fragment (TAILN % 4 == 1: the 16-channel G-Net head, the 9th fragment of the 144-channel mask head) is row-split over the
// waves.  The layer's output replaces its input in place: one barrier after the last read, one after the last write.
// (Round 4 measured a 2 (row halves) x 4 (column groups) ownership for the 8-wave form — half the fragment reads from LDS, twice the
// weights through the vector-memory path, bit-identical results: 2.07 - 2.09 / 1.90 - 1.93 ms against 2.07 - 2.08 / 1.88 - 1.89 ms for
// this form on the two stacks, same box (profiles/r4/conv_tail_ownership_ab.log).  Not kept: the tails, like the K loop, do not speed
// up by moving load between pipes.)
template <int TAILN, bool LAST, int ROWS, int NW = 4, bool UP = false>
__device__ __forceinline__ void tail_layer_cols(const uint16_t* __restrict__ w_hi, const uint16_t* __restrict__ w_lo,
                                                const float* __restrict__ bias, unsigned char* act_hi, unsigned char* act_lo,
                                                float* __restrict__ out, int out_ld, long long row0, long long rows, int lane, int wv,
                                                const UpArgs up = UpArgs{nullptr, nullptr, 0, 0, 0, 0}) {
    // NW waves: wave w owns output fragments w, w + NW, ...; MA row blocks, read from LDS 8 at a time
    constexpr int NJ = TAILN / NW, REM = TAILN % NW, MA = ROWS / 16, MR = MA / NW;   // MR: remainder-fragment row blocks per wave
    static_assert(REM <= 1 && MA % 8 == 0 && MA % NW == 0, "one row-split remainder fragment at most");
    f32x4_t acc[NJ > 0 ? NJ : 1][MA], accr[MR];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int m = 0; m < MA; ++m) acc[j][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MR; ++m) accr[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, kslot = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if constexpr (NJ > 0) {
            bf16x8_t wh[NJ], wl[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const size_t e = (size_t)((wv + NW * j) * 16 + frow) * 128 + kk * 32 + kslot * 8;
                wh[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w_hi + e));
                wl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w_lo + e));
            }
#pragma unroll
            for (int mb = 0; mb < MA; mb += 8) {
                bf16x8_t xh[8], xl[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    xh[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_hi + act_swz((mb + m) * 16 + frow, kk * 4 + kslot)));
                    xl[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_lo + act_swz((mb + m) * 16 + frow, kk * 4 + kslot)));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc[j][mb + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xl[m], acc[j][mb + m], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc[j][mb + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], xh[m], acc[j][mb + m], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 8; ++m) acc[j][mb + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh[m], acc[j][mb + m], 0, 0, 0);
                }
            }
        }
        if constexpr (REM) {
            const size_t e = (size_t)(NJ * NW * 16 + frow) * 128 + kk * 32 + kslot * 8;
            const bf16x8_t wh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w_hi + e));
            const bf16x8_t wl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w_lo + e));
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int row = (wv * MR + m) * 16 + frow;
                const bf16x8_t rh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_hi + act_swz(row, kk * 4 + kslot)));
                const bf16x8_t rl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_lo + act_swz(row, kk * 4 + kslot)));
                accr[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, rl, accr[m], 0, 0, 0);
                accr[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, rh, accr[m], 0, 0, 0);
                accr[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, rh, accr[m], 0, 0, 0);
            }
        }
    }
    if constexpr (UP) {
        // ---- learned convex upsampling (models/MAGNET.py:15-27) behind the mask head's last layer, column-owned: the 144 logits of a
        // position sit in 9 different waves (fragment n = neighbour n of the 3x3 window), so they meet in LDS: 128 rows x 144 fp32 at
        // a time in the (now dead) activation tile, then one thread per (position, sub-pixel row i) does what upsample_cl_kernel does
        // — same arithmetic, same order: bit-identical — with the mask read from LDS instead of HBM.
        static_assert(TAILN == 9 && LAST && ROWS % 128 == 0, "mask head's last layer");
        constexpr int SP = 148;                               // stage row pitch in floats: 16 rows x 4 quads of a fragment store hit distinct banks
        float* stage = reinterpret_cast<float*>(act_hi);      // 128 x 148 x 4 B = 74 KB <= ROWS x 512 B
        const int wp = up.w + 2, img_rows = (up.h + 2) * wp;
        const size_t hw = (size_t)up.h * up.w;
        const int tid = wv * 64 + lane;
        for (int half = 0; half < ROWS / 128; ++half) {
            __syncthreads();                                  // every wave is done with the activation tile / with the previous half's stage
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int m = 0; m < MA; ++m) {
                    if (m / 8 != half) continue;
                    const int ch = (wv + NW * j) * 16 + (lane >> 4) * 4, srow = (m & 7) * 16 + (lane & 15);
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + ch);
                    *reinterpret_cast<float4*>(stage + srow * SP + ch) =
                        make_float4(acc[j][m][0] + b4.x, acc[j][m][1] + b4.y, acc[j][m][2] + b4.z, acc[j][m][3] + b4.w);
                }
            if constexpr (REM) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int mrow = wv * MR + m;
                    if (mrow / 8 != half) continue;
                    const int ch = NJ * NW * 16 + (lane >> 4) * 4, srow = (mrow & 7) * 16 + (lane & 15);
                    const float4 b4 = *reinterpret_cast<const float4*>(bias + ch);
                    *reinterpret_cast<float4*>(stage + srow * SP + ch) =
                        make_float4(accr[m][0] + b4.x, accr[m][1] + b4.y, accr[m][2] + b4.z, accr[m][3] + b4.w);
                }
            }
            __syncthreads();
            for (int t = tid; t < 128 * 4; t += NW * 64) {    // (position of this half, sub-pixel row i)
                const int srow = t >> 2, i = t & 3;
                const long long row = row0 + half * 128 + srow;
                const int b = (int)((unsigned)row / (unsigned)img_rows);                  // rows < 2^31 (checked by the API)
                const int rem = (int)((unsigned)row - (unsigned)b * (unsigned)img_rows);
                const int yy = rem / wp, xx = rem - yy * wp;
                if (row >= rows || yy < 1 || yy > up.h || xx < 1 || xx > up.w) continue;  // border position: nothing to write
                const int y = yy - 1, x = xx - 1;
                const float* mrow = stage + srow * SP + i * 4;
                float4 mv[9];
                float4 mx = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
#pragma unroll
                for (int n = 0; n < 9; ++n) {
                    mv[n] = *reinterpret_cast<const float4*>(mrow + n * 16);
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
                for (int pi = 0; pi < up.npred; ++pi) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const size_t plane = ((size_t)pi * up.B + b) * 2 + c;
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int n = 0; n < 9; ++n) {
                            const int y2 = y + n / 3 - 1, x2 = x + n % 3 - 1;
                            const float dv = (y2 >= 0 && y2 < up.h && x2 >= 0 && x2 < up.w) ? up.depth[plane * hw + (size_t)y2 * up.w + x2] : 0.f;
                            a.x += mv[n].x * dv; a.y += mv[n].y * dv; a.z += mv[n].z * dv; a.w += mv[n].w * dv;
                        }
                        *reinterpret_cast<float4*>(up.out + (plane * up.h * 4 + (size_t)y * 4 + i) * ((size_t)up.w * 4) + (size_t)x * 4) = a;
                    }
                }
            }
        }
        return;
    }
    if constexpr (!LAST) __syncthreads();                     // every wave is done reading the layer's input: overwrite it in place
    auto emit = [&](const f32x4_t& a, int n, int mrow) {
        const int ch = n * 16 + (lane >> 4) * 4;
        const int trow = mrow * 16 + (lane & 15);
        const float4 b4 = *reinterpret_cast<const float4*>(bias + ch);
        float v[4] = {a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w};
        if constexpr (!LAST) {
            uint32_t h01, l01, h23, l23;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
            split_bf16x2(v[0], v[1], h01, l01); split_bf16x2(v[2], v[3], h23, l23);
            const int off = act_swz(trow, ch >> 3) + (ch & 7) * 2;              // 8 bytes: channels ch..ch+3
            *reinterpret_cast<uint2*>(act_hi + off) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(act_lo + off) = make_uint2(l01, l23);
        } else {
            const long long row = row0 + trow;
            if constexpr (TAILN == 1) {
                if (up.npred < 0) {
                    // fused Gaussian update (models/MAGNET.py:60-69; gaussian_update_cl_kernel's arithmetic): channels 0, 1 of the head =
                    // the first quad's lanes; interior positions only
                    if (ch != 0 || row >= rows) return;
                    const int wp = up.w + 2, img_rows = (up.h + 2) * wp;
                    const int b = (int)((unsigned)row / (unsigned)img_rows);
                    const int rem = (int)((unsigned)row - (unsigned)b * (unsigned)img_rows);
                    const int yy = rem / wp, xx = rem - yy * wp;
                    if (yy < 1 || yy > up.h || xx < 1 || xx > up.w) return;
                    const size_t hw = (size_t)up.h * up.w, pp = (size_t)(yy - 1) * up.w + (xx - 1);
                    const float mu0 = up.depth[((size_t)b * 2 + 0) * hw + pp], sg0 = up.depth[((size_t)b * 2 + 1) * hw + pp];
                    const float mu1 = mu0 + (v[0] * sg0);
                    const float e = (v[1] > 0.f) ? v[1] : expm1f(v[1]);
                    up.out[((size_t)b * 2 + 0) * hw + pp] = mu1;
                    up.out[((size_t)b * 2 + 1) * hw + pp] = ((e + 1.0f) + 1e-10f) * sg0;
                    return;
                }
            }
            if (row < rows) *reinterpret_cast<float4*>(out + (size_t)row * out_ld + ch) = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int m = 0; m < MA; ++m) emit(acc[j][m], wv + NW * j, m);
    if constexpr (REM) {
#pragma unroll
        for (int m = 0; m < MR; ++m) emit(accr[m], NJ * NW, wv * MR + m);
    }
    if constexpr (!LAST) __syncthreads();
}

// One MFMA of the K loop.  SWAP (kernels with a fused tail): operands exchanged, so the accumulator holds C^T — a lane has 4
// consecutive output CHANNELS of one row, and the tail's activation tile is written with packed conversions and 8-byte LDS stores
// (the plain orientation gives 4 consecutive rows of one channel: 128 two-byte stores per lane and tile).
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// 32x32x16 form, operands swapped as in the fused-tail kernels (weights = A operand): C^T block [32 channels][32 tile rows]; lane l,
// register i: channel (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), tile row l & 31 (cdna_hip_programming.md, fragment layout)
__device__ __forceinline__ f32x16_t cv_mma32(const bf16x8_t& act, const bf16x8_t& wgt, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(wgt, act, c, 0, 0, 0);
}

template <bool SWAP>
__device__ __forceinline__ f32x4_t cv_mma(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
    if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// TAIL = 0: plain layer.  TAIL = 16-column fragments of the fused tail's last layer (1, 8 or 9): see ConvParams::tail_*.
// PP = "ping-pong" K loop: NT = 512 threads = 8 waves = two wave groups (waves 0-3 / 4-7: one wave of each group per SIMD) that
// run the same program half a K step apart — while one group issues its fragment reads and the LDS-DMA of the stage two steps
// ahead, the other owns the matrix pipe — over a 3-slot LDS ring with counted vmcnt waits and raw s_barriers (no DMA drain at a
// barrier).  The 4-wave / 2-slot loop (PP = false) drains the DMA queue (vmcnt(0)) at every step's __syncthreads and relies on a
// second workgroup per CU to fill the gap: 41 % of the matrix pipe; see DESIGN.md §4.3.
// WIN = row-window K loop: the taps of one tap ROW (ty fixed, tx = 0..tap_n-1) read the same activation rows shifted by tap_sx, so
// their A operand is staged ONCE per (K chunk, ty) as a window of CV_BM + (tap_n-1)*tap_sx rows and the tx sub-steps read their
// fragments at a row offset; only the weight tile changes per sub-step.  L2 requests per 3x3 tap row: 272 + 3*256 instead of
// 3*(256 + 256).  WIN = 1: two window slots + two weight slots in LDS, one flat loop over the sub-steps (2x2 taps, dev A/B).
// WIN = 2 (3x3 layers, default): the window's fragments for all three tx are held in registers, one window slot + a 3-slot weight
// ring with counted vmcnt waits (prefetch distance 2).
// __launch_bounds__(NT, 2): two waves per SIMD is what the LDS budget allows anyway, and with <= 256 registers hipcc selects the
// VGPR form of the MFMAs — with the default bound it kept the accumulators in AGPRs and moved all 64 of them through VGPRs
// (64 v_accvgpr_read + 64 v_accvgpr_write) on every trip of the K loop.
// One output tile (CV_BM rows x BN channels).  `bid` of `n_tiles`: the tile's position in launch order (the block id of a one-tile-per-
// workgroup launch, the loop counter of a persistent one); `by`: the channel block.
// M32_ (round 5, DEV ONLY — MAGNET_CONV_VARIANT=16384): the fused-tail kernels' register-window loops (WIN == 2, with and without
// ping-pong) on v_mfma_f32_32x32x16_bf16 — a wave's 64 x 64 tile as 2 x 2 blocks of 32 x 32 instead of 4 x 4 of 16 x 16: the same
// MACs, fragment reads, registers and (by the guide's lane-group model) conflict-free LDS reads with the cv_swz32 image.  The bet: the
// kernel is power-limited and the guide's microbenchmark floors are 2 382 TF for the 32x32 shape against 2 075 TF for 16x16.  Parity
// green (tests/test_gpu_conv.py), measured 5.5 % SLOWER on both stacks (2.135 vs 2.02 ms per 3x3 launch, profiles/r5/conv_m32_ab.log):
// the 32x32x16 shape halves the operand reads per MAC but doubles the fp32 accumulator traffic (K = 16 per instruction): 0.625 against
// 0.5 register bytes per MAC.  Kept as a record; the product library does not instantiate it.
template <int NF, int WN, int CV_BM, int SPB, int TAIL, int NT, bool PP, int WIN, bool M32_ = false>
__device__ __forceinline__ void conv_mfma_tile(const ConvParams& p, const unsigned bid, const unsigned n_tiles, const unsigned by, const int tid) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-descriptor builtins do not exist in the host pass (which only needs the stub)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // ring of 2*SPB stages (declared here, not passed in: a pointer
                                                                            // parameter is a generic pointer, and its casts to LDS pointers get null checks)
    constexpr int BN = NF * 16;
    constexpr bool M32 = M32_ && TAIL > 0 && NF == 8 && WN == 2 && WIN == 2;
    constexpr int RP = NT / 4;                        // tile rows filled by one DMA instruction per wave set (4 lanes per 64-byte row)
    constexpr int A_PT = CV_BM / RP;                           // 16-byte vectors per thread per A plane (2 or 4)
    constexpr int WM = (NT / 64) / WN;                // waves along M
    constexpr int MF = CV_BM / (WM * 16);             // M fragments per wave (2 or 4)
    constexpr int NFW = NF / WN;                      // N fragments per wave
    static_assert(NF % WN == 0, "N fragments must split evenly over the waves");
    constexpr int AW_ROWS = WIN ? CV_BM + 8 : CV_BM;           // window: up to (3-1)*2 extra rows (dilation 2), padded to 8
    constexpr int A_BYTES = AW_ROWS * CV_ROW, B_BYTES = BN * CV_ROW;
    constexpr int STAGE_BYTES = 2 * A_BYTES + 2 * B_BYTES;    // hi + lo planes of A and B
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = wv / WN, wn = wv % WN;
    // XCD-aware tile order: hardware deals consecutive block ids round-robin to the 8 XCDs; remap so each XCD (one
    // private L2) owns a contiguous run of row tiles — vertically adjacent tiles share their halo rows (bijective).
    long long tile_id;
    {
        const unsigned n = n_tiles, q = n / 8, r = n % 8, xcd = bid % 8, idx = bid / 8;
        tile_id = (long long)((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long row0 = tile_id * CV_BM;
    const int n0 = by * BN;

    // staging by LDS-DMA (global_load_lds_dwordx4: global -> LDS without a VGPR round trip or ds_write): a
    // 64-byte K-slice of one row = 4 x 16 B; thread -> (row = tid>>2 (+64, +128 ...), physical slot = tid&3).  The DMA
    // writes lane-linearly (wave-uniform base + lane*16), so the XOR swizzle is applied to the SOURCE: the lane
    // that fills physical slot ps of row r fetches logical K slot ps ^ ((r>>1)&3)  (cdna_hip_programming.md rule 21).
    constexpr int B_PT = (BN + RP - 1) / RP;                   // 16-byte vectors per thread per B plane
    const int ksteps_per_tap = p.cin / CV_BK;
    const int nsteps = p.taps * ksteps_per_tap;
    const int st_r = tid >> 2, st_q = tid & 3;
    const int st_k = (st_q ^ (M32 ? ((st_r >> 2) & 3) : ((st_r >> 1) & 3))) * 8;   // logical K offset (elements) this lane fetches (cv_swz / cv_swz32)
    const int wave_row = __builtin_amdgcn_readfirstlane(wv * 16);   // first tile row this wave's DMA instruction fills

    // step order: K-chunk outer, tap inner — the 9 taps of one 32-channel chunk re-read (shifted) the same 64-byte
    // row slices back to back, a working set of ~50 KB per workgroup that stays in the XCD's L2; tap-major order
    // swept 164 KB per workgroup between re-reads (x64 resident workgroups >> 4 MiB L2).
    //
    // Addressing: buffer descriptors (wave-uniform SGPRs) + one 32-bit per-lane byte offset.  The activation descriptor
    // covers exactly the row window this workgroup can touch, [row0 + min tap offset, row0 + BM + max tap offset) cut to
    // [0, rows): a row outside it is out of range for the hardware bounds check and reads as zero — no per-lane
    // clamping, no 64-bit per-lane arithmetic (the first version spent 137 VALU + 85 SALU instructions per K step
    // next to 48 MFMAs, mostly on this; such rows only ever feed border / guard outputs).  Per step and DMA piece:
    // one v_add of a scalar.
    static_assert(SPB == 1, "the step counters below assume one K stage per barrier interval");
    long long base_row = row0 + p.min_off;  base_row = base_row < 0 ? 0 : base_row;
    long long end_row = row0 + CV_BM + p.max_off;  end_row = end_row > p.rows ? p.rows : end_row;
    const uint32_t flags = 0x00020000u;
    const __amdgpu_buffer_rsrc_t ra_hi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in_hi + base_row * p.in_ld), 0,
                                                                           (int)((end_row - base_row) * p.in_ld * 2), flags);
    const __amdgpu_buffer_rsrc_t ra_lo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in_lo + base_row * p.in_ld), 0,
                                                                           (int)((end_row - base_row) * p.in_ld * 2), flags);
    const int w_bytes = p.taps * p.cout_pad * p.cin * 2;
    const __amdgpu_buffer_rsrc_t rb_hi = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_hi, 0, w_bytes, flags);
    const __amdgpu_buffer_rsrc_t rb_lo = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_lo, 0, w_bytes, flags);
    int a_v[A_PT], b_v[B_PT];                                  // per-lane byte offsets, constant over the K loop
#pragma unroll
    for (int i = 0; i < A_PT; ++i) a_v[i] = (((int)(row0 - base_row) + st_r + i * RP) * p.in_ld + st_k) * 2;
#pragma unroll
    for (int i = 0; i < B_PT; ++i) b_v[i] = ((n0 + st_r + i * RP) * p.cin + st_k) * 2;
    int pf_tap = 0, pf_ty = 0, pf_tx = 0, pf_k0 = 0;           // (tap = ty*n+tx, first channel) of the next stage to fetch
    typedef void __attribute__((address_space(3)))* lptr_t;
#define CV_BLDS(rsrc, lp, voff) __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lptr_t)(lp), 16, (voff), 0, 0, 0)
#define CV_DMA(S, BUF)                                                                                 \
    {                                                                                                  \
        unsigned char* sa_hi = smem + (BUF) * STAGE_BYTES;                                             \
        unsigned char* sa_lo = sa_hi + A_BYTES;                                                        \
        unsigned char* sb_hi = sa_hi + 2 * A_BYTES;                                                    \
        unsigned char* sb_lo = sb_hi + B_BYTES;                                                        \
        const int a_u = (((pf_ty + p.tap_o0) * p.tap_sy + (pf_tx + p.tap_o0) * p.tap_sx) * p.in_ld + pf_k0) * 2;    \
        const int b_u = (pf_tap * p.cout_pad * p.cin + pf_k0) * 2;                                     \
        _Pragma("unroll") for (int i = 0; i < A_PT; ++i) {                                             \
            CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);                        \
            CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);                        \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < B_PT; ++i) {                                             \
            if (st_r + i * RP < BN) {                                                                  \
                CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);                    \
                CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);                    \
            }                                                                                          \
        }                                                                                              \
        ++pf_tap;                                                                                      \
        if (++pf_tx == p.tap_n) { pf_tx = 0; if (++pf_ty == p.tap_n) { pf_ty = 0; pf_tap = 0; pf_k0 += CV_BK; } }   \
    }

    f32x4_t acc[MF][NFW];                                      // (dead in the M32 instances)
    f32x16_t acc32[M32 ? 2 : 1][M32 ? 2 : 1];                  // M32: [row block of 32][channel block of 32] of the wave's 64 x 64 tile
#pragma unroll
    for (int m = 0; m < MF; ++m)
#pragma unroll
        for (int n = 0; n < NFW; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < (M32 ? 2 : 1); ++m)
#pragma unroll
        for (int n = 0; n < (M32 ? 2 : 1); ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc32[m][n][i] = 0.f;
    // fused-tail kernels (C^T accumulators: 4 consecutive channels of one row per lane): the loop-invariant partial sums of the
    // hoisted G-Net layer (ConvParams::addend) are the accumulators' INITIAL value — sixteen 16-byte loads whose latency hides
    // behind the K loop's prologue instead of sitting exposed in the epilogue (the short K = 288 / 576 per-iteration layers)
    if constexpr (M32) {
        if (p.addend) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const long long row = row0 + wm * 64 + mb * 32 + (lane & 31);
                        const int ch = wn * 64 + nb * 32 + g4 * 8 + (lane >> 5) * 4;
                        const float4 a4 = *reinterpret_cast<const float4*>(p.addend + (size_t)(row < p.rows ? row : p.rows - 1) * p.addend_ld + ch);
                        acc32[mb][nb][g4 * 4 + 0] = a4.x; acc32[mb][nb][g4 * 4 + 1] = a4.y; acc32[mb][nb][g4 * 4 + 2] = a4.z; acc32[mb][nb][g4 * 4 + 3] = a4.w;
                    }
        }
    } else if constexpr (TAIL > 0) {
        if (p.addend) {
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int n = 0; n < NFW; ++n) {
                    const long long row = row0 + wm * (MF * 16) + m * 16 + (lane & 15);
                    const int ch = wn * (NFW * 16) + n * 16 + (lane >> 4) * 4;
                    const float4 a4 = *reinterpret_cast<const float4*>(p.addend + (size_t)(row < p.rows ? row : p.rows - 1) * p.addend_ld + ch);
                    acc[m][n] = f32x4_t{a4.x, a4.y, a4.z, a4.w};
                }
        }
    }

    // fragment addressing: lane -> (row/col = lane & 15, K slot = lane >> 4); every fragment's first row is a
    // multiple of 16, so the swizzle term depends on the lane only
    const int frow = lane & 15;
    const int a_off = cv_swz(wm * (MF * 16) + frow, lane >> 4);       // + m*16*CV_ROW
    const int b_off = cv_swz(wn * (NFW * 16) + frow, lane >> 4);      // + n*16*CV_ROW

    // one K stage of MFMAs from ring slot BUF
    auto compute = [&](int buf) {
        const unsigned char* sa_hi = smem + buf * STAGE_BYTES;
        const unsigned char* sa_lo = sa_hi + A_BYTES;
        const unsigned char* sb_hi = sa_hi + 2 * A_BYTES;
        const unsigned char* sb_lo = sb_hi + B_BYTES;
        bf16x8_t ah[MF], al[MF];
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            ah[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_hi + a_off + m * 16 * CV_ROW));
            al[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_lo + a_off + m * 16 * CV_ROW));
        }
#pragma unroll
        for (int n = 0; n < NFW; ++n) {
            const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
            const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off + n * 16 * CV_ROW));
            // term-major order: consecutive MFMAs write DIFFERENT accumulators (a dependent MFMA on the same
            // accumulator waits out the full pipeline latency); small terms first
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(al[m], bh, acc[m][n]);
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ah[m], bl, acc[m][n]);
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ah[m], bh, acc[m][n]);
        }
    };

    if constexpr (WIN == 3 && PP) {
        // ---- ping-pong x LDS window x fragment double-buffering (dev, MAGNET_CONV_VARIANT=128): the fragments of sub-step s+1 are read
        // at the START of the compute phase of sub-step s into a second register set (in the shadow of the 48 MFMAs), so a LOAD phase
        // is only DMA issue + counted wait + barrier.  That needs every stage visible one phase earlier: 4-slot weight ring (DMA
        // distance 3), 2-slot window filled at tx = 0 of the previous group.  With the groups staggered by one barrier:
        //   L_s: [tx = 0: DMA window g+1 -> slot (g+1)&1]  DMA weights s+3 -> slot (s+3)&3;  vmcnt(pieces of THIS phase);  barrier
        //   C_s: ds_read fragments of sub-step s+1 (window g' = (s+1)/3, weights slot (s+1)&3);  48 MFMAs of sub-step s;  lgkmcnt(0);  barrier
        // RAW: what L_{s-1}'s waits retired (weights s+1, window parts issued at or before L_{s-2}) has passed both groups' waits and
        // a barrier when C_s starts.  WAR: slot (s+3)&3 held stage s-1, read at the start of C_{s-2} and retired before the barrier
        // that ends it; the window slot of g+1 held window g-1, last read at the start of C_{3g-2}.
        static_assert(NT == 512 && SPB == 1 && BN % RP == 0 && CV_BM % RP == 0, "8 waves, whole DMA passes");
        constexpr int BP = 2 * B_PT, AP = 2 * A_PT;
        unsigned char* const a_ring = smem;                   // 2 x [hi | lo]
        unsigned char* const b_ring = smem + 4 * A_BYTES;     // 4 x [hi | lo]
        const int aw = CV_BM + 2 * p.tap_sx;
        const int ngroups = 3 * ksteps_per_tap;               // even (checked by the launcher)
        const int nsub = 3 * ngroups;
        int g_ty = 0, g_k0 = 0, bs_tap = 0, bs_k0 = 0;
        auto dma_a = [&](int slot) {
            unsigned char* sa_hi = a_ring + slot * (2 * A_BYTES);
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) {
                CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
                CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
            }
            if (wv == 0 && st_r < aw - CV_BM) {
                const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
            }
            if (++g_ty == 3) { g_ty = 0; g_k0 += CV_BK; }
        };
        auto dma_b = [&](int slot) {
            unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            unsigned char* sb_lo = sb_hi + B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
#pragma unroll
            for (int i = 0; i < B_PT; ++i) {
                CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
            }
            if (++bs_tap == 9) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, lane >> 4);
        bf16x8_t xah0[MF], xal0[MF], xbh0[NFW], xbl0[NFW], xah1[MF], xal1[MF], xbh1[NFW], xbl1[NFW];
        auto read_frags = [&](bf16x8_t (&fa_h)[MF], bf16x8_t (&fa_l)[MF], bf16x8_t (&fb_h)[NFW], bf16x8_t (&fb_l)[NFW], int aslot, int a_of, int bslot) {
            const unsigned char* sa_hi = a_ring + aslot * (2 * A_BYTES);
            const unsigned char* sb_hi = b_ring + bslot * (2 * B_BYTES);
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                fb_h[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
                fb_l[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + B_BYTES + b_off + n * 16 * CV_ROW));
            }
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                fa_h[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_hi + a_of + m * 16 * CV_ROW));
                fa_l[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_hi + A_BYTES + a_of + m * 16 * CV_ROW));
            }
        };
        int sidx = 0;                                         // sub-step counter
        // one sub-step: TX static, (c*) = fragments of this sub-step, (n*) = register set filled for the next one
        auto substep = [&](auto TXc, int gslot, bool lastg, bf16x8_t (&ca_h)[MF], bf16x8_t (&ca_l)[MF], bf16x8_t (&cb_h)[NFW], bf16x8_t (&cb_l)[NFW],
                           bf16x8_t (&na_h)[MF], bf16x8_t (&na_l)[MF], bf16x8_t (&nb_h)[NFW], bf16x8_t (&nb_l)[NFW]) {
            constexpr int TX = decltype(TXc)::value;
            // ---- LOAD phase ----
            asm volatile("" ::: "memory");
            if (!lastg) {
                dma_b((sidx + 3) & 3);
                if constexpr (TX == 0) {
                    dma_a(gslot ^ 1);
                    if (wv == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BP + AP + 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BP + AP) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BP) : "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            // ---- COMPUTE phase ----
            asm volatile("" ::: "memory");
            if (sidx + 1 < nsub)
                read_frags(na_h, na_l, nb_h, nb_l, TX == 2 ? (gslot ^ 1) : gslot, a_offx[(TX + 1) % 3], (sidx + 1) & 3);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ca_l[m], cb_h[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ca_h[m], cb_l[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ca_h[m], cb_h[n], acc[m][n]);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ++sidx;
        };
        const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
        dma_a(0);
        dma_b(0); dma_b(1); dma_b(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_frags(xah0, xal0, xbh0, xbl0, 0, a_offx[0], 0);
        if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier (= half a sub-step) behind group 0
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        for (int g = 0; g < ngroups; g += 2) {
            const bool last2 = g + 2 == ngroups;
            substep(I0{}, 0, false, xah0, xal0, xbh0, xbl0, xah1, xal1, xbh1, xbl1);
            substep(I1{}, 0, false, xah1, xal1, xbh1, xbl1, xah0, xal0, xbh0, xbl0);
            substep(I2{}, 0, false, xah0, xal0, xbh0, xbl0, xah1, xal1, xbh1, xbl1);
            substep(I0{}, 1, last2, xah1, xal1, xbh1, xbl1, xah0, xal0, xbh0, xbl0);
            substep(I1{}, 1, last2, xah0, xal0, xbh0, xbl0, xah1, xal1, xbh1, xbl1);
            substep(I2{}, 1, last2, xah1, xal1, xbh1, xbl1, xah0, xal0, xbh0, xbl0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();           // balance group 1's extra barrier
        __syncthreads();                                      // nothing in flight; the ring is dead
    } else
    if constexpr (WIN == 1 && PP) {
        // ---- ping-pong x LDS window (dev, MAGNET_CONV_VARIANT=64): as the register-window ping-pong loop below, but the window has
        // two LDS slots (one workgroup per CU: 117 KB), so its DMA is spread over tx = 0 / 1 and every LOAD phase is the same 16
        // fragment reads + 2 - 4 DMA pieces.  Phase order per group: L_s = reads of stage
        // s, DMA of stage s+2 (and of window g+1 at tx = 1), counted vmcnt for everything older, lgkmcnt(0), barrier; C_s = MFMAs,
        // barrier.  RAW: a stage is read one L phase after every wave's wait for it and a barrier both groups passed; WAR: a
        // slot is refilled in the L phase after the one whose reads of it retired before a barrier both groups passed.
        static_assert(NT == 512 && SPB == 1 && BN % RP == 0 && CV_BM % RP == 0, "8 waves, whole DMA passes");
        constexpr int BP = 2 * B_PT, AP = 2 * A_PT;
        unsigned char* const a_ring = smem;                   // 2 x [hi | lo]
        unsigned char* const b_ring = smem + 4 * A_BYTES;
        const int aw = CV_BM + 2 * p.tap_sx;
        const int ngroups = 3 * ksteps_per_tap;
        int g_ty = 0, g_k0 = 0, bs_tap = 0, bs_k0 = 0;
        // window g + 1 -> slot (g + 1) & 1 in two parts: part 0 = first row pass + the rows past the tile, part 1 = second row pass
        auto dma_a = [&](int slot, int part) {
            unsigned char* sa_hi = a_ring + slot * (2 * A_BYTES);
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
            CV_BLDS(ra_hi, sa_hi + (wave_row + part * RP) * CV_ROW, a_v[part] + a_u);
            CV_BLDS(ra_lo, sa_lo + (wave_row + part * RP) * CV_ROW, a_v[part] + a_u);
            if (part == 0) {
                if (wv == 0 && st_r < aw - CV_BM) {
                    const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                    CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                    CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
                }
            } else if (++g_ty == 3) { g_ty = 0; g_k0 += CV_BK; }
        };
        auto dma_b = [&](int slot) {
            unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            unsigned char* sb_lo = sb_hi + B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
#pragma unroll
            for (int i = 0; i < B_PT; ++i) {
                CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
            }
            if (++bs_tap == 9) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, lane >> 4);
        static_assert(A_PT == 2, "two window row passes");
        bf16x8_t ah[MF], al[MF], fbh[NFW], fbl[NFW];
        auto load_a = [&](int slot, int a_of) {
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                ah[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_ring + slot * (2 * A_BYTES) + a_of + m * 16 * CV_ROW));
                al[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_ring + slot * (2 * A_BYTES) + A_BYTES + a_of + m * 16 * CV_ROW));
            }
        };
        auto load_b = [&](int slot) {
            const unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            const unsigned char* sb_lo = sb_hi + B_BYTES;
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                fbh[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
                fbl[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off + n * 16 * CV_ROW));
            }
        };
        auto mfmas = [&](const bf16x8_t (&xh)[MF], const bf16x8_t (&xl)[MF]) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xl[m], fbh[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], fbl[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], fbh[n], acc[m][n]);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
#define CV_END_LOAD(N)                                                                         \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");                    \
        __builtin_amdgcn_s_barrier();
        const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
        dma_a(0, 0); dma_a(0, 1);
        dma_b(0);
        dma_b(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BP) : "memory");                 // window 0 and stage 0 landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier (= half a sub-step) behind group 0
        for (int g = 0; g < ngroups; ++g) {
            const bool lastg = g + 1 == ngroups;
            const int as = g & 1;
            // ---- tx = 0 ----
            asm volatile("" ::: "memory");
            load_a(as, a_offx[0]);
            load_b(0);
            dma_b(2);                                         // stage 3g + 2
            if (!lastg) {
                dma_a(as ^ 1, 0);                             // window g + 1, first part (its slot was last read in group g - 1)
                if (wv == 0) { CV_END_LOAD(BP + 4) } else { CV_END_LOAD(BP + 2) }            // stage 3g + 1 landed
            } else { CV_END_LOAD(BP) }
            mfmas(ah, al);
            // ---- tx = 1 ----
            asm volatile("" ::: "memory");
            load_a(as, a_offx[1]);
            load_b(1);
            if (!lastg) { dma_b(0); dma_a(as ^ 1, 1); CV_END_LOAD(BP + 2) }                  // stage 3g + 3, window part 2; stage 3g + 2 landed
            else { CV_END_LOAD(0) }
            mfmas(ah, al);
            // ---- tx = 2 ----
            asm volatile("" ::: "memory");
            load_a(as, a_offx[2]);
            load_b(2);
            if (!lastg) { dma_b(1); CV_END_LOAD(BP) }         // stage 3g + 4; stage 3g + 3 and window g + 1 landed
            else { CV_END_LOAD(0) }
            mfmas(ah, al);
        }
#undef CV_END_LOAD
        if (grp == 0) __builtin_amdgcn_s_barrier();           // balance group 1's extra barrier
        __syncthreads();                                      // nothing in flight; the ring is dead
    } else
    if constexpr (WIN == 4 && PP) {
        // ---- round 4: the "2-unit" operand split on the 8-wave ping-pong loop (ConvParams::in_sc != nullptr) ----
        // x = hi + lo with hi = fp16(x): the main term hi_x * hi_w runs on v_mfma_f32_16x16x32_f16 (1 matrix-pipe unit per 32 K instead of
        // 3 bf16 ones); the two correction terms lo_x * hi_w + hi_x * lo_w are 2^-12 of the result, so OCP e4m3 operands with one E8M0
        // scale per (row, 32 channels) suffice: v_mfma_scale_f32_16x16x128_f8f6f4.  That instruction contracts 128 K, but its four
        // 32-K blocks sit in different lane groups (lane = (row l & 15, group g = l >> 4): bytes 0..15 = K 16 g + t, bytes 16..31 =
        // K 64 + 16 g + t; tools/ubench/mx_split.hip), i.e. each block can come from its own LDS row: block tx = the 32 channels of tap
        // (ty, tx), block 3 = zeros.  ONE scaled MFMA per term and accumulator covers the three taps of a tap row, and the K loop keeps
        // its 32-channel chunks: the "lo" plane of the activation / weight buffers holds, per 64-byte (row, chunk) slice, the 32 e4m3
        // bytes of hi ("q") and the 32 of lo ("r"), staged by the same DMA pieces as the bf16 lo plane was.
        // Per group (chunk, ty) and accumulator: 3 f16 MFMAs + 2 scaled MFMAs (~127 matrix-pipe cycles) instead of 9 bf16 ones (~175).
        // Registers decide the schedule here (64 accumulators + 8-register e4m3 operands): the window is DOUBLE-buffered in LDS, so every
        // sub-step reads only the fragments it multiplies — tx = 0: f16; tx = 1: f16 + term lo_x * hi_w; tx = 2: f16 + term hi_x * lo_w.
        static_assert(NT == 512 && SPB == 1 && BN % RP == 0 && CV_BM % RP == 0 && B_PT == 1, "8 waves, whole DMA passes");
        constexpr int AP = 2 * A_PT;
        constexpr int QB_ZERO = 3 * B_BYTES + 512 * 4;                    // 16 zero bytes behind the scales: block 3 of the weight operand
        constexpr int QB_BYTES = QB_ZERO + 16;
        constexpr int AW_BYTES = 2 * A_BYTES + 512 * 4;                   // [f16 | qr] window + the scales of its rows [512] u32
        unsigned char* const a_base = smem;                               // 2 x window
        unsigned char* const b_ring = smem + 2 * AW_BYTES;                // 4 x f16 weight tile (prefetch distance 3 sub-steps)
        unsigned char* const qb_base = b_ring + 4 * B_BYTES;              // 2 x {3 taps x qr weight tile, weight scales [512] u32, zeros}
        const int aw = CV_BM + 2 * p.tap_sx;
        const int ngroups = 3 * ksteps_per_tap;
        int g_ty = 0, g_k0 = 0, bs_tap = 0, bs_k0 = 0, q_ty = 0, q_k0 = 0;
        // buffers of the scales: activations [chunk][rows] u32 {E8M0 of q, E8M0 of r, 0, 0}; weights [tap][chunk][cout_pad] u32
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in_sc + base_row), 0,
                                                                              (int)(((long long)(p.cin / 32 - 1) * p.sc_rows + (end_row - base_row)) * 4), flags);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_sc, 0, p.taps * (p.cin / 32) * p.cout_pad * 4, flags);
        auto dma_a = [&](int buf) {                                       // window of group g_ty/g_k0: f16 plane, qr plane, scales
            unsigned char* sa_hi = a_base + buf * AW_BYTES;
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) {
                CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
                CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
            }
            if (wv == 0 && st_r < aw - CV_BM) {
                const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
            }
            // scales of window rows [0, 512): one dword per lane, wave w rows 64 w .. 64 w + 63 (rows past the window read as 0 or are never used)
            const int srow = (int)(row0 - base_row) + (g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx + wv * 64 + lane;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(sa_hi + 2 * A_BYTES + wv * 256), 4, (srow + (g_k0 / 32) * (int)p.sc_rows) * 4, 0, 0, 0);
            if (++g_ty == 3) { g_ty = 0; g_k0 += CV_BK; }
        };
        auto dma_b = [&](int slot) {                                      // f16 weight tile of the next sub-step
            unsigned char* sb_hi = b_ring + slot * B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
            CV_BLDS(rb_hi, sb_hi + wave_row * CV_ROW, b_v[0] + b_u);
            if (++bs_tap == 9) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        auto dma_q = [&](int buf) {                                       // qr weight tiles of the three taps of group q_ty/q_k0 + their scales
            unsigned char* qb = qb_base + buf * QB_BYTES;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int b_u = ((q_ty * 3 + tx) * p.cout_pad * p.cin + q_k0) * 2;
                CV_BLDS(rb_lo, qb + tx * B_BYTES + wave_row * CV_ROW, b_v[0] + b_u);
            }
            // scales [3][BN]: 3 * BN dwords of the 512-dword field; wave w fetches entries 64 w .. 64 w + 63 (past 3 * BN: out of range -> 0)
            const int e = wv * 64 + lane, tx = e / BN, col = e - tx * BN;
            const int so = e < 3 * BN ? (((q_ty * 3 + tx) * (p.cin / 32) + q_k0 / 32) * p.cout_pad + n0 + col) * 4 : 0x7ffffff0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(qb + 3 * B_BYTES + wv * 256), 4, so, 0, 0, 0);
            if (++q_ty == 3) { q_ty = 0; q_k0 += CV_BK; }
        };
        // fragment addresses.  f16 window / weights: as the bf16 hi plane.  Correction operands: lane (r = lane & 15, g = lane >> 4)
        // holds 16 bytes of block g >> 1 (tap tx = g >> 1, channels 16 (g & 1) ..) and 16 bytes of block 2 + (g >> 1) (g < 2: tap 2;
        // g >= 2: block 3 — the weight side reads zeros, the activation side re-reads its first half: finite bytes x 0)
        const int g4 = lane >> 4, gh = g4 >> 1, gl = g4 & 1;
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, g4);
        int xq_off[2], xr_off[2];                                         // [half] within the qr window (slots 0, 1 = q; 2, 3 = r)
        {
            const int r1 = wm * (MF * 16) + frow + gh * p.tap_sx, r2 = g4 < 2 ? wm * (MF * 16) + frow + 2 * p.tap_sx : r1;
            xq_off[0] = A_BYTES + cv_swz(r1, gl); xq_off[1] = A_BYTES + cv_swz(r2, gl);
            xr_off[0] = A_BYTES + cv_swz(r1, 2 + gl); xr_off[1] = A_BYTES + cv_swz(r2, 2 + gl);
        }
        const int xs_off = 2 * A_BYTES + (wm * (MF * 16) + frow + (g4 < 3 ? g4 : 2) * p.tap_sx) * 4;   // scale of this lane's block (block 3: a valid one)
        const int wcol = wn * (NFW * 16) + frow;
        const int wq_off0 = gh * B_BYTES + cv_swz(wcol, gl), wr_off0 = gh * B_BYTES + cv_swz(wcol, 2 + gl);
        // second half: lanes g < 2 tap 2; lanes g >= 2 the zero bytes (an address independent of the fragment column: the column stride is folded in)
        const int wq_off1 = g4 < 2 ? 2 * B_BYTES + cv_swz(wcol, gl) : QB_ZERO, wr_off1 = g4 < 2 ? 2 * B_BYTES + cv_swz(wcol, 2 + gl) : QB_ZERO;
        const int wn_str = g4 < 2 ? 16 * CV_ROW : 0;
        const int ws_off = 3 * B_BYTES + ((g4 < 3 ? g4 : 2) * BN + wcol) * 4;
        const bool no_corr_mma = (p.variant & 0x1000) != 0;               // dev: timing ablation (wrong results)
        f16x8_t ah[MF], fbh[NFW];
        i32x8_t xr[MF], xq[MF], wq[NFW], wr[NFW];                          // correction operands: e4m3 lo / hi of the window rows, hi / lo of the weights
        int xsc[MF], wsc[NFW];
        auto ld8 = [&](const unsigned char* p0, const unsigned char* p1) {
            const uint4 u0 = *reinterpret_cast<const uint4*>(p0), u1 = *reinterpret_cast<const uint4*>(p1);
            return i32x8_t{(int)u0.x, (int)u0.y, (int)u0.z, (int)u0.w, (int)u1.x, (int)u1.y, (int)u1.z, (int)u1.w};
        };
        auto load_f16 = [&](const unsigned char* win, int slot, int tx) {
            const unsigned char* sb_hi = b_ring + slot * B_BYTES;
#pragma unroll
            for (int m = 0; m < MF; ++m) ah[m] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(win + a_offx[tx] + m * 16 * CV_ROW));
#pragma unroll
            for (int n = 0; n < NFW; ++n) fbh[n] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
        };
        // 16 f16 MFMAs (main term of this tap) + the correction MFMAs of fragment columns [a0, a1) of term 1 (hi_w x lo_x) and [b0, b1) of
        // term 2 (lo_w x hi_x).  Operands swapped as in cv_mma<true> (weights = A): C^T accumulators.  Scale bytes: 0 = hi (q), 1 = lo (r).
        // The 32 correction MFMAs of a group are spread 8 / 12 / 12 over its three compute phases: a LOAD phase costs ~400 cycles + ~14 per
        // fragment read whatever the other wave group computes, so the compute phases must be as even as the reads allow.
        auto mfmas = [&](const int a0, const int a1, const int b0, const int b1) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NFW; ++n)
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh[n], ah[m], acc[m][n], 0, 0, 0);
            if (!no_corr_mma) {
#pragma unroll
                for (int n = 0; n < NFW; ++n) {
                    if (n < a0 || n >= a1) continue;
#pragma unroll
                    for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wq[n], xr[m], acc[m][n], 0, 0, 0, wsc[n], 1, xsc[m]);
                }
#pragma unroll
                for (int n = 0; n < NFW; ++n) {
                    if (n < b0 || n >= b1) continue;
#pragma unroll
                    for (int m = 0; m < MF; ++m) acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wr[n], xq[m], acc[m][n], 0, 0, 1, wsc[n], 0, xsc[m]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            // (an MFMA may not leave its phase: without pinning the accumulators here hipcc sank correction MFMAs behind the barrier into the
            // next compute phase — operand registers of two phases live at once, spills, unbalanced phases)
            static_assert(MF == 4 && NFW == 4, "the accumulator pin below lists 16 accumulators");
            asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]),
                              "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3]));
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        auto ld_wq = [&](const unsigned char* qb, const int n) {
            wq[n] = ld8(qb + wq_off0 + n * 16 * CV_ROW, qb + wq_off1 + n * wn_str);
            wsc[n] = *reinterpret_cast<const int*>(qb + ws_off + n * 16 * 4);
        };
        auto ld_wr = [&](const unsigned char* qb, const int n) { wr[n] = ld8(qb + wr_off0 + n * 16 * CV_ROW, qb + wr_off1 + n * wn_str); };
#define CV_END_LOAD(N)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");                    \
        __builtin_amdgcn_s_barrier();                                                          \
        __builtin_amdgcn_sched_barrier(0);
        const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
        if (tid < 8) reinterpret_cast<uint32_t*>(qb_base + (tid >> 2) * QB_BYTES + QB_ZERO)[tid & 3] = 0u;
        // Prefetch distances.  The compute phases are short now (tx = 0: 16 MFMAs), so an f16 weight tile is fetched THREE sub-steps
        // ahead (4-slot ring) and the window / qr weights of group g + 1 at the first sub-step of group g.  Issue order per LOAD phase:
        // L0: window + qr weights of g + 1, then stage 3g + 3;  L1: stage 3g + 4;  L2: stage 3g + 5.  Needed at the end of L_s: stage
        // s + 1 (and, at L2, everything of group g + 1) — loads retire in order, so the waits count what may still be outstanding:
        //   L0: stage 3g+2 | window, qr | stage 3g+3  -> AP + 5 (+ 2 for wave 0) + 2;   L1: the same + stage 3g+4 minus stage 3g+2 (landed);
        //   L2: stages 3g+4, 3g+5 -> 2.
        dma_a(0);
        dma_q(0);
        dma_b(0);
        dma_b(1);
        dma_b(2);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                  // window 0, qr weights 0 and stage 0 landed (this wave's pieces)
        __syncthreads();
        if (grp == 1) __builtin_amdgcn_s_barrier();                       // group 1 runs one barrier (= half a sub-step) behind group 0
        int bslot = 0;                                                    // ring slot of the sub-step being read
        for (int g = 0; g < ngroups; ++g) {
            const bool lastg = g + 1 == ngroups;
            const unsigned char* win = a_base + (g & 1) * AW_BYTES;
            const unsigned char* qb = qb_base + (g & 1) * QB_BYTES;
            // ---- tx = 0: main term + term 1 of columns 0, 1 ----
            asm volatile("" ::: "memory");
            load_f16(win, bslot, 0);
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                xr[m] = ld8(win + xr_off[0] + m * 16 * CV_ROW, win + xr_off[1] + m * 16 * CV_ROW);
                xsc[m] = *reinterpret_cast<const int*>(win + xs_off + m * 64);
            }
            ld_wq(qb, 0); ld_wq(qb, 1);
            if (!lastg) {
                dma_a((g + 1) & 1); dma_q((g + 1) & 1); dma_b((bslot + 3) & 3);
                if (wv == 0) { CV_END_LOAD(AP + 5 + 2 + 2) } else { CV_END_LOAD(AP + 5 + 2) }
            } else { CV_END_LOAD(0) }
            mfmas(0, 2, 0, 0);
            bslot = (bslot + 1) & 3;
            // ---- tx = 1: main term + term 1 of columns 2, 3 + term 2 of column 0 ----
            asm volatile("" ::: "memory");
            load_f16(win, bslot, 1);
            ld_wq(qb, 2); ld_wq(qb, 3);
#pragma unroll
            for (int m = 0; m < MF; ++m) xq[m] = ld8(win + xq_off[0] + m * 16 * CV_ROW, win + xq_off[1] + m * 16 * CV_ROW);
            ld_wr(qb, 0);
            if (!lastg) {
                dma_b((bslot + 3) & 3);
                if (wv == 0) { CV_END_LOAD(AP + 5 + 2 + 2) } else { CV_END_LOAD(AP + 5 + 2) }
            } else { CV_END_LOAD(0) }
            mfmas(2, 4, 0, 1);
            bslot = (bslot + 1) & 3;
            // ---- tx = 2: main term + term 2 of columns 1, 2, 3 ----
            asm volatile("" ::: "memory");
            load_f16(win, bslot, 2);
            ld_wr(qb, 1); ld_wr(qb, 2); ld_wr(qb, 3);
            if (!lastg) { dma_b((bslot + 3) & 3); CV_END_LOAD(2) }
            else { CV_END_LOAD(0) }
            mfmas(0, 0, 1, 4);
            bslot = (bslot + 1) & 3;
        }
#undef CV_END_LOAD
        if (grp == 0) __builtin_amdgcn_s_barrier();                       // balance group 1's extra barrier
        __syncthreads();                                                  // nothing in flight; the ring is dead
    } else
    if constexpr (WIN == 5 && PP) {
        // ---- round 4: the ping-pong x register-window loop below with DEEPER prefetch.  Counters of the WIN == 2 loop (profiles/r4): the matrix
        // pipe is 72 % busy and a LOAD phase waits ~700 cycles whatever it reads — the LDS-DMA round trip under load is ~1.1 us = 2 000 cycles,
        // and a weight tile fetched two sub-steps (2 x 768 MFMA cycles) ahead has not landed.  The fused tail needs 128 KB of LDS anyway, so
        // the K loop may use it: the window is DOUBLE-buffered (the next group's window is fetched a whole group ahead, at tx = 0) and the
        // weight ring has FOUR slots (prefetch distance 3 sub-steps): 133 KB.
        // ---- ping-pong x register window: 256-row tile, two wave groups half a step apart as in the PP
        // loop below, but the LOAD phase of a sub-step is only the 8 weight-fragment reads (+ the 24 window reads once per group)
        // and ~3.4 DMA pieces per wave, so it fits under the other group's 48 MFMAs.  Phase order per group: L_s = reads of stage
        // s, DMA of stage s+2 (and of window g+1 at tx = 1), counted vmcnt for everything older, lgkmcnt(0), barrier; C_s = MFMAs,
        // barrier.  RAW: a stage is read one L phase after every wave's wait for it and a barrier both groups passed; WAR: a
        // slot is refilled in the L phase after the one whose reads of it retired before a barrier both groups passed.
        static_assert(NT == 512 && SPB == 1 && BN % RP == 0 && CV_BM % RP == 0, "8 waves, whole DMA passes");
        constexpr int BP = 2 * B_PT, AP = 2 * A_PT;
        unsigned char* const a_base = smem;                                // 2 x [hi | lo] window
        unsigned char* const b_ring = smem + 4 * A_BYTES;                  // 4 x [hi | lo] weight tile
        const int aw = CV_BM + 2 * p.tap_sx;
        const int ngroups = 3 * ksteps_per_tap;
        int g_ty = 0, g_k0 = 0, bs_tap = 0, bs_k0 = 0;
        auto dma_a = [&](int buf) {
            unsigned char* sa_hi = a_base + buf * (2 * A_BYTES);
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) {
                CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
                CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
            }
            if (wv == 0 && st_r < aw - CV_BM) {
                const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
            }
            if (++g_ty == 3) { g_ty = 0; g_k0 += CV_BK; }
        };
        auto dma_b = [&](int slot) {
            unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            unsigned char* sb_lo = sb_hi + B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
#pragma unroll
            for (int i = 0; i < B_PT; ++i) {
                CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
            }
            if (++bs_tap == 9) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, lane >> 4);
        bf16x8_t ah[3][MF], al[3][MF], fbh[NFW], fbl[NFW];
        auto load_b = [&](int slot) {
            const unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            const unsigned char* sb_lo = sb_hi + B_BYTES;
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                fbh[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
                fbl[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off + n * 16 * CV_ROW));
            }
        };
        auto mfmas = [&](const bf16x8_t (&xh)[MF], const bf16x8_t (&xl)[MF]) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xl[m], fbh[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], fbl[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], fbh[n], acc[m][n]);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
#define CV_END_LOAD(N)                                                                         \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");                    \
        __builtin_amdgcn_s_barrier();
        const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
        // issue order per LOAD phase — L0: window g + 1, stage 3g + 3;  L1: stage 3g + 4;  L2: stage 3g + 5.  Needed at the end of L_s: stage s + 1
        // (at L2 also the window of g + 1, for the fragment reads of the next L0).  Loads retire in order; still outstanding may be:
        //   L0: stage 3g+2 | window | stage 3g+3;   L1: window | stage 3g+3 | stage 3g+4;   L2: stage 3g+4 | stage 3g+5.
        dma_a(0);
        dma_b(0);
        dma_b(1);
        dma_b(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BP) : "memory");             // window 0 and stage 0 landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier (= half a sub-step) behind group 0
        int bslot = 0;
        for (int g = 0; g < ngroups; ++g) {
            const bool lastg = g + 1 == ngroups;
            const unsigned char* a_win = a_base + (g & 1) * (2 * A_BYTES);
            // ---- tx = 0 ----
            asm volatile("" ::: "memory");
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    ah[tx][m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_win + a_offx[tx] + m * 16 * CV_ROW));
                    al[tx][m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_win + A_BYTES + a_offx[tx] + m * 16 * CV_ROW));
                }
            load_b(bslot);
            if (!lastg) {
                dma_a((g + 1) & 1); dma_b((bslot + 3) & 3);
                if (wv == 0) { CV_END_LOAD(2 * BP + AP + 2) } else { CV_END_LOAD(2 * BP + AP) }
            } else { CV_END_LOAD(0) }
            mfmas(ah[0], al[0]);
            bslot = (bslot + 1) & 3;
            // ---- tx = 1 ----
            asm volatile("" ::: "memory");
            load_b(bslot);
            if (!lastg) {
                dma_b((bslot + 3) & 3);
                if (wv == 0) { CV_END_LOAD(2 * BP + AP + 2) } else { CV_END_LOAD(2 * BP + AP) }
            } else { CV_END_LOAD(0) }
            mfmas(ah[1], al[1]);
            bslot = (bslot + 1) & 3;
            // ---- tx = 2 ----
            asm volatile("" ::: "memory");
            load_b(bslot);
            if (!lastg) { dma_b((bslot + 3) & 3); CV_END_LOAD(2 * BP) }
            else { CV_END_LOAD(0) }
            mfmas(ah[2], al[2]);
            bslot = (bslot + 1) & 3;
        }
#undef CV_END_LOAD
        if (grp == 0) __builtin_amdgcn_s_barrier();           // balance group 1's extra barrier
        __syncthreads();                                      // nothing in flight; the ring is dead
    } else
    if constexpr (WIN == 2 && PP) {
        // ---- ping-pong x register window (DEFAULT for 128-wide 3x3 layers): 256-row tile, two wave groups half a step apart as in the PP
        // loop below, but the LOAD phase of a sub-step is only the 8 weight-fragment reads (+ the 24 window reads once per group)
        // and ~3.4 DMA pieces per wave, so it fits under the other group's 48 MFMAs.  Phase order per group: L_s = reads of stage
        // s, DMA of stage s+2 (and of window g+1 at tx = 1), counted vmcnt for everything older, lgkmcnt(0), barrier; C_s = MFMAs,
        // barrier.  RAW: a stage is read one L phase after every wave's wait for it and a barrier both groups passed; WAR: a
        // slot is refilled in the L phase after the one whose reads of it retired before a barrier both groups passed.
        static_assert(NT == 512 && SPB == 1 && BN % RP == 0 && CV_BM % RP == 0, "8 waves, whole DMA passes");
        constexpr int BP = 2 * B_PT, AP = 2 * A_PT;
        unsigned char* const a_win = smem;
        unsigned char* const b_ring = smem + 2 * A_BYTES;
        const int aw = CV_BM + 2 * p.tap_sx;
        const int ngroups = 3 * ksteps_per_tap;
        int g_ty = 0, g_k0 = 0, bs_tap = 0, bs_k0 = 0;
        auto dma_a = [&]() {
            unsigned char* sa_hi = a_win;
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) {
                CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
                CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
            }
            if (wv == 0 && st_r < aw - CV_BM) {
                const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
            }
            if (++g_ty == 3) { g_ty = 0; g_k0 += CV_BK; }
        };
        auto dma_b = [&](int slot) {
            unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            unsigned char* sb_lo = sb_hi + B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
#pragma unroll
            for (int i = 0; i < B_PT; ++i) {
                CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
            }
            if (++bs_tap == 9) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, lane >> 4);
        // M32: fragment (row block mb, K half h) of the window at tap tx / of the weight tile: index mb * 2 + h of the same register arrays
        int a_offx32[3][2], b_off32[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) a_offx32[tx][h] = cv_swz32(wm * 64 + (lane & 31) + tx * p.tap_sx, 2 * h + (lane >> 5));
            b_off32[h] = cv_swz32(wn * 64 + (lane & 31), 2 * h + (lane >> 5));
        }
        bf16x8_t ah[3][MF], al[3][MF], fbh[NFW], fbl[NFW];
#ifdef CONV_ABL      // ablation builds: LDS reads by 32-bit address (a conditional read through the generic pointer trips a backend bug: null check of the LDS cast)
        typedef uint32_t cv_u32x4 __attribute__((ext_vector_type(4)));
#define CV_LD16(ptr) (*reinterpret_cast<const __attribute__((address_space(3))) cv_u32x4*>((uint32_t)(uintptr_t)(ptr)))
#else
#define CV_LD16(ptr) (*reinterpret_cast<const uint4*>(ptr))
#endif
        auto load_b = [&](int slot) {
            const unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            const unsigned char* sb_lo = sb_hi + B_BYTES;
            if constexpr (M32) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        fbh[nb * 2 + h] = __builtin_bit_cast(bf16x8_t, CV_LD16(sb_hi + b_off32[h] + nb * 32 * CV_ROW));
                        fbl[nb * 2 + h] = __builtin_bit_cast(bf16x8_t, CV_LD16(sb_lo + b_off32[h] + nb * 32 * CV_ROW));
                    }
                return;
            }
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                fbh[n] = __builtin_bit_cast(bf16x8_t, CV_LD16(sb_hi + b_off + n * 16 * CV_ROW));
                fbl[n] = __builtin_bit_cast(bf16x8_t, CV_LD16(sb_lo + b_off + n * 16 * CV_ROW));
            }
        };
        auto load_win = [&]() {
            if constexpr (M32) {
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ah[tx][mb * 2 + h] = __builtin_bit_cast(bf16x8_t, CV_LD16(a_win + a_offx32[tx][h] + mb * 32 * CV_ROW));
                            al[tx][mb * 2 + h] = __builtin_bit_cast(bf16x8_t, CV_LD16(a_win + A_BYTES + a_offx32[tx][h] + mb * 32 * CV_ROW));
                        }
                return;
            }
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    ah[tx][m] = __builtin_bit_cast(bf16x8_t, CV_LD16(a_win + a_offx[tx] + m * 16 * CV_ROW));
                    al[tx][m] = __builtin_bit_cast(bf16x8_t, CV_LD16(a_win + A_BYTES + a_offx[tx] + m * 16 * CV_ROW));
                }
        };
        auto mfmas = [&](const bf16x8_t (&xh)[MF], const bf16x8_t (&xl)[MF]) {
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M32) {
                // small terms first; consecutive MFMAs write different accumulators (4 blocks between two updates of one)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc32[mb][nb] = cv_mma32(xl[mb * 2 + h], fbh[nb * 2 + h], acc32[mb][nb]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc32[mb][nb] = cv_mma32(xh[mb * 2 + h], fbl[nb * 2 + h], acc32[mb][nb]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc32[mb][nb] = cv_mma32(xh[mb * 2 + h], fbh[nb * 2 + h], acc32[mb][nb]);
            } else {
#if defined(CONV_ABL) && (CONV_ABL & 8)
            if (false)
#endif
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xl[m], fbh[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], fbl[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], fbh[n], acc[m][n]);
            }
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };
#define CV_END_LOAD(N)                                                                         \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");                    \
        __builtin_amdgcn_s_barrier();
        const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
#ifdef CONV_ABL
        // timing ablations, compile-time (hipcc -DMAGNET_DEV -DCONV_ABL=bits; results are wrong): 1 no DMA inside the loop, 2 no weight-fragment
        // reads, 4 no window reads, 8 no MFMAs
        constexpr bool abl_dma = (CONV_ABL & 1) != 0, abl_b = (CONV_ABL & 2) != 0, abl_a = (CONV_ABL & 4) != 0;
#else
        constexpr bool abl_dma = false, abl_b = false, abl_a = false;
#endif
        dma_a();
        dma_b(0);
        dma_b(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BP) : "memory");                 // window 0 and stage 0 landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier (= half a sub-step) behind group 0
        if constexpr (abl_a) load_win();
        if constexpr (abl_b) load_b(0);
        for (int g = 0; g < ngroups; ++g) {
            const bool lastg = g + 1 == ngroups;
            // ---- tx = 0 ----
            asm volatile("" ::: "memory");
            if constexpr (!abl_a) load_win();
            if constexpr (!abl_b) load_b(0);
            if constexpr (!abl_dma) dma_b(2);                           // stage 3g + 2
            CV_END_LOAD(BP)                                   // stage 3g + 1 landed
            mfmas(ah[0], al[0]);
            // ---- tx = 1 ----
            asm volatile("" ::: "memory");
            if constexpr (!abl_b) load_b(1);
            if (!lastg) {
                if constexpr (!abl_dma) { dma_b(0); dma_a(); }          // stage 3g + 3, window g + 1
                if (wv == 0) { CV_END_LOAD(BP + AP + 2) } else { CV_END_LOAD(BP + AP) }      // stage 3g + 2 landed
            } else { CV_END_LOAD(0) }
            mfmas(ah[1], al[1]);
            // ---- tx = 2 ----
            asm volatile("" ::: "memory");
            if constexpr (!abl_b) load_b(2);
            if (!lastg) { if constexpr (!abl_dma) dma_b(1); CV_END_LOAD(BP) }         // stage 3g + 4; stage 3g + 3 and window g + 1 landed
            else { CV_END_LOAD(0) }
            mfmas(ah[2], al[2]);
        }
#undef CV_END_LOAD
#undef CV_LD16
        if (grp == 0) __builtin_amdgcn_s_barrier();           // balance group 1's extra barrier
        __syncthreads();                                      // nothing in flight; the ring is dead
    } else
    if constexpr (WIN == 2) {
        // ---- row-window loop, prefetch distance 2 (3x3 layers): the window's fragments for all three tx live in REGISTERS
        // (read once per group), so one LDS window slot suffices and the weight tiles get a 3-slot ring: stage s+2's LDS-DMA is
        // issued right after the barrier of sub-step s and stays in flight across two barriers behind counted s_waitcnt vmcnt
        // (loads retire in order: "all but the pieces issued in the previous sub-step" = stage s landed).  Issuing the pieces
        // later in the sub-step, between the MFMAs, was measured 2.5 % slower, s_setprio around the MFMA block 1 % slower (same box A/B).  3 sub-steps per
        // group and 3 slots: stage s = 3g + tx lives in slot tx — every LDS address below is static.
        static_assert(!PP && SPB == 1 && NT == 256 && (BN % RP == 0 || BN == 32), "window loop with register-resident A: 4 waves, whole DMA passes (or the 32-wide case)");
        // Measured and rejected: the same loop with a 2-slot weight ring and a two-pass (2 x 64 rows) fused tail so that THREE
        // workgroups fit a CU (49 KB of LDS, <= 168 registers): 96 of the 168 registers hold the window, the compiler serialises the
        // weight fragment reads with the MFMAs: 9.4 vs 5.0 ms of convolutions per C2 step.
        // BN = 32 (F-Net's 32-wide trunk): the weight tile is 4 pieces of 16 rows — wave w stages plane w & 1, rows (w >> 1) * 16 ...,
        // ONE piece per wave, so the counted waits stay wave-uniform (row-pass staging would give waves 2 and 3 nothing to count)
        constexpr int BP = BN == 32 ? 1 : 2 * B_PT;           // weight-tile DMA instructions per wave and stage
        constexpr int AP = 2 * A_PT;                          // window DMA instructions per wave (wave 0: + 2 for the rows past the tile)
        unsigned char* const a_win = smem;                    // [hi | lo], AW_ROWS rows each
        unsigned char* const b_ring = smem + 2 * A_BYTES;     // 3 x [hi | lo]
        const int aw = CV_BM + 2 * p.tap_sx;
        const int ngroups = 3 * ksteps_per_tap;               // (K chunk, ty) pairs, ty inner
        int g_ty = 0, g_k0 = 0, bs_tap = 0, bs_k0 = 0;
        auto dma_a = [&]() {
            unsigned char* sa_hi = a_win;
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) {
                CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
                CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
            }
            if (wv == 0 && st_r < aw - CV_BM) {
                const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
            }
            if (++g_ty == 3) { g_ty = 0; g_k0 += CV_BK; }
        };
        const int b_v32 = ((n0 + (wv >> 1) * 16 + (lane >> 2)) * p.cin + st_k) * 2;       // BN = 32: this wave's 16 weight rows
        auto dma_b = [&](int slot) {
            unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            unsigned char* sb_lo = sb_hi + B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
            if constexpr (BN == 32) {
                if (wv & 1) CV_BLDS(rb_lo, sb_lo + (wv >> 1) * 16 * CV_ROW, b_v32 + b_u);
                else        CV_BLDS(rb_hi, sb_hi + (wv >> 1) * 16 * CV_ROW, b_v32 + b_u);
            } else {
#pragma unroll
                for (int i = 0; i < B_PT; ++i) {
                    CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                    CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                }
            }
            if (++bs_tap == 9) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, lane >> 4);
        int a_offx32[3][2], b_off32[2];                      // M32: see the ping-pong loop above
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) a_offx32[tx][h] = cv_swz32(wm * 64 + (lane & 31) + tx * p.tap_sx, 2 * h + (lane >> 5));
            b_off32[h] = cv_swz32(wn * 64 + (lane & 31), 2 * h + (lane >> 5));
        }
        bf16x8_t ah[3][MF], al[3][MF];
        auto mfma_b = [&](int slot, const bf16x8_t (&xh)[MF], const bf16x8_t (&xl)[MF]) {
            const unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            const unsigned char* sb_lo = sb_hi + B_BYTES;
            bf16x8_t fbh[NFW], fbl[NFW];
            if constexpr (M32) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        fbh[nb * 2 + h] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off32[h] + nb * 32 * CV_ROW));
                        fbl[nb * 2 + h] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off32[h] + nb * 32 * CV_ROW));
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc32[mb][nb] = cv_mma32(xl[mb * 2 + h], fbh[nb * 2 + h], acc32[mb][nb]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc32[mb][nb] = cv_mma32(xh[mb * 2 + h], fbl[nb * 2 + h], acc32[mb][nb]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) acc32[mb][nb] = cv_mma32(xh[mb * 2 + h], fbh[nb * 2 + h], acc32[mb][nb]);
                return;
            }
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                fbh[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
                fbl[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off + n * 16 * CV_ROW));
            }
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                const bf16x8_t bh = fbh[n], bl = fbl[n];
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xl[m], bh, acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], bl, acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(xh[m], bh, acc[m][n]);
            }
        };
#define CV_WAIT_BARRIER(N)                                                                     \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");                    \
        __builtin_amdgcn_s_barrier();                                                          \
        asm volatile("" ::: "memory");
        dma_a();
        dma_b(0);
        dma_b(1);
        for (int g = 0; g < ngroups; ++g) {
            const bool lastg = g + 1 == ngroups;
            // ---- tx = 0: stage 3g (slot 0) and window g have landed; refill slot 2 with stage 3g + 2 ----
            CV_WAIT_BARRIER(BP)
            dma_b(2);
            if constexpr (M32) {
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ah[tx][mb * 2 + h] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_win + a_offx32[tx][h] + mb * 32 * CV_ROW));
                            al[tx][mb * 2 + h] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_win + A_BYTES + a_offx32[tx][h] + mb * 32 * CV_ROW));
                        }
            } else {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int m = 0; m < MF; ++m) {
                    ah[tx][m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_win + a_offx[tx] + m * 16 * CV_ROW));
                    al[tx][m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(a_win + A_BYTES + a_offx[tx] + m * 16 * CV_ROW));
                }
            }
            mfma_b(0, ah[0], al[0]);
            // ---- tx = 1: every wave holds window g in registers (lgkmcnt(0) before the barrier): fetch window g + 1 ----
            CV_WAIT_BARRIER(BP)
            if (!lastg) { dma_b(0); dma_a(); }
            mfma_b(1, ah[1], al[1]);
            // ---- tx = 2 (the window's pieces, issued after the weight pieces, stay in flight) ----
            if (!lastg) { CV_WAIT_BARRIER(BP + AP) dma_b(1); }
            else { CV_WAIT_BARRIER(0) }
            mfma_b(2, ah[2], al[2]);
        }
#undef CV_WAIT_BARRIER
        __syncthreads();                                      // nothing in flight; the ring is dead (the epilogue reuses it)
    } else if constexpr (WIN == 1) {
        static_assert(!PP && SPB == 1, "row-window loop: 2-slot ring");
        // LDS: [A window slot 0 | A window slot 1 | B slot 0 | B slot 1], each hi + lo
        unsigned char* const a_ring = smem;
        unsigned char* const b_ring = smem + 2 * (2 * A_BYTES);
        const int aw = CV_BM + (p.tap_n - 1) * p.tap_sx;      // window rows of this layer
        const int ngroups = p.tap_n * ksteps_per_tap;         // (K chunk, ty) pairs, ty inner
        int g_ty = 0, g_k0 = 0;                               // next window to fetch
        int bs_tap = 0, bs_k0 = 0;                            // next weight tile to fetch: tap = ty*tap_n + tx, K chunk
        auto dma_a = [&](int slot) {
            unsigned char* sa_hi = a_ring + slot * (2 * A_BYTES);
            unsigned char* sa_lo = sa_hi + A_BYTES;
            const int a_u = (((g_ty + p.tap_o0) * p.tap_sy + p.tap_o0 * p.tap_sx) * p.in_ld + g_k0) * 2;
#pragma unroll
            for (int i = 0; i < A_PT; ++i) {
                CV_BLDS(ra_hi, sa_hi + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
                CV_BLDS(ra_lo, sa_lo + (wave_row + i * RP) * CV_ROW, a_v[i] + a_u);
            }
            if (wv == 0 && st_r < aw - CV_BM) {               // the (tap_n-1)*tap_sx rows past the tile: a few lanes of wave 0
                const int ax = a_v[0] + CV_BM * p.in_ld * 2;
                CV_BLDS(ra_hi, sa_hi + CV_BM * CV_ROW, ax + a_u);
                CV_BLDS(ra_lo, sa_lo + CV_BM * CV_ROW, ax + a_u);
            }
            if (++g_ty == p.tap_n) { g_ty = 0; g_k0 += CV_BK; }
        };
        auto dma_b = [&](int slot) {
            unsigned char* sb_hi = b_ring + slot * (2 * B_BYTES);
            unsigned char* sb_lo = sb_hi + B_BYTES;
            const int b_u = (bs_tap * p.cout_pad * p.cin + bs_k0) * 2;
#pragma unroll
            for (int i = 0; i < B_PT; ++i) {
                if (st_r + i * RP < BN) {
                    CV_BLDS(rb_hi, sb_hi + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                    CV_BLDS(rb_lo, sb_lo + (wave_row + i * RP) * CV_ROW, b_v[i] + b_u);
                }
            }
            if (++bs_tap == p.taps) { bs_tap = 0; bs_k0 += CV_BK; }
        };
        // fragment rows shifted by tx*tap_sx: the swizzle term follows the ACTUAL LDS row, so one offset per tx
        int a_offx[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) a_offx[tx] = cv_swz(wm * (MF * 16) + frow + tx * p.tap_sx, lane >> 4);
        auto compute_w = [&](int aslot, int bslot, int a_of) {
            const unsigned char* sa_hi = a_ring + aslot * (2 * A_BYTES);
            const unsigned char* sa_lo = sa_hi + A_BYTES;
            const unsigned char* sb_hi = b_ring + bslot * (2 * B_BYTES);
            const unsigned char* sb_lo = sb_hi + B_BYTES;
            bf16x8_t ah[MF], al[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                ah[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_hi + a_of + m * 16 * CV_ROW));
                al[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_lo + a_of + m * 16 * CV_ROW));
            }
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
                const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off + n * 16 * CV_ROW));
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(al[m], bh, acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ah[m], bl, acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(ah[m], bh, acc[m][n]);
            }
        };
        dma_a(0);
        dma_b(0);
        __syncthreads();
        // ONE flat loop over the sub-steps (group g = (K chunk, ty), tx inner) with a single MFMA block per iteration: as a
        // g / tx loop nest the compiler peeled tx into three copies plus a remainder loop and resolved the accumulator phis
        // with 64 register moves (128 v_accvgpr moves before __launch_bounds__(NT, 2)) per group, draining the MFMA pipe.
        int bslot = 0, aslot = 0, tx = 0, a_of = a_offx[0];
        const int nsub = ngroups * p.tap_n;
        for (int s = 0; s < nsub; ++s) {
            const bool last_tx = tx + 1 == p.tap_n;
            if (s + 1 < nsub) dma_b(bslot ^ 1);                                    // next sub-step's weights
            if (last_tx && s + 1 < nsub) dma_a(aslot ^ 1);                         // next group's window
            compute_w(aslot, bslot, a_of);
            __syncthreads();
            bslot ^= 1;
            if (last_tx) { tx = 0; aslot ^= 1; a_of = a_offx[0]; }
            else { ++tx; a_of = tx == 1 ? a_offx[1] : a_offx[2]; }
        }
    } else
    if constexpr (PP) {
        static_assert(NT == 512 && BN % RP == 0 && CV_BM % RP == 0 && A_PT + B_PT <= NFW, "ping-pong loop: 8 waves, whole DMA passes, one DMA pass per N fragment");
        constexpr int GL = 2 * A_PT + 2 * B_PT;               // LDS-DMA instructions per wave and stage
        // fragments of one stage: read in the LOAD phase, consumed in the COMPUTE phase
        bf16x8_t fah[MF], fal[MF], fbh[NFW], fbl[NFW];
        auto load_frags = [&](int buf) {
            const unsigned char* sa_hi = smem + buf * STAGE_BYTES;
            const unsigned char* sa_lo = sa_hi + A_BYTES;
            const unsigned char* sb_hi = sa_hi + 2 * A_BYTES;
            const unsigned char* sb_lo = sb_hi + B_BYTES;
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                fah[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_hi + a_off + m * 16 * CV_ROW));
                fal[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sa_lo + a_off + m * 16 * CV_ROW));
            }
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
                fbh[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + b_off + n * 16 * CV_ROW));
                fbl[n] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + b_off + n * 16 * CV_ROW));
            }
        };
        const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
        if (0 < nsteps) CV_DMA(0, 0)
        if (1 < nsteps) CV_DMA(1, 1)
        if (nsteps > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");   // stage 0 landed (this wave's pieces)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... and every other wave's
        if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier (= half a step) behind group 0
        int slot = 0;
        for (int s = 0; s < nsteps; ++s) {
            // ---- LOAD phase (the other group computes meanwhile) ----
            asm volatile("" ::: "memory");
            load_frags(slot);
            // refill the slot last read one step ago: both groups' reads of it retired (lgkmcnt(0)) before barriers this wave
            // has passed since.  (Issuing the DMA pieces between the MFMAs of the compute phase instead was measured slower:
            // 6.15 vs 5.83 ms of convolutions per C2 step.)
            const int nslot = slot == 0 ? 2 : slot - 1;       // (slot + 2) % 3
            if (s + 2 < nsteps) {
                CV_DMA(s + 2, nslot)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");             // stage s+1 landed; stage s+2 stays in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // ---- COMPUTE phase ----
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NFW; ++n) {
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(fal[m], fbh[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(fah[m], fbl[n], acc[m][n]);
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m][n] = cv_mma<(TAIL > 0)>(fah[m], fbh[n], acc[m][n]);
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot = slot == 2 ? 0 : slot + 1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();           // balance group 1's extra barrier
        __syncthreads();                                      // nothing in flight (last waits were vmcnt(0)): the ring is dead
    } else {
#pragma unroll
    for (int q = 0; q < SPB; ++q)
        if (q < nsteps) CV_DMA(q, q)
    __syncthreads();                                          // drains the DMA (vmcnt) and publishes the first half
    for (int s = 0, half = 0; s < nsteps; s += SPB, half ^= 1) {
        // the other half of the ring was last read one interval ago and every wave has passed that barrier:
        // refill it now, the DMA flies during this interval's MFMAs
#pragma unroll
        for (int q = 0; q < SPB; ++q)
            if (s + SPB + q < nsteps) CV_DMA(s + SPB + q, (half ^ 1) * SPB + q)
#pragma unroll
        for (int q = 0; q < SPB; ++q)
            if (s + q < nsteps) compute(half * SPB + q);
        __syncthreads();                                      // next half landed; everyone done with this half
    }
    }

    if constexpr (TAIL > 0) {
        // ---- fused 1x1 tail: the 128 x 128 tile (bias, ReLU, re-split) becomes the LDS-resident input of three 1x1 layers ----
        static_assert(NF == 8 && WN == 2 && (CV_BM == 128 || CV_BM == 256) && CV_BM == (NT / 64) * 32,
                      "the fused tail is written for 128-channel tiles with 32 rows per wave");
        unsigned char* act_hi = smem;                                   // [CV_BM rows][256 B]; the K ring is dead (barrier above)
        unsigned char* act_lo = smem + CV_BM * 256;
        if constexpr (M32) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {          // C^T blocks (cv_mma32): registers 4 g4 .. 4 g4 + 3 = 4 consecutive channels of one row
                        const int trow = wm * 64 + mb * 32 + (lane & 31);
                        const int ch = wn * 64 + nb * 32 + g4 * 8 + (lane >> 5) * 4;
                        float v[4] = {acc32[mb][nb][g4 * 4 + 0], acc32[mb][nb][g4 * 4 + 1], acc32[mb][nb][g4 * 4 + 2], acc32[mb][nb][g4 * 4 + 3]};
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                        if (p.relu) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
                        }
                        uint32_t h01, l01, h23, l23;
                        split_bf16x2(v[0], v[1], h01, l01); split_bf16x2(v[2], v[3], h23, l23);
                        const int off = act_swz(trow, ch >> 3) + (ch & 7) * 2;
                        *reinterpret_cast<uint2*>(act_hi + off) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(act_lo + off) = make_uint2(l01, l23);
                    }
        } else
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int n = 0; n < NFW; ++n) {                   // C^T accumulators (cv_mma<true>): 4 consecutive channels of one row
                const int trow = wm * (MF * 16) + m * 16 + (lane & 15);
                const int ch = wn * (NFW * 16) + n * 16 + (lane >> 4) * 4;
                float v[4] = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]};   // includes ConvParams::addend (initial value)
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                if (p.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
                }
                uint32_t h01, l01, h23, l23;
                split_bf16x2(v[0], v[1], h01, l01); split_bf16x2(v[2], v[3], h23, l23);
                const int off = act_swz(trow, ch >> 3) + (ch & 7) * 2;
                *reinterpret_cast<uint2*>(act_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(act_lo + off) = make_uint2(l01, l23);
            }
        __syncthreads();
        if constexpr ((NT == 256 && CV_BM == 128) || (NT == 512 && CV_BM == 256)) {
            if (!(p.variant & 32)) {                          // dev (MAGNET_CONV_VARIANT=32): the row-owned tail below
                constexpr int NW = NT / 64;
                tail_layer_cols<8, false, CV_BM, NW>(p.tail_w_hi, p.tail_w_lo, p.tail_bias, act_hi, act_lo, nullptr, 0, row0, p.rows, lane, wv);
                tail_layer_cols<8, false, CV_BM, NW>(p.tail_w_hi + 128 * 128, p.tail_w_lo + 128 * 128, p.tail_bias + 128, act_hi, act_lo, nullptr,
                                                     0, row0, p.rows, lane, wv);
                if constexpr (TAIL == 9) {
                    if (p.up_out) {                           // fused convex upsampling: the 144 logits meet in LDS, never in HBM
                        tail_layer_cols<9, true, CV_BM, NW, true>(p.tail_w_hi + 2 * 128 * 128, p.tail_w_lo + 2 * 128 * 128, p.tail_bias + 256, act_hi,
                                                                  act_lo, nullptr, 0, row0, p.rows, lane, wv,
                                                                  UpArgs{p.up_depth, p.up_out, p.up_npred, p.up_h, p.up_w, p.up_B});
                        return;
                    }
                }
                tail_layer_cols<TAIL, true, CV_BM, NW>(p.tail_w_hi + 2 * 128 * 128, p.tail_w_lo + 2 * 128 * 128, p.tail_bias + 256, act_hi, act_lo,
                                                       p.out_f32, p.tail_cout, row0, p.rows, lane, wv,
                                                       (TAIL == 1 && p.gu_out) ? UpArgs{p.gu_in, p.gu_out, -1, p.up_h, p.up_w, p.up_B} : UpArgs{nullptr, nullptr, 0, 0, 0, 0});
                return;
            }
        }
        tail_layer<8, false>(p.tail_w_hi, p.tail_w_lo, p.tail_bias, act_hi, act_lo, nullptr, 0, row0, p.rows, lane, wv);
        tail_layer<8, false>(p.tail_w_hi + 128 * 128, p.tail_w_lo + 128 * 128, p.tail_bias + 128, act_hi, act_lo, nullptr, 0, row0,
                             p.rows, lane, wv);
        tail_layer<TAIL, true>(p.tail_w_hi + 2 * 128 * 128, p.tail_w_lo + 2 * 128 * 128, p.tail_bias + 256, act_hi, act_lo, p.out_f32,
                               p.tail_cout, row0, p.rows, lane, wv);
        return;
    }

#ifdef MAGNET_DEV
    if (p.variant & 8192) return;                              // dev timing ablation (tools/conv_kscale.py): no epilogue
#endif
    // ---- epilogue through LDS, one 16-row fragment per wave at a time: [16 rows][NFW*16] fp32 per wave ----
    // Round 4 (tools/conv_kscale.py, profiles/r4/conv_no_epilogue.log): of the 8-wave kernel's 0.16 ms that do not depend on K (10 % of
    // the launch at K = 9 x 256), 0.13 ms is this epilogue = 6.6 us per tile = the tile's 128 KB at ~20 GB/s per CU — x 256 CUs in
    // lockstep (one workgroup per CU, equal tile times) 5 - 6 TB/s: the stores of all tiles arrive as ONE burst at the fabric's write
    // rate while the matrix pipes idle.  Tried against it: (1) persistent workgroups (see conv_mfma_kernel; no gain: a wave's loads and
    // stores share vmcnt, so the next tile's first counted wait still drains the stores); (2) start skew of the first-round workgroups
    // by eighths of a tile time (conv_start_skew.log: fixed part 0.162 -> 0.118 ms, but the late starters end the launch late: +3 % at
    // K = 9 x 512, -2.6 % at K = 9 x 64, a wash at the G-Net shapes); (3) C^T accumulators stored straight from registers, no LDS stage
    // (conv_direct_epilogue.log: fp32 outputs -1.5 % at K = 9 x 128, but the bf16 planes leave as 8-byte stores = 32-byte row segments and
    // the F-Net slows down 18.4 -> 19.3 ms).  What would remove it is an epilogue that overlaps another tile's K loop on the same CU
    // (two co-resident workgroups out of phase: needs the 8-wave loop in <= 80 KB of LDS and <= 128 registers) — not built.
    const long long tile_img = p.img_rows ? row0 / p.img_rows : 0;            // wave-uniform (scalar) division, once
    const int tile_rem = p.img_rows ? (int)(row0 - tile_img * p.img_rows) : 0;
    const float inv_wp = 1.0f / (float)(p.wp > 0 ? p.wp : 1);
    constexpr int WCOLS = NFW * 16, SROW = WCOLS + 4;          // +4 floats row pad
    float* stage = reinterpret_cast<float*>(smem) + wv * (16 * SROW);
    // a lane's work items of every fragment cover the same 8 channels when 64 is a multiple of the items per row: its bias values are
    // loaded once per tile (they used to be re-loaded behind every staging read)
    constexpr bool BIAS_INV = (64 % (WCOLS / 8)) == 0;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (BIAS_INV) {
        const int ch = n0 + wn * WCOLS + (lane % (WCOLS / 8)) * 8;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ch), b1 = *reinterpret_cast<const float4*>(p.bias + ch + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll
    // the staging rows are wave-private: after the K loop's final barrier (every wave is done with the ring) only the
    // wave's own LDS accesses need ordering — in-order in hardware, a scheduling fence for the compiler — no workgroup barrier
    for (int m = 0; m < MF; ++m) {                            // fully unrolled: acc indices stay static
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int n = 0; n < NFW; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                stage[((lane >> 4) * 4 + r) * SROW + n * 16 + (lane & 15)] = acc[m][n][r];   // C: col = lane&15, row = (lane>>4)*4+r
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 16 rows x WCOLS channels, 8 channels (one 16-byte bf16 vector / two fp32 vectors) per work item
        for (int it = lane; it < 16 * (WCOLS / 8); it += 64) {
            const int r = it / (WCOLS / 8), c8 = (it % (WCOLS / 8)) * 8;
            const long long row = row0 + wm * (MF * 16) + m * 16 + r;
            if (row >= p.rows) continue;
            const int ch = n0 + wn * WCOLS + c8;
            long long orow = row;
            bool interior = true;
            if (p.img_rows) {                                 // row -> (image, y, x) of the zero-bordered grid
                // the tile's first row is decoded once per workgroup on the scalar unit (tile_img, tile_rem); per work item a
                // 32-bit add, a wrap and a float-reciprocal division by the row pitch (exact: img_rows < 2^24, fixed up)
                long long img = tile_img;
                int rem = tile_rem + (int)(row - row0);
                while (rem >= p.img_rows) { rem -= p.img_rows; ++img; }
                int y = (int)((float)rem * inv_wp);
                if (y * p.wp > rem) --y;
                if ((y + 1) * p.wp <= rem) ++y;
                const int x = rem - y * p.wp;
                interior = (y >= p.pad) && (y < p.hp - p.pad) && (x >= p.pad) && (x < p.wp - p.pad);
                if (p.repad) {
                    if (!interior) continue;
                    const int q = p.repad - 1, hh = p.hp - 2 * p.pad, ww = p.wp - 2 * p.pad;
                    orow = (img * (hh + 2 * q) + (y - p.pad + q)) * (ww + 2 * q) + (x - p.pad + q);
                }
            }
            float v[8];
            float ad[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.add_hi) {                                   // residual input, stored as split bf16 planes
                const uint4 ah = *reinterpret_cast<const uint4*>(p.add_hi + (size_t)row * p.add_ld + ch);
                const uint4 al = *reinterpret_cast<const uint4*>(p.add_lo + (size_t)row * p.add_ld + ch);
                const uint32_t hw_[4] = {ah.x, ah.y, ah.z, ah.w}, lw_[4] = {al.x, al.y, al.z, al.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ad[2 * i]     = __uint_as_float(hw_[i] << 16) + __uint_as_float(lw_[i] << 16);
                    ad[2 * i + 1] = __uint_as_float(hw_[i] & 0xffff0000u) + __uint_as_float(lw_[i] & 0xffff0000u);
                }
            }
            if (p.addend) {                                   // loop-invariant partial sums computed once per forward
                const float4 a0 = *reinterpret_cast<const float4*>(p.addend + (size_t)row * p.addend_ld + ch);
                const float4 a1 = *reinterpret_cast<const float4*>(p.addend + (size_t)row * p.addend_ld + ch + 4);
                ad[0] = a0.x; ad[1] = a0.y; ad[2] = a0.z; ad[3] = a0.w; ad[4] = a1.x; ad[5] = a1.y; ad[6] = a1.z; ad[7] = a1.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = (stage[r * SROW + c8 + i] + ad[i]) + (BIAS_INV ? bv[i] : p.bias[ch + i]);
                v[i] = (p.relu && x < 0.f) ? 0.f : x;
                if (!interior) v[i] = 0.f;
            }
            const size_t e = (size_t)orow * p.out_ld + ch;
            if (p.out_mode == 2) {                            // single bf16 plane (RNE): the matcher's bf16 feature storage
                uint32_t h[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = f32x2_to_bf16x2_rne(v[2 * i], v[2 * i + 1]);
                *reinterpret_cast<uint4*>(p.out_hi + e) = make_uint4(h[0], h[1], h[2], h[3]);
            } else if (p.out_mode == 0) {
                uint32_t h[4], l[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) split_bf16x2(v[2 * i], v[2 * i + 1], h[i], l[i]);
                *reinterpret_cast<uint4*>(p.out_hi + e) = make_uint4(h[0], h[1], h[2], h[3]);
                *reinterpret_cast<uint4*>(p.out_lo + e) = make_uint4(l[0], l[1], l[2], l[3]);
            } else {
                *reinterpret_cast<float4*>(p.out_f32 + e) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.out_f32 + e + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
#endif
}

// PERSIST (round 4, dev only): one workgroup per CU walks the tiles bid, bid + gridDim.x, ... instead of one workgroup per tile.
// tools/conv_kscale.py splits the 8-wave kernel's launch time into 182 us per 32-channel group (507 TFLOP/s fp32-equivalent inside the K
// loop) + 0.16 - 0.20 ms that do not depend on K (10 - 12 % at K = 9 x 256: ~10 us per tile).  The bet was that those are workgroup
// relaunch, kernel-argument loads and the drain of the epilogue's stores, which a persistent loop removes (no vmcnt wait between tiles:
// the K loops' counted waits only get stricter with stores outstanding).  Measured (profiles/r4/conv_persistent_ab.log): 1.616 vs
// 1.604 ms at K = 9 x 256, 2.134 vs 2.118 / 1.950 vs 1.914 ms on the two fused-tail stacks — the hardware's own relaunch already costs
// nothing; the fixed part is the tile's exposed first-stage DMA latency, the ping-pong ramp and the epilogue, which only a next-tile
// prefetch issued BEFORE the epilogue could hide (LDS for it exists only in the TAIL = 0 kernel).  Kept for that experiment.
template <int NF, int WN, int CV_BM, int SPB, int TAIL = 0, int NT = 256, bool PP = false, int WIN = 0, bool PERSIST = false, bool M32 = false>
__global__ __launch_bounds__(NT, 2) void conv_mfma_kernel(const ConvParams p) {
    if constexpr (PERSIST) {
        const unsigned n_tiles = (unsigned)((p.rows + CV_BM - 1) / CV_BM);     // (loop-invariant hoisting is the trap of this form: see below)
        for (unsigned t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            // the tail's weight / bias loads have the same addresses for every tile: opaque pointers per trip, or the compiler hoists
            // hundreds of registers of them out of this loop (first version: 1.2 - 2 KB of scratch per lane)
            ConvParams q = p;
            asm volatile("" : "+s"(q.tail_w_hi), "+s"(q.tail_w_lo), "+s"(q.tail_bias), "+s"(q.bias), "+s"(q.w_hi), "+s"(q.w_lo));
            // ... and an opaque thread id: every per-lane constant of a tile (fragment offsets, DMA offsets, the tails' column indices) is
            // the same for each tile, and hoisted out of the loop they would all be live across it
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            conv_mfma_tile<NF, WN, CV_BM, SPB, TAIL, NT, PP, WIN, M32>(q, t, n_tiles, blockIdx.y, tid);
            // the epilogue's LDS reads (staging rows, the tail's activation tile) retire before any wave's next-tile DMA lands
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        conv_mfma_tile<NF, WN, CV_BM, SPB, TAIL, NT, PP, WIN, M32>(p, blockIdx.x, gridDim.x, blockIdx.y, threadIdx.x);
    }
}

template <int NF, int WN, int BM, int SPB, int NT = 256, bool PP = false, int WIN = 0>
static size_t conv_lds_bytes() {
    const size_t tiles = (WIN == 4 && PP) ? 2 * (2 * (size_t)(BM + 8) * CV_ROW + 512 * 4) + 4 * (size_t)(NF * 16) * CV_ROW + 2 * (3 * (size_t)(NF * 16) * CV_ROW + 512 * 4 + 16)
                       : (WIN == 3 && PP) ? 4 * (size_t)(BM + 8) * CV_ROW + 4 * 2 * (size_t)(NF * 16) * CV_ROW
                       : (WIN == 1 && PP) ? 4 * (size_t)(BM + 8) * CV_ROW + 3 * 2 * (size_t)(NF * 16) * CV_ROW
                       : (WIN == 5 && PP) ? 4 * (size_t)(BM + 8) * CV_ROW + 4 * 2 * (size_t)(NF * 16) * CV_ROW
                       : WIN == 2 ? 2 * (size_t)(BM + 8) * CV_ROW + 3 * 2 * (size_t)(NF * 16) * CV_ROW
                                  : (PP ? 3 : 2 * SPB) * (2 * (size_t)(WIN ? BM + 8 : BM) * CV_ROW + 2 * (size_t)(NF * 16) * CV_ROW);
    const size_t stage = (size_t)(NT / 64) * 16 * ((NF / WN) * 16 + 4) * 4;
    return tiles > stage ? tiles : stage;
}

static int conv_cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <int NF, int WN, int BM, int SPB, int TAIL = 0, int NT = 256, bool PP = false, int WIN = 0, bool PERSIST = false, bool M32 = false>
static hipError_t launch_conv_nf(const ConvParams& p, hipStream_t s) {
    dim3 grid((unsigned)((p.rows + BM - 1) / BM), (unsigned)(p.cout_pad / (NF * 16))), block(NT);
    size_t lds = conv_lds_bytes<NF, WN, BM, SPB, NT, PP, WIN>();
    if (TAIL > 0 && lds < (size_t)BM * 512) lds = (size_t)BM * 512;     // the fused tail's activation tile: BM rows x 256 B x (hi, lo)
    if (TAIL == 9 && lds < (size_t)128 * 148 * 4) lds = (size_t)128 * 148 * 4;   // fused upsampling: 128 rows x 144 logits (+4 pad) fp32 (two 4-wave workgroups still fit a CU)
    if (PERSIST) {
        // one workgroup per CU (these tiles take more than half a CU's LDS); a multiple of 8 keeps a workgroup's tiles on one XCD
        unsigned ncu = (unsigned)conv_cu_count() / 8 * 8;
        if (ncu == 0) ncu = 8;
        if (grid.x > ncu) grid.x = ncu;
    }
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {          // > 64 KiB of dynamic LDS needs the opt-in attribute
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<NF, WN, BM, SPB, TAIL, NT, PP, WIN, PERSIST, M32>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<NF, WN, BM, SPB, TAIL, NT, PP, WIN, PERSIST, M32>), grid, block, lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv_mfma(const ConvParams& p, hipStream_t s) {
    // 128 channels: 2x2 waves of 64x64 on a 128-row tile.
    // Measured alternatives on the G-Net 3x3 layer (64 frames; this configuration: 2.31 ms): 64-row tile / 3 workgroups
    // per CU 2.80 ms; two K stages per barrier (4-stage ring, 128 KB LDS, 1 workgroup per CU) 3.51 ms; 256-row tile
    // (1 workgroup per CU) 3.45 ms.  The kernel lives on inter-workgroup overlap: keep 2 workgroups per CU.
    const bool pp = (p.variant & 2) != 0;        // dev (MAGNET_CONV_VARIANT=2): the 8-wave ping-pong K loop — measured equal to the
                                                 // default 4-wave / 2-slot loop (5.83 vs 5.91 ms per C2 step), so it is not the default
    if (p.tail_w_hi) {                           // 3x3 (or 1x1) 128-wide layer + its three 1x1 successors in one kernel
        if (p.cout_pad != 128) return hipErrorInvalidValue;
        if ((p.variant & 32) && (p.up_out || p.gu_out)) return hipErrorInvalidValue;   // dev: the row-owned tail has no fused update / upsampling
        if (pp) {
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true>(p, s);
        }
        if (p.in_sc) {                                          // round 4: fp16 + block-scaled e4m3 operand format (see the WIN == 4 loop)
            if (p.tap_n != 3 || !p.w_sc || p.rows < 256ll * 256 || p.addend) return hipErrorInvalidValue;
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 4>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 4>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 4>(p, s);
            return hipErrorInvalidValue;
        }
        const bool win = p.tap_n > 1 && !(p.variant & 1);       // dev (MAGNET_CONV_VARIANT=1): one A stage per tap
        if (win && (p.variant & 4)) {                           // dev (MAGNET_CONV_VARIANT=4): 256-row tile, 8 waves, one workgroup per CU
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, false, 1>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, false, 1>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, false, 1>(p, s);
        }
        if (win && p.tap_n == 3 && (p.variant & 128) && (p.cin / 32) % 2 == 0) {   // dev (MAGNET_CONV_VARIANT=128): + fragment double-buffering
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 3>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 3>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 3>(p, s);
        }
        if (win && p.tap_n == 3 && (p.variant & 64)) {          // dev (MAGNET_CONV_VARIANT=64): ping-pong x 2-slot LDS window, 256-row tile
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 1>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 1>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 1>(p, s);
        }
        // default: ping-pong x register window, 256-row tile, 8 waves — when there is at least one such tile per CU (256 CUs);
        // fewer rows (single-frame inference: 78 tiles at 120x160) spread better as 128-row tiles of the 4-wave kernel
        // — and not for the per-iteration launches of a hoisted first layer (fp32 addend, K = 9 x 32 .. 9 x 64): those are short K
        // loops between an exposed 128 KB addend load and the tail, where two co-resident 4-wave workgroups overlap better than one
        // 8-wave workgroup (same-box A/B: 425 vs 464 us at K = 288, 766 vs 804 us at K = 576)
        if (win && p.tap_n == 3 && !(p.variant & (16 | 8)) && p.rows >= 256ll * 256 && !p.addend && (p.variant & 512)) {   // dev (MAGNET_CONV_VARIANT=512): round 4's deeper-prefetch loop (double-buffered window, 4-slot weight ring) — measured 4 % SLOWER (2.19 vs 2.10 ms): the LOAD phases are not waiting for the DMA
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 5>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 5>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 5>(p, s);
        }
        if (win && p.tap_n == 3 && !(p.variant & (16 | 8)) && p.rows >= 256ll * 256 && (!p.addend || (p.variant & 256))) {   // dev (MAGNET_CONV_VARIANT=256): 8-wave form for addend launches too
                                                                // dev (MAGNET_CONV_VARIANT=16): the 4-wave register-window loop below
#ifdef MAGNET_DEV
            if (p.variant & 4096) {                             // dev (MAGNET_CONV_VARIANT=4096): persistent workgroups (see conv_mfma_kernel): 0.7 % SLOWER
                if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 2, true>(p, s);
                if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 2, true>(p, s);
                if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 2, true>(p, s);
            }
#endif
#ifdef MAGNET_DEV
            if (p.variant & 16384) {                            // dev (MAGNET_CONV_VARIANT=16384): round 5's 32x32x16 form (M32_), measured 5.5 % SLOWER
                if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 2, false, true>(p, s);
                if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 2, false, true>(p, s);
                if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 2, false, true>(p, s);
            }
#endif
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 256, 1, 1, 512, true, 2>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 256, 1, 8, 512, true, 2>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 256, 1, 9, 512, true, 2>(p, s);
        }
        if (win && p.tap_n == 3 && !(p.variant & 8)) {          // dev (MAGNET_CONV_VARIANT=8): the 2-slot window loop below
#ifdef MAGNET_DEV
            if (p.variant & 16384) {                            // (the 4-wave kernel in the same form: a frame gives the same bits alone and in a batch)
                if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 128, 1, 1, 256, false, 2, false, true>(p, s);
                if (p.tail_cout == 128) return launch_conv_nf<8, 2, 128, 1, 8, 256, false, 2, false, true>(p, s);
                if (p.tail_cout == 144) return launch_conv_nf<8, 2, 128, 1, 9, 256, false, 2, false, true>(p, s);
            }
#endif
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 128, 1, 1, 256, false, 2>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 128, 1, 8, 256, false, 2>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 128, 1, 9, 256, false, 2>(p, s);
        }
        if (win) {
            if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 128, 1, 1, 256, false, 1>(p, s);
            if (p.tail_cout == 128) return launch_conv_nf<8, 2, 128, 1, 8, 256, false, 1>(p, s);
            if (p.tail_cout == 144) return launch_conv_nf<8, 2, 128, 1, 9, 256, false, 1>(p, s);
        }
        if (p.tail_cout == 16)  return launch_conv_nf<8, 2, 128, 1, 1>(p, s);
        if (p.tail_cout == 128) return launch_conv_nf<8, 2, 128, 1, 8>(p, s);
        if (p.tail_cout == 144) return launch_conv_nf<8, 2, 128, 1, 9>(p, s);
        return hipErrorInvalidValue;
    }
    // (the 8-wave ping-pong form of this loop, the default of the fused-tail kernels above, is no faster on the F-Net's plain
    // 128-wide layers: 18.93 vs 18.79 ms per 40 images)
    // the loop-invariant x_d3 part of G-Net's first layer (fp32 partial sums, I >= 2): the fused-tail kernels' 8-wave ping-pong x
    // register-window loop (same-box: C3 step -1 %); on the F-Net's plain 128-wide layers (split-bf16 outputs) it is equal to the
    // 4-wave loop (18.64 - 18.72 vs 18.69 - 18.75 ms per 40 images; dev MAGNET_CONV_VARIANT=2048 forces it there)
    if (p.cout_pad == 128 && p.tap_n == 3 && (p.out_mode == 1 || (p.variant & 2048)) && !(p.variant & (16 | 9)) && p.rows >= 256ll * 256 && !p.add_hi && !p.img_rows)
    {
#ifdef MAGNET_DEV
        if (p.variant & 4096) return launch_conv_nf<8, 2, 256, 1, 0, 512, true, 2, true>(p, s);
#endif
        return launch_conv_nf<8, 2, 256, 1, 0, 512, true, 2>(p, s);
    }
    if (p.cout_pad % 128 == 0 && p.tap_n == 3 && !(p.variant & 9) && !pp) return launch_conv_nf<8, 2, 128, 1, 0, 256, false, 2>(p, s);
    if (p.cout_pad % 128 == 0 && p.tap_n > 1 && !(p.variant & 1) && !pp) return launch_conv_nf<8, 2, 128, 1, 0, 256, false, 1>(p, s);
    if (p.cout_pad % 128 == 0 && pp) return launch_conv_nf<8, 2, 256, 1, 0, 512, true>(p, s);
    if (p.cout_pad % 128 == 0) return launch_conv_nf<8, 2, 128, 1>(p, s);
    if (p.cout_pad == 144) return launch_conv_nf<9, 1, 128, 1>(p, s);
    if (p.cout_pad == 16)  return launch_conv_nf<1, 1, 128, 1>(p, s);
    // F-Net trunk widths: 4 waves stacked along M.  Measured on the whole F-Net (40 images, 23.5 ms): 256-row tiles for
    // the 32- / 64-wide layers 24.1 / 24.2 ms, 192-row tile for 64-wide 24.6 ms, 2x2 waves for 64-wide 23.6 ms — no better.
    // 32- and 64-wide 3x3 layers of the trunk: the register-window loop too (their A operand is 2/3 of the DMA pieces of a K
    // step); dev
    // (MAGNET_CONV_VARIANT=8): one A stage per tap
    // (round 3, with the register window: 256-row tiles = 64 rows per wave for the 64-wide layers 18.5 vs 18.0 ms per 40 images, for the
    // 32-wide ones 18.1 vs 18.0 — their fragment-read share was not the limit; not kept)
    if (p.cout_pad == 32 && p.tap_n == 3 && !(p.variant & 9))  return launch_conv_nf<2, 1, 128, 1, 0, 256, false, 2>(p, s);
    if (p.cout_pad == 64 && p.tap_n == 3 && !(p.variant & 9))  return launch_conv_nf<4, 1, 128, 1, 0, 256, false, 2>(p, s);
    if (p.cout_pad == 32)  return launch_conv_nf<2, 1, 128, 1>(p, s);
    if (p.cout_pad == 64)  return launch_conv_nf<4, 1, 128, 1>(p, s);
    return hipErrorInvalidValue;
}

// =====================================================================================================
// Fused tail of a stack: Conv1x1(128->128)+ReLU, Conv1x1(128->128)+ReLU, Conv1x1(128->cout) in ONE kernel.
// As separate launches these layers are HBM-bound (each reads and writes a (rows,128) split-bf16 tensor:
// 1.3 GB per layer per 64-frame launch).  Here a workgroup keeps its 128-row activation tile in LDS across
// the three layers; only the (rows,128) input is read and the (rows,cout) fp32 result written.
//   * 4 waves, each owns 32 rows of the tile for all layers (rows are wave-private: no barrier is needed
//     to hand a layer's output to the next layer, only for the shared weight tiles);
//   * operands are swapped (weights as the MFMA A operand, activations as B), so the accumulator layout is
//     C^T: a lane holds 4 CONSECUTIVE output channels of one row -> 8-byte bf16x4 LDS writes for hidden
//     layers, 16-byte fp32 global stores for the last one;
//   * activation tile rows are 256 B with slot ^= (row & 15) (conflict-free fragment reads); weight tiles are
//     streamed per 32-wide K chunk through the same double-buffered swizzled image as conv_mfma_kernel.
// one layer: acc[n][m] over K = 128 for this wave's 32 rows; NF = output fragments (channels / 16)
template <int NF, bool LAST>
__device__ __forceinline__ void chain_layer(const ChainParams& p, const uint16_t* __restrict__ w_hi,
                                            const uint16_t* __restrict__ w_lo, const float* __restrict__ bias,
                                            unsigned char* act_hi, unsigned char* act_lo, unsigned char* wst,
                                            long long row0, int tid, int lane, int wv) {
    constexpr int BN = NF * 16;
    constexpr int B_BYTES = BN * CV_ROW, WSTAGE = 2 * B_BYTES;
    constexpr int B_PT = (BN * 4 + 255) / 256;
    static_assert(B_PT <= 3, "weight staging registers are spelled out for these sizes");
    const int st_r = tid >> 2, st_q = tid & 3;
    uint4 bh0, bh1, bh2, bl0, bl1, bl2;
    bh1 = bh2 = bl1 = bl2 = make_uint4(0, 0, 0, 0);
#define CH_ISSUE(KK)                                                                               \
    {                                                                                              \
        const size_t e0 = (size_t)(st_r < BN ? st_r : 0) * 128 + (KK) * 32 + st_q * 8;             \
        bh0 = *reinterpret_cast<const uint4*>(w_hi + e0); bl0 = *reinterpret_cast<const uint4*>(w_lo + e0); \
        if constexpr (B_PT >= 2) { const size_t e1 = (size_t)(st_r + 64 < BN ? st_r + 64 : 0) * 128 + (KK) * 32 + st_q * 8;  \
            bh1 = *reinterpret_cast<const uint4*>(w_hi + e1); bl1 = *reinterpret_cast<const uint4*>(w_lo + e1); }            \
        if constexpr (B_PT >= 3) { const size_t e2 = (size_t)(st_r + 128 < BN ? st_r + 128 : 0) * 128 + (KK) * 32 + st_q * 8; \
            bh2 = *reinterpret_cast<const uint4*>(w_hi + e2); bl2 = *reinterpret_cast<const uint4*>(w_lo + e2); }            \
    }
#define CH_COMMIT(BUF)                                                                             \
    {                                                                                              \
        unsigned char* sb_hi = wst + (BUF) * WSTAGE; unsigned char* sb_lo = sb_hi + B_BYTES;       \
        if (st_r < BN) { *reinterpret_cast<uint4*>(sb_hi + cv_swz(st_r, st_q)) = bh0;             \
                         *reinterpret_cast<uint4*>(sb_lo + cv_swz(st_r, st_q)) = bl0; }            \
        if constexpr (B_PT >= 2) if (st_r + 64 < BN) { *reinterpret_cast<uint4*>(sb_hi + cv_swz(st_r + 64, st_q)) = bh1;    \
                                                       *reinterpret_cast<uint4*>(sb_lo + cv_swz(st_r + 64, st_q)) = bl1; }  \
        if constexpr (B_PT >= 3) if (st_r + 128 < BN) { *reinterpret_cast<uint4*>(sb_hi + cv_swz(st_r + 128, st_q)) = bh2;  \
                                                        *reinterpret_cast<uint4*>(sb_lo + cv_swz(st_r + 128, st_q)) = bl2; } \
    }
    f32x4_t acc[NF][2];
#pragma unroll
    for (int n = 0; n < NF; ++n) { acc[n][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[n][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const int frow = lane & 15, kslot = lane >> 4;
    const int w_off = cv_swz(frow, kslot);                              // + n*16*CV_ROW
    CH_ISSUE(0)
    __syncthreads();                                                    // previous layer is done with the weight stage
    CH_COMMIT(0)
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) CH_ISSUE(kk + 1)
        const unsigned char* sb_hi = wst + (kk & 1) * WSTAGE;
        const unsigned char* sb_lo = sb_hi + B_BYTES;
        bf16x8_t xh[2], xl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = wv * 32 + m * 16 + frow;
            xh[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_hi + act_swz(row, kk * 4 + kslot)));
            xl[m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(act_lo + act_swz(row, kk * 4 + kslot)));
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const bf16x8_t wh = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_hi + w_off + n * 16 * CV_ROW));
            const bf16x8_t wl = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(sb_lo + w_off + n * 16 * CV_ROW));
            // swapped operands: D = W_frag (rows = channels) x act_frag (cols = tile rows)  ->  C^T
            acc[n][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[0], acc[n][0], 0, 0, 0);
            acc[n][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[1], acc[n][1], 0, 0, 0);
            acc[n][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[0], acc[n][0], 0, 0, 0);
            acc[n][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[1], acc[n][1], 0, 0, 0);
            acc[n][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[0], acc[n][0], 0, 0, 0);
            acc[n][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[1], acc[n][1], 0, 0, 0);
        }
        if (kk + 1 < 4) CH_COMMIT((kk + 1) & 1)
        __syncthreads();
    }
    // epilogue: C^T layout — lane: channels n*16 + (lane>>4)*4 + r (r = 0..3), tile row m*16 + (lane & 15)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ch = n * 16 + (lane >> 4) * 4;
            const int trow = wv * 32 + m * 16 + (lane & 15);
            const float4 b4 = *reinterpret_cast<const float4*>(bias + ch);
            float v[4] = {acc[n][m][0] + b4.x, acc[n][m][1] + b4.y, acc[n][m][2] + b4.z, acc[n][m][3] + b4.w};
            if constexpr (!LAST) {
                uint32_t h01, l01, h23, l23;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] < 0.f ? 0.f : v[r];
                split_bf16x2(v[0], v[1], h01, l01); split_bf16x2(v[2], v[3], h23, l23);
                const int off = act_swz(trow, ch >> 3) + (ch & 7) * 2;          // 8 bytes: channels ch..ch+3
                *reinterpret_cast<uint2*>(act_hi + off) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(act_lo + off) = make_uint2(l01, l23);
            } else {
                const long long row = row0 + trow;
                if (row < p.rows)
                    *reinterpret_cast<float4*>(p.out + (size_t)row * p.cout_pad + ch) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#undef CH_ISSUE
#undef CH_COMMIT
}

template <int NFL>
__global__ __launch_bounds__(256) void conv1x1_chain_kernel(const ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* act_hi = smem;                      // [128 rows][256 B]
    unsigned char* act_lo = smem + 128 * 256;
    unsigned char* wst    = smem + 2 * 128 * 256;      // 2 stages x (hi, lo) x [max(128, NFL*16) rows][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long row0 = (long long)blockIdx.x * 128;
    // each wave loads its own 32 rows (x 16 slots x 2 planes)
    for (int i = lane; i < 32 * 16; i += 64) {
        const int r = wv * 32 + (i >> 4), slot = i & 15;
        long long row = row0 + r; row = row < p.rows ? row : p.rows - 1;
        const size_t e = (size_t)row * 128 + slot * 8;
        *reinterpret_cast<uint4*>(act_hi + act_swz(r, slot)) = *reinterpret_cast<const uint4*>(p.in_hi + e);
        *reinterpret_cast<uint4*>(act_lo + act_swz(r, slot)) = *reinterpret_cast<const uint4*>(p.in_lo + e);
    }
    chain_layer<8, false>(p, p.w_hi, p.w_lo, p.bias, act_hi, act_lo, wst, row0, tid, lane, wv);
    chain_layer<8, false>(p, p.w_hi + 128 * 128, p.w_lo + 128 * 128, p.bias + 128, act_hi, act_lo, wst, row0, tid, lane, wv);
    chain_layer<NFL, true>(p, p.w_hi + 2 * 128 * 128, p.w_lo + 2 * 128 * 128, p.bias + 256, act_hi, act_lo, wst, row0, tid, lane, wv);
}

template <int NFL>
static hipError_t launch_chain_nf(const ChainParams& p, hipStream_t s) {
    constexpr int BNMAX = (NFL * 16 > 128) ? NFL * 16 : 128;
    const size_t lds = 2 * 128 * 256 + 2 * 2 * (size_t)BNMAX * CV_ROW;
    hipLaunchKernelGGL((conv1x1_chain_kernel<NFL>), dim3((unsigned)((p.rows + 127) / 128)), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv1x1_chain(const ChainParams& p, hipStream_t s) {
    if (p.cout_pad == 16)  return launch_chain_nf<1>(p, s);
    if (p.cout_pad == 144) return launch_chain_nf<9>(p, s);
    if (p.cout_pad == 128) return launch_chain_nf<8>(p, s);
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
            split_bf16x2(tile[vc + 2 * k][q], tile[vc + 2 * k + 1][q], hh[k], ll[k]);
        }
        const size_t e = row * ctot + c_off + f0 + vc;
        *reinterpret_cast<uint4*>(out_hi + e) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(out_lo + e) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    }
}

// Wide variant (h*w % 4 == 0, C % 8 == 0): 128 pixels x 64 channels per workgroup, 8 independent 16-byte loads per thread
// (512-byte runs along the pixel axis), tile rows of 132 floats with the pixel index XOR-swizzled in 4-word blocks by the channel
// octet: the float4 tile writes and the transposed channel reads (8 lanes = one pixel's 8 octets) are both conflict-free.
constexpr int PW_PIX = 128, PW_S = 132;
__device__ __forceinline__ int pw_idx(int c, int q) { return c * PW_S + (q ^ (((c >> 3) & 7) << 2)); }
__global__ __launch_bounds__(256) void pack_split_wide_kernel(const float* __restrict__ in, uint16_t* __restrict__ out_hi,
                                                               uint16_t* __restrict__ out_lo, int C, int h, int w,
                                                               int ctot, int c_off, long long in_img_stride) {
    __shared__ __attribute__((aligned(16))) float tile[64 * PW_S];
    const int hw = h * w;
    const int bpi = (hw + PW_PIX - 1) / PW_PIX;
    const int n = blockIdx.x / bpi, p0 = (blockIdx.x % bpi) * PW_PIX, f0 = blockIdx.y * 64;
    const int tid = threadIdx.x, l4 = (tid & 31) * 4, cr = tid >> 5;          // 32 lanes x float4 = one channel row of the tile
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cr + 8 * k, f = f0 + c, pp = p0 + l4;
        v[k] = (pp < hw && f < C) ? *reinterpret_cast<const float4*>(in + (size_t)n * in_img_stride + (size_t)f * hw + pp)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(&tile[pw_idx(cr + 8 * k, l4)]) = v[k];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {                          // 128 pixels x 8 octets
        const int i = tid + it * 256;
        const int q = i >> 3, vc = (i & 7) * 8, pq = p0 + q;
        if (pq >= hw || f0 + vc >= C) continue;
        const int y = pq / w, x = pq - y * w;
        const size_t row = ((size_t)n * (h + 2) + (y + 1)) * (w + 2) + (x + 1);
        uint32_t hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            split_bf16x2(tile[pw_idx(vc + 2 * k, q)], tile[pw_idx(vc + 2 * k + 1, q)], hh[k], ll[k]);
        }
        const size_t e = row * ctot + c_off + f0 + vc;
        *reinterpret_cast<uint4*>(out_hi + e) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
        *reinterpret_cast<uint4*>(out_lo + e) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
    }
}

hipError_t launch_pack_split(const float* in, uint16_t* out_hi, uint16_t* out_lo, int N, int C, int h, int w,
                             int ctot, int c_off, long long in_img_stride, hipStream_t s) {
#ifdef MAGNET_DEV
    static const bool narrow = getenv("MAGNET_PACK_NARROW") != nullptr;      // dev A/B: the 64-pixel kernel
#else
    constexpr bool narrow = false;
#endif
    if (!narrow && (h * w) % 4 == 0 && C % 8 == 0 && in_img_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0) {
        const int bpw = (h * w + PW_PIX - 1) / PW_PIX;
        hipLaunchKernelGGL(pack_split_wide_kernel, dim3((unsigned)(N * bpw), (unsigned)((C + 63) / 64)), dim3(256), 0, s, in, out_hi, out_lo,
                           C, h, w, ctot, c_off, in_img_stride);
        return hipGetLastError();
    }
    const int bpi = (h * w + 63) / 64;
    const dim3 grid((unsigned)(N * bpi), (unsigned)((C + 63) / 64)), block(256);
    hipLaunchKernelGGL(pack_split_kernel, grid, block, 0, s, in, out_hi, out_lo, C, h, w, ctot, c_off, in_img_stride);
    return hipGetLastError();
}


// ---- round 4: NCHW fp32 -> the "2-unit" operand format of the 128-wide 3x3 layers (ConvParams::in_sc) -----------------------------------
// Per interior position (row) and 32-channel block: out_f16 = fp16(x) (64 B); out_qr = 32 e4m3 bytes of hi / 2^e_h, then 32 e4m3 bytes of
// lo / 2^e_l with lo = x - fp16(x) (exact in fp32); out_sc [block][row] = {e_h + 127, e_l + 127, 0, 0}: E8M0 exponents with
// max |.| / 2^e in [128, 256) (<= 448, the e4m3 maximum), 0 for an all-zero block.  Same tile / transposition as pack_split_wide_kernel;
// one thread = one (pixel, block).
__device__ __forceinline__ int mx_block_exp(float m) {                        // m = max |v| >= 0
    if (!(m > 0.f)) return -127;
    const int e = (int)((__float_as_uint(m) >> 23) & 0xff) - 127 - 7;        // floor(log2 m) - 7 (denormal m: exponent field 0 -> -134 -> clamped)
    return e < -127 ? -127 : e;
}
__global__ __launch_bounds__(256) void pack_mx_kernel(const float* __restrict__ in, uint16_t* __restrict__ out_f16, uint8_t* __restrict__ out_qr,
                                                       uint32_t* __restrict__ out_sc, int C, int h, int w, int ctot, int c_off,
                                                       long long sc_rows, long long in_img_stride) {
    __shared__ __attribute__((aligned(16))) float tile[64 * PW_S];
    const int hw = h * w;
    const int bpi = (hw + PW_PIX - 1) / PW_PIX;
    const int n = blockIdx.x / bpi, p0 = (blockIdx.x % bpi) * PW_PIX, f0 = blockIdx.y * 64;
    const int tid = threadIdx.x, l4 = (tid & 31) * 4, cr = tid >> 5;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cr + 8 * k, f = f0 + c, pp = p0 + l4;
        v[k] = (pp < hw && f < C) ? *reinterpret_cast<const float4*>(in + (size_t)n * in_img_stride + (size_t)f * hw + pp)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(&tile[pw_idx(cr + 8 * k, l4)]) = v[k];
    __syncthreads();
    const int q = tid & 127, blk = tid >> 7, pq = p0 + q;
    if (pq >= hw || f0 + blk * 32 >= C) return;
    const int y = pq / w, x = pq - y * w;
    const size_t row = ((size_t)n * (h + 2) + (y + 1)) * (w + 2) + (x + 1);
    float hi[32], lo[32], mh = 0.f, ml = 0.f;
    uint32_t hw16[16];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        // domain of the format: |x| <= 65504 (the fp16 plane); larger magnitudes are clamped — an Inf in the fp16 plane would make the
        // residual -Inf and poison the block exponent (the bf16x3 format has no such limit; include/magnet_hip.h states the domain)
        // NaN is not laundered by the clamp (v_med3_f32 would return a finite bound): it goes through to the fp16 plane and both
        // e4m3 planes and propagates through the matrix instructions exactly as in the bf16x3 format
        const float tv = tile[pw_idx(blk * 32 + c, q)];
        const float xv = tv != tv ? tv : __builtin_amdgcn_fmed3f(tv, -65504.0f, 65504.0f);
        const _Float16 hh = (_Float16)xv;
        hi[c] = (float)hh; lo[c] = xv - hi[c];
        mh = fmaxf(mh, fabsf(hi[c])); ml = fmaxf(ml, fabsf(lo[c]));
        const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, hh);
        if (c & 1) hw16[c >> 1] |= hb << 16; else hw16[c >> 1] = hb;
    }
    const int eh = mx_block_exp(mh), el = mx_block_exp(ml);
    const float ih = __uint_as_float((uint32_t)(127 - eh) << 23), il = __uint_as_float((uint32_t)(127 - el) << 23);   // 2^-e (e = -127: 2^127 x 0)
    uint32_t qw[8], rw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int a = 0, b = 0;
        a = __builtin_amdgcn_cvt_pk_fp8_f32(hi[4 * k] * ih, hi[4 * k + 1] * ih, a, false);
        a = __builtin_amdgcn_cvt_pk_fp8_f32(hi[4 * k + 2] * ih, hi[4 * k + 3] * ih, a, true);
        b = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * k] * il, lo[4 * k + 1] * il, b, false);
        b = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * k + 2] * il, lo[4 * k + 3] * il, b, true);
        qw[k] = (uint32_t)a; rw[k] = (uint32_t)b;
    }
    const size_t e = row * ctot + c_off + f0 + blk * 32;
    uint4* o16 = reinterpret_cast<uint4*>(out_f16 + e);
    uint4* oqr = reinterpret_cast<uint4*>(out_qr + e * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) o16[k] = make_uint4(hw16[4 * k], hw16[4 * k + 1], hw16[4 * k + 2], hw16[4 * k + 3]);
    oqr[0] = make_uint4(qw[0], qw[1], qw[2], qw[3]); oqr[1] = make_uint4(qw[4], qw[5], qw[6], qw[7]);
    oqr[2] = make_uint4(rw[0], rw[1], rw[2], rw[3]); oqr[3] = make_uint4(rw[4], rw[5], rw[6], rw[7]);
    out_sc[(size_t)((c_off + f0) / 32 + blk) * sc_rows + row] = (uint32_t)(eh + 127) | ((uint32_t)(el + 127) << 8);
}

hipError_t launch_pack_mx(const float* in, uint16_t* out_f16, uint8_t* out_qr, uint32_t* out_sc, int N, int C, int h, int w, int ctot, int c_off,
                          long long sc_rows, long long in_img_stride, hipStream_t s) {
    const int bpw = (h * w + PW_PIX - 1) / PW_PIX;
    hipLaunchKernelGGL(pack_mx_kernel, dim3((unsigned)(N * bpw), (unsigned)((C + 63) / 64)), dim3(256), 0, s, in, out_f16, out_qr, out_sc,
                       C, h, w, ctot, c_off, sc_rows, in_img_stride);
    return hipGetLastError();
}

#undef CV_DMA
#undef CV_BLDS

}  // namespace magnet
