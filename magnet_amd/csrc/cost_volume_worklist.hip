// cost_volume_worklist.hip — the fast matching kernel: sparse worklist of distinct (pixel, quad)
// items + bf16 dot2 / fp32 fma correlation, per-candidate bilinear combine.
//
// Why not a dense LDS window + MFMA: with per-pixel Gaussian candidates the union footprint of a
// 64-pixel tile in a source view is large and heavy-tailed (C2, synthetic §8d inputs: mean 680
// texels, p99 3200) while one pixel's 64 candidates touch ~7 distinct 2x2 quads — a 2-3 % dense
// tile.  A dense [64 px x window] MFMA correlation would do ~40x the useful MACs and its fp32
// result tile would not fit LDS.  Instead the bilinear interpolation is pulled out of the channel
// sum (it is linear):   sum_f ref[f] * bilerp(src[f]; taps) = bilerp( <ref, src[tap]> ; taps ),
// so each DISTINCT quad of a pixel needs its four 64-channel dot products only once, and
// candidates whose consistency gate is closed (or that fall outside the image) need none.
//
// One workgroup = 16x4 reference pixels x all D candidates; thread (l, s) = pixel l, candidate
// slice s (wave s).  For each valid view, in rounds of R=4 candidates per thread:
//   P1  geometry (bit-identical to the oracle: warp_math.hpp), (mu,sigma) taps from the padded
//       interleaved source gmm, the gate; a candidate that is in-image, gate-open and on a different
//       quad than the thread's current one appends an ITEM (pixel, padded texel index) to its wave's
//       LDS list (ballot/popcount compaction — no cross-wave scan, no atomics).
//   P2  all 256 threads sweep the items: unit = (item, tap), 8 lanes per unit, 16 B of channels per
//       lane per step (one 128-B bf16 texel = one coalesced 8-lane load), v_dot2c_f32_bf16 / v_fma,
//       3-step DPP reduction, result into the LDS table C[item][tap].
//   P3  per candidate: C quad (fresh from LDS, or carried in registers when the quad did not change),
//       fused bilerp with the candidate's weights, fp64 view accumulation (homography.py:116,159).
// Gates and sample positions are exactly the reference's; only the association of the fp32 channel
// sum differs (documented tolerance in tests/parity.py).
#include "cv_common.hpp"

namespace magnet {

constexpr int WL_R = 4;                       // candidates per thread per round
constexpr int WL_ITEMS_PER_WAVE = 64 * WL_R;  // worst case: every candidate of the round opens a quad

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot_chunk(const uint4 a, const uint4 b, float acc, uint16_t) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), acc, false);
    return acc;
}
__device__ __forceinline__ float dot_chunk(const uint4 a, const uint4 b, float acc, float) {
    acc = __builtin_fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
    acc = __builtin_fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
    acc = __builtin_fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
    acc = __builtin_fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
    return acc;
}

__device__ __forceinline__ float reduce8(float v) {     // sum over aligned groups of 8 lanes (all lanes get it)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}

struct __attribute__((aligned(8))) GmmPair { float mu0, sg0, mu1, sg1; };   // two adjacent padded texels

// NR = rounds per view the kernel is compiled for (acc lives in NR*WL_R fp64 registers per thread)
template <typename FeatT, int NR>
__global__ __launch_bounds__(256) void cv_worklist_kernel(const CvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int DPT = (p.D + 3) >> 2;
    int tile, b;
    tile_of_block(p, tile, b);
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x = tx * TILE_W + (lane & (TILE_W - 1));
    const int y = ty * TILE_H + (lane / TILE_W);
    const bool inb = (x < p.w) && (y < p.h);
    const int xc = inb ? x : 0, yc = inb ? y : 0;
    const size_t hw = (size_t)p.h * p.w;
    const size_t pix = (size_t)yc * p.w + xc;
    const int Wp = p.w + 2, Hp = p.h + 2;

    // ---- LDS carve-up (all offsets multiples of 16) ----
    const int ref_stride = p.F * (int)sizeof(FeatT) + 16;                    // bytes, padded against bank conflicts
    unsigned char* ref_lds = smem;                                           // [64][ref_stride]
    float4*   ctab  = reinterpret_cast<float4*>(smem + ((64 * ref_stride + 15) & ~15));   // [4*ITEMS_PER_WAVE]
    uint32_t* items = reinterpret_cast<uint32_t*>(ctab + 4 * WL_ITEMS_PER_WAVE);          // [4*ITEMS_PER_WAVE]
    int*      counts = reinterpret_cast<int*>(items + 4 * WL_ITEMS_PER_WAVE);             // [4]

    // ---- stage the tile's reference features: 64 px x F, 16-byte vectors ----
    {
        const int vec_per_px = p.F * (int)sizeof(FeatT) / 16;
        for (int i = tid; i < 64 * vec_per_px; i += 256) {
            const int l = i / vec_per_px, c = i % vec_per_px;
            const int lx = tx * TILE_W + (l & (TILE_W - 1)), ly = ty * TILE_H + (l / TILE_W);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (lx < p.w && ly < p.h)
                v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.ref_feat) +
                        (((size_t)b * hw + (size_t)ly * p.w + lx) * p.F) * sizeof(FeatT) + (size_t)c * 16);
            *reinterpret_cast<uint4*>(ref_lds + l * ref_stride + c * 16) = v;
        }
    }

    const float r0 = p.rays[((size_t)b * 3 + 0) * hw + pix];
    const float r1 = p.rays[((size_t)b * 3 + 1) * hw + pix];
    const float r2 = p.rays[((size_t)b * 3 + 2) * hw + pix];
    float mu = 0.f, sg = 0.f;
    if (!p.d_volume) {
        mu = p.ref_gmm[((size_t)b * 2 + 0) * hw + pix];
        sg = p.ref_gmm[((size_t)b * 2 + 1) * hw + pix];
    }
    const GridConst gc = grid_const(p);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    double acc[NR * WL_R];
#pragma unroll
    for (int i = 0; i < NR * WL_R; ++i) acc[i] = 0.0;

    uint32_t* my_items = items + wv * WL_ITEMS_PER_WAVE;

    for (int v = 0; v < p.V; ++v) {
        if (p.is_valid[b * p.V + v] != 1) continue;                          // homography.py:97 (workgroup-uniform)
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16,
                                             r0, r1, r2);
        const size_t sidx = (size_t)v * p.B + b;                             // view-major, homography.py:105
        const unsigned char* __restrict__ src =
            reinterpret_cast<const unsigned char*>(p.src_feat) + sidx * (size_t)Hp * Wp * p.F * sizeof(FeatT);
        const float* __restrict__ sgm = p.src_gmm + sidx * (size_t)Hp * Wp * 2;
        int   cur_q = -1;                                                    // quad whose C the thread holds
        float4 cur_c = make_float4(0.f, 0.f, 0.f, 0.f);

#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r * WL_R >= DPT) break;                                      // workgroup-uniform
            // ------------------------------ P1: geometry, gate, item creation ----------------------
            Taps  tp[WL_R];
            int   q[WL_R];          // padded texel index of the quad origin, -1 = no contribution
            int   slot[WL_R];       // >= 0: index into ctab of a fresh item; -1: reuse cur_c / nothing
            int   cnt = 0;          // wave-uniform running item count of this round
#pragma unroll
            for (int k = 0; k < WL_R; ++k) {
                const int i = r * WL_R + k;
                const int j = wv * DPT + i;
                const bool live = (i < DPT) && (j < p.D);
                float d = 0.f;
                if (live) {
                    if (p.d_volume) d = p.d_volume[((size_t)b * p.D + j) * hw + pix];
                    else { const float sk = sg * p.k[j]; d = mu + sk; }      // MAGNET.py:155
                }
                float ix, iy, zw;
                project(pv, gc, d, ix, iy, zw);
                tp[k] = make_taps(ix, iy);
                const bool inwin = live && (tp[k].x0 >= -1) && (tp[k].x0 <= p.w - 1) &&
                                   (tp[k].y0 >= -1) && (tp[k].y0 <= p.h - 1);
                const int qi = inwin ? (tp[k].y0 + 1) * Wp + (tp[k].x0 + 1) : 0;
                bool gate = false;
                if (inwin) {
                    const GmmPair g0 = *reinterpret_cast<const GmmPair*>(sgm + (size_t)qi * 2);
                    const GmmPair g1 = *reinterpret_cast<const GmmPair*>(sgm + ((size_t)qi + Wp) * 2);
                    const float mu_w = bilerp(g0.mu0, g0.mu1, g1.mu0, g1.mu1, tp[k]);
                    const float sg_w = bilerp(g0.sg0, g0.sg1, g1.sg0, g1.sg1, tp[k]);
                    gate = __builtin_fabsf(zw - mu_w) < sg_w * p.kappa;     // homography.py:157-158
                }
                q[k] = gate ? qi : -1;
                // a fresh item is needed when the gate is open and the quad differs from the one held
                const bool fresh = gate && (qi != cur_q);
                const unsigned long long bal = __ballot(fresh);
                slot[k] = -1;
                if (fresh) {
                    const int pos = cnt + __popcll(bal & lt_mask);
                    my_items[pos] = ((uint32_t)lane << 26) | (uint32_t)qi;
                    slot[k] = wv * WL_ITEMS_PER_WAVE + pos;
                    cur_q = qi;
                }
                cnt += __popcll(bal);
            }
            if (lane == 0) counts[wv] = cnt;
            __syncthreads();

            // ------------------------------ P2: correlation of all (item, tap) units -----------------
            {
                const int c0 = counts[0], c1 = counts[1], c2 = counts[2], c3 = counts[3];
                const int units = 4 * (c0 + c1 + c2 + c3);
                const int sub = tid & 7;
                const int nchunk = p.F * (int)sizeof(FeatT) / 16;
                for (int u = tid >> 3; u < units; u += 32) {
                    int idx = u >> 2, w_ = 0;
                    const int tap = u & 3;
                    if (idx >= c0) { idx -= c0; w_ = 1; if (idx >= c1) { idx -= c1; w_ = 2; if (idx >= c2) { idx -= c2; w_ = 3; } } }
                    const int gi = w_ * WL_ITEMS_PER_WAVE + idx;
                    const uint32_t item = items[gi];
                    const int l = (int)(item >> 26);
                    const size_t texel = (size_t)(item & 0x3ffffffu) + (tap & 1) + (size_t)(tap >> 1) * Wp;
                    const unsigned char* sp = src + texel * p.F * sizeof(FeatT);
                    const unsigned char* rp = ref_lds + l * ref_stride;
                    float part = 0.f;
                    for (int c = sub; c < nchunk; c += 8) {
                        const uint4 sv = *reinterpret_cast<const uint4*>(sp + (size_t)c * 16);
                        const uint4 rv = *reinterpret_cast<const uint4*>(rp + c * 16);
                        part = dot_chunk(rv, sv, part, FeatT());
                    }
                    part = reduce8(part);
                    if (sub == 0) reinterpret_cast<float*>(ctab)[gi * 4 + tap] = part;
                }
            }
            __syncthreads();

            // ------------------------------ P3: bilinear combine + fp64 view accumulation -------------
#pragma unroll
            for (int k = 0; k < WL_R; ++k) {
                if (slot[k] >= 0) cur_c = ctab[slot[k]];
                if (q[k] >= 0) {
                    const float c = bilerp(cur_c.x, cur_c.y, cur_c.z, cur_c.w, tp[k]);
                    acc[r * WL_R + k] += (double)c;                          // homography.py:159,116
                }
            }
            // the next round's P1 overwrites items/counts only after every wave has passed the second
            // barrier above; ctab is rewritten only in the next P2, behind the next round's first barrier.
        }
    }

    if (inb) {
        const float fV = (float)p.V;
#pragma unroll
        for (int i = 0; i < NR * WL_R; ++i) {
            const int j = wv * DPT + i;
            if (i < DPT && j < p.D)
                p.cost[(size_t)b * p.cost_bstride + (size_t)j * hw + pix] = (float)acc[i] / fV;   // :118,120
        }
    }
    if (p.stats && tid == 0) atomicAdd(p.stats + 0, 1u);
}

static size_t worklist_lds_bytes(const CvParams& p) {
    const int esz = p.feat_bf16 ? 2 : 4;
    const size_t ref = ((size_t)64 * (p.F * esz + 16) + 15) & ~(size_t)15;
    return ref + (size_t)4 * WL_ITEMS_PER_WAVE * (16 + 4) + 16;
}

template <typename FeatT, int NR>
static hipError_t launch_wl(const CvParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    hipLaunchKernelGGL((cv_worklist_kernel<FeatT, NR>), grid, block, worklist_lds_bytes(p), stream, p);
    return hipGetLastError();
}

template <typename FeatT>
static hipError_t launch_wl_t(const CvParams& p, hipStream_t stream, bool* handled) {
    const int dpt = (p.D + 3) / 4;
    *handled = true;
    if (dpt <= 1 * WL_R)      return launch_wl<FeatT, 1>(p, stream);
    else if (dpt <= 2 * WL_R) return launch_wl<FeatT, 2>(p, stream);
    else if (dpt <= 4 * WL_R) return launch_wl<FeatT, 4>(p, stream);
    else if (dpt <= 8 * WL_R) return launch_wl<FeatT, 8>(p, stream);
    *handled = false;                                                     // D > 128: generic kernel
    return hipSuccess;
}

hipError_t launch_cv_worklist(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 26)) return hipSuccess;   // item packing: 26-bit texel index
    if (worklist_lds_bytes(p) > 64 * 1024) return hipSuccess;                    // very wide F: generic kernel
    return p.feat_bf16 ? launch_wl_t<uint16_t>(p, stream, handled) : launch_wl_t<float>(p, stream, handled);
}

}  // namespace magnet
