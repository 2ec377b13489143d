// cost_volume_worklist.hip — the fast matching kernel: sparse worklist of distinct (pixel, quad)
// items + bf16 dot2 / fp32 fma correlation, per-candidate bilinear combine.
//
// Why not a dense LDS window + MFMA: with per-pixel Gaussian candidates the union footprint of a
// 64-pixel tile in a source view is large and heavy-tailed (C2, synthetic §8d inputs: mean 680
// texels, p99 3200) while one pixel's 64 candidates touch ~5-8 distinct 2x2 quads — a 2-3 % dense
// tile.  A dense [64 px x window] MFMA correlation would do ~40x the useful MACs and its fp32
// result tile would not fit LDS.  Instead the bilinear interpolation is pulled out of the channel
// sum (it is linear):   sum_f ref[f] * bilerp(src[f]; taps) = bilerp( <ref, src[tap]> ; taps ),
// so each DISTINCT quad of a pixel needs its four F-channel dot products only once, and
// candidates whose consistency gate is closed (or that fall outside the image) need none.
//
// One workgroup = 16x4 reference pixels x all D candidates; thread (l, s) = pixel l, candidate
// slice s = wave s.  The four waves only share the staged reference tile; everything else is
// WAVE-PRIVATE (own item list, own correlation table), so there is no barrier in the main loop and
// waves in geometry (VALU) and waves in correlation (memory) phases overlap freely on a CU.
// For each valid view, in rounds of R candidates per thread, each wave runs:
//   P1  geometry (bit-identical to the oracle: warp_math.hpp), (mu,sigma) taps from the padded
//       interleaved source gmm, the gate; a candidate that is in-image, gate-open and on a different
//       quad than the thread's current one appends an ITEM (pixel, padded texel index) to the wave's
//       LDS list (ballot/popcount compaction, no atomics).  Branch-free: closed or out-of-image
//       candidates read texel 0 of the zero border and are masked out.
//   P2  the wave sweeps its items: unit = (item, tap), 8 lanes per unit (lane>>3 & 3 = tap, lane & 7 =
//       16-byte channel chunk), so one 128-B bf16 texel is one coalesced 8-lane load and 8 units
//       (2 items) go per instruction; U passes of loads are issued before any is consumed;
//       v_dot2c_f32_bf16 / v_fma, 3-step DPP reduction, result into the LDS table C[item][tap].
//   P3  per candidate: C quad (fresh from LDS, or carried in registers when the quad did not change),
//       fused bilerp with the candidate's weights, fp64 view accumulation (homography.py:116,159).
// Gates and sample positions are exactly the reference's; only the association of the fp32 channel
// sum differs (documented tolerance in tests/parity.py).
#include "cv_common.hpp"

namespace magnet {

constexpr int WL_U = 4;                       // P2: units in flight per lane

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

constexpr int CODE_NONE = -2;     // candidate contributes nothing (gate closed / out of image / dead)
constexpr int CODE_REUSE = -1;    // same quad as the thread's previous open candidate: C is in registers

// R   = candidates per thread per round, NR = rounds per view (D <= 4*R*NR),
// CPL = 16-byte channel chunks per lane in P2 (F*sizeof(FeatT)/16 <= 8*CPL),
// FULL = the texel is exactly 8*CPL chunks (no per-chunk guards), MINW = waves/SIMD to compile for
template <typename FeatT, int R, int NR, int CPL, bool FULL, int MINW>
__global__ __launch_bounds__(256, MINW) void cv_worklist_kernel(const CvParams p) {
    constexpr int IPW = 64 * R + 8;                      // items per wave: worst case + dummy padding
    static_assert(NR <= 4, "the accumulate switch below has 4 cases");
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
    const float fw = (float)p.w, fh = (float)p.h;

    // ---- LDS carve-up (all offsets multiples of 16) ----
    const int ref_stride = p.F * (int)sizeof(FeatT) + 16;                    // bytes, +16 against bank conflicts
    unsigned char* ref_lds = smem;                                           // [64][ref_stride], shared
    unsigned char* wbase = smem + ((64 * ref_stride + 15) & ~15) + wv * (IPW * 20);
    float*    ctab  = reinterpret_cast<float*>(wbase);                       // [IPW][4]   wave-private
    uint32_t* items = reinterpret_cast<uint32_t*>(wbase + IPW * 16);         // [IPW]      wave-private

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
    __syncthreads();                                      // the only workgroup barrier

    const float r0 = p.rays[((size_t)b * 3 + 0) * hw + pix];
    const float r1 = p.rays[((size_t)b * 3 + 1) * hw + pix];
    const float r2 = p.rays[((size_t)b * 3 + 2) * hw + pix];
    float mu = 0.f, sg = 0.f;
    if (!p.d_volume) {
        mu = p.ref_gmm[((size_t)b * 2 + 0) * hw + pix];
        sg = p.ref_gmm[((size_t)b * 2 + 1) * hw + pix];
    }
    const GridConst gc = grid_const(p);
    const uint32_t lane_tag = (uint32_t)lane << 26;
    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    // P2 lane constants: which tap of the quad and which 16-byte channel chunk this lane handles
    const int sub = lane & 7, tap = (lane >> 3) & 3, upair = lane >> 5;       // upair: item 0/1 of the pass
    const uint32_t lane_src_off = (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes + (uint32_t)sub * 16u;
    const int nchunk = (int)(texel_bytes / 16);

    double acc[NR * R];
#pragma unroll
    for (int i = 0; i < NR * R; ++i) acc[i] = 0.0;

    for (int v = 0; v < p.V; ++v) {
        if (p.is_valid[b * p.V + v] != 1) continue;                          // homography.py:97 (workgroup-uniform)
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16,
                                             r0, r1, r2);
        const size_t sidx = (size_t)v * p.B + b;                             // view-major, homography.py:105
        // wave-uniform 64-bit bases + 32-bit per-lane byte offsets (one image < 4 GiB, checked on the host)
        const unsigned char* __restrict__ src =
            reinterpret_cast<const unsigned char*>(p.src_feat) + sidx * (size_t)Hp * Wp * texel_bytes;
        const unsigned char* __restrict__ sgm =
            reinterpret_cast<const unsigned char*>(p.src_gmm) + sidx * (size_t)Hp * Wp * 8;
        int    cur_q = -1;                                                   // quad whose C the thread holds
        float4 cur_c = make_float4(0.f, 0.f, 0.f, 0.f);

        // rounds are a REAL loop (unrolling them triples the register footprint); the fp64
        // accumulators stay in registers because the accumulate step below dispatches on r with
        // compile-time indices.
#pragma unroll 1
        for (int r = 0; r < NR; ++r) {
            if (r * R >= DPT) break;                                         // wave-uniform
            // ------------------------------ P1: geometry, gate, item creation ----------------------
            // (written with scalars and explicit mbcnt: bools kept in struct arrays get materialised
            //  into VGPRs and re-compared — that formulation cost 2x the VALU instructions)
            float4 wts[R];          // bilinear weights (nw, ne, sw, se) of each candidate
            int    code[R];         // >= 0: ctab slot of a fresh item; CODE_REUSE; CODE_NONE
            int    cnt = 0;         // wave-uniform running item count of this round
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int i = r * R + k;
                const int j = wv * DPT + i;
                const bool live = (i < DPT) && (j < p.D);                    // wave-uniform
                const int jc = live ? j : 0;
                float d;
                if (p.d_volume) d = p.d_volume[((size_t)b * p.D + jc) * hw + pix];
                else { const float sk = sg * p.k[jc]; d = mu + sk; }         // MAGNET.py:155
                d = live ? d : __builtin_nanf("");                           // dead slot -> out of image below
                float ix, iy, zw;
                project(pv, gc, d, ix, iy, zw);
                const float x0f = __builtin_floorf(ix), y0f = __builtin_floorf(iy);
                const float x1 = x0f + 1.0f, y1 = y0f + 1.0f;
                const float ax = x1 - ix, bx = ix - x0f, ay = y1 - iy, by = iy - y0f;
                Taps t;                                                      // ATen's weights, homography.py:150-152
                t.nw = ax * ay; t.ne = bx * ay; t.sw = ax * by; t.se = bx * by;
                // floor(ix) in [-1, w-1] <=> -1 <= ix < w (false for NaN): some tap may be inside the image
                const bool inwin = (ix >= -1.0f) && (ix < fw) && (iy >= -1.0f) && (iy < fh);
                const int x0 = (int)x0f, y0 = (int)y0f;
                // (h+2)*(w+2) < 2^24 (checked on the host): full-rate 24-bit multiply-add
                const uint32_t qi = inwin ? (uint32_t)(__mul24(y0 + 1, Wp) + (x0 + 1)) : 0u;
                const GmmPair g0 = *reinterpret_cast<const GmmPair*>(sgm + qi * 8u);
                const GmmPair g1 = *reinterpret_cast<const GmmPair*>(sgm + (qi + (uint32_t)Wp) * 8u);
                const float mu_w = bilerp(g0.mu0, g0.mu1, g1.mu0, g1.mu1, t);
                const float sg_w = bilerp(g0.sg0, g0.sg1, g1.sg0, g1.sg1, t);
                const bool gate = inwin && ((__builtin_fabsf(zw - mu_w) < sg_w * p.kappa) || (CV_DEV(p) & 2));   // homography.py:157-158
                // a fresh item is needed when the gate is open and the quad differs from the one held
                const bool fresh = gate && ((int)qi != cur_q);
                const unsigned long long bal = __ballot(fresh);
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)bal, (uint32_t)cnt));
                if (fresh) items[pos] = lane_tag | qi;
                code[k] = gate ? (fresh ? pos : CODE_REUSE) : CODE_NONE;
                cur_q = fresh ? (int)qi : cur_q;
                cnt += __popcll(bal);
                wts[k] = make_float4(t.nw, t.ne, t.sw, t.se);
            }
            // pad the list to a whole number of passes (2 items = 8 units per pass) with a harmless
            // dummy item: pixel 0, texel 0 (the zero border)
            if (lane == 0) items[cnt] = 0u;
            if (p.stats && lane == 0) atomicAdd(p.stats + 2, (unsigned)cnt);
            // LDS is in-order per wave: later reads by other lanes see these writes; only the
            // compiler has to be told not to reorder across this point.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ------------------------------ P2: correlation of the wave's (item, tap) units ------------
            {
                const int passes = (CV_DEV(p) & 1) ? 0 : (cnt + 1) >> 1;      // 2 items (8 units) per pass
                auto issue = [&](int ps, uint4 (&sv)[CPL], int& lpx) {
                    const uint32_t item = items[2 * ps + upair];
                    lpx = (int)(item >> 26);
                    // texel index < 2^24 and texel_bytes < 2^24: one full-rate 24-bit multiply, 32-bit offset
                    const unsigned char* sp = src + (__umul24(item & 0xffffffu, texel_bytes) + lane_src_off);
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc)
                        sv[cc] = (FULL || sub + 8 * cc < nchunk) ? *reinterpret_cast<const uint4*>(sp + cc * 128)
                                                                 : make_uint4(0, 0, 0, 0);
                };
                auto consume = [&](int ps, const uint4 (&sv)[CPL], int lpx) {
                    const unsigned char* rp = ref_lds + (__umul24((uint32_t)lpx, (uint32_t)ref_stride) + (uint32_t)sub * 16u);
                    float part = 0.f;
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        const uint4 rv = (FULL || sub + 8 * cc < nchunk) ? *reinterpret_cast<const uint4*>(rp + cc * 128)
                                                                         : make_uint4(0, 0, 0, 0);
                        part = dot_chunk(rv, sv[cc], part, FeatT());
                    }
                    part = reduce8(part);
                    if (sub == 0) ctab[(2 * ps + upair) * 4 + tap] = part;
                };
                int ps = 0;
                for (; ps + WL_U <= passes; ps += WL_U) {                    // U passes of loads in flight
                    uint4 sv[WL_U][CPL];
                    int   lpx[WL_U];
#pragma unroll
                    for (int a = 0; a < WL_U; ++a) issue(ps + a, sv[a], lpx[a]);
#pragma unroll
                    for (int a = 0; a < WL_U; ++a) consume(ps + a, sv[a], lpx[a]);
                }
                for (; ps < passes; ++ps) {
                    uint4 sv[CPL];
                    int   lpx;
                    issue(ps, sv, lpx);
                    consume(ps, sv, lpx);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // ------------------------------ P3: bilinear combine + fp64 view accumulation -------------
            float cval[R];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (code[k] >= 0) cur_c = *reinterpret_cast<const float4*>(ctab + code[k] * 4);
                Taps t; t.nw = wts[k].x; t.ne = wts[k].y; t.sw = wts[k].z; t.se = wts[k].w;
                const float c = bilerp(cur_c.x, cur_c.y, cur_c.z, cur_c.w, t);
                cval[k] = (code[k] != CODE_NONE) ? c : 0.f;
            }
            // acc[r*R + k] += (double)cval[k]   (homography.py:159,116) with static register indices
            switch (r) {
                case 0: { _Pragma("unroll") for (int k = 0; k < R; ++k) acc[k] += (double)cval[k]; } break;
                case 1: if constexpr (NR > 1) { _Pragma("unroll") for (int k = 0; k < R; ++k) acc[1 * R + k] += (double)cval[k]; } break;
                case 2: if constexpr (NR > 2) { _Pragma("unroll") for (int k = 0; k < R; ++k) acc[2 * R + k] += (double)cval[k]; } break;
                case 3: if constexpr (NR > 3) { _Pragma("unroll") for (int k = 0; k < R; ++k) acc[3 * R + k] += (double)cval[k]; } break;
                default: break;
            }
            // the next round's P1 rewrites `items` and the next P2 rewrites `ctab`: same wave, program
            // order, and the P3 reads above have been consumed -> one more compiler-only fence
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (inb) {
        const float fV = (float)p.V;
#pragma unroll
        for (int i = 0; i < NR * R; ++i) {
            const int j = wv * DPT + i;
            if (i < DPT && j < p.D)
                p.cost[(size_t)b * p.cost_bstride + (size_t)j * hw + pix] = (float)acc[i] / fV;   // :118,120
        }
    }
    if (p.stats && tid == 0) atomicAdd(p.stats + 0, 1u);
}

template <int R>
static size_t worklist_lds_bytes(const CvParams& p) {
    const int esz = p.feat_bf16 ? 2 : 4;
    const size_t ref = ((size_t)64 * (p.F * esz + 16) + 15) & ~(size_t)15;
    return ref + (size_t)4 * (64 * R + 8) * 20;
}

template <typename FeatT, int R, int NR, int CPL, bool FULL, int MINW>
static hipError_t launch_wl(const CvParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    hipLaunchKernelGGL((cv_worklist_kernel<FeatT, R, NR, CPL, FULL, MINW>), grid, block, worklist_lds_bytes<R>(p), stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL>
static hipError_t launch_wl_c(const CvParams& p, hipStream_t stream, bool* handled) {
    const int dpt = (p.D + 3) / 4;
    *handled = true;
    if (FULL && dpt > 8 && dpt <= 16) {
        // D in (32, 64] at F = 64 — the headline shapes.  Measured on MI355X (C2, 64 frames/launch):
        // bf16: R=4 @4 waves/SIMD 2.06 ms, R=4 @3 2.38, R=8 @2 2.31;  fp32: R=4 @3 2.62, R=4 @4 2.80, R=8 @2 3.30.
        // Dev overrides via ablate bits 2-3 (tools/ablate.py).
        const int sel = (CV_DEV(p) >> 2) & 3;
        if (sel == 1) return launch_wl<FeatT, 4, 4, CPL, FULL, 3>(p, stream);
        if (sel == 2) return launch_wl<FeatT, 4, 4, CPL, FULL, 4>(p, stream);
        if (sel == 3) return launch_wl<FeatT, 8, 2, CPL, FULL, 2>(p, stream);
        if (sizeof(FeatT) == 2) return launch_wl<FeatT, 4, 4, CPL, FULL, 4>(p, stream);
        return launch_wl<FeatT, 4, 4, CPL, FULL, 3>(p, stream);
    }
    if (dpt <= 4)       return launch_wl<FeatT, 4, 1, CPL, FULL, 2>(p, stream);
    else if (dpt <= 8)  return launch_wl<FeatT, 8, 1, CPL, FULL, 2>(p, stream);
    else if (dpt <= 16) return launch_wl<FeatT, 8, 2, CPL, FULL, 2>(p, stream);
    else if (dpt <= 32) return launch_wl<FeatT, 8, 4, CPL, FULL, 2>(p, stream);
    *handled = false;                                                     // D > 128: generic kernel
    return hipSuccess;
}

hipError_t launch_cv_worklist(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 24)) return hipSuccess;          // 24-bit texel index (u24 multiplies)
    if ((size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets
    const int nchunk = (int)(p.F * esz / 16);
    if (p.feat_bf16) {
        if (nchunk == 8)  return launch_wl_c<uint16_t, 1, true>(p, stream, handled);      // F = 64
        if (nchunk <= 8)  return launch_wl_c<uint16_t, 1, false>(p, stream, handled);
        if (nchunk <= 16) return launch_wl_c<uint16_t, 2, false>(p, stream, handled);
    } else {
        if (nchunk == 16) return launch_wl_c<float, 2, true>(p, stream, handled);         // F = 64
        if (nchunk <= 8)  return launch_wl_c<float, 1, false>(p, stream, handled);
        if (nchunk <= 16) return launch_wl_c<float, 2, false>(p, stream, handled);
        if (nchunk <= 32) return launch_wl_c<float, 4, false>(p, stream, handled);
    }
    return hipSuccess;                                                    // very wide F: generic kernel
}

}  // namespace magnet
