// cost_volume_cand.hip — fused sampling + warp + gate + score, "lane = candidate" mapping.
//
// A wave walks over reference pixels; its 64 lanes are the D candidates of one pixel (D = 64;
// for smaller D several pixels share a wave: lanes = PPW pixels x DL candidates, DL = pow2 >= D;
// for larger D a pixel takes several candidate blocks).  Consequences of this mapping:
//   * the 64 samples of a wave lie on ONE epipolar segment of the source view, sorted by depth:
//     the (mu,sigma) taps and the feature texels they touch are a handful of cache lines, so the
//     texture path coalesces them (the pixel-per-lane mapping scatters 64 lanes over the whole
//     depth range: one cache line per lane);
//   * "which candidates share a 2x2 quad" is a comparison with the previous lane (DPP wave_shr:1)
//     plus a ballot: the distinct gate-open quads of the wave (~5 at C2) become ITEMS in one step —
//     no per-thread rounds, no carried state, no accumulator arrays: one fp64 accumulator per lane;
//   * everything per pixel (ray, mu, sigma, K R ray, K t) is the same for all lanes of the pixel's
//     group; the depth-linear projection terms are precomputed once per (pixel, view) into a small
//     wave-private LDS table and read back as broadcasts;
//   * ~64 VGPRs instead of 128+, so 6-8 waves per SIMD hide the two dependent memory round trips of
//     an iteration (taps, then feature texels).
// The bilinear interpolation is pulled out of the channel sum exactly as in the worklist kernel:
//   sum_f ref[f]*bilerp(src[f]; taps) = bilerp(<ref, src[tap]>; taps): one F-channel dot product per
// distinct (quad, tap), LPU lanes per dot product (4 x 32 B for bf16 F = 64, else 8 x 16 B), v_dot2c_f32_bf16 / v_fma_f32,
// DPP reduction.  Geometry and gates are the oracle's to the bit (warp_math.hpp); only the
// association of the fp32 channel sum differs (tolerance in tests/parity.py).
//
// Workgroup = 256 threads = 4 independent waves; wave w owns row w of a 16x4 pixel tile.  There is
// no workgroup barrier; all LDS (projection table, item list, correlation table) is wave-private.
#include "cv_common.hpp"

namespace magnet {

typedef __attribute__((ext_vector_type(2))) __bf16 cbf16x2_t;

__device__ __forceinline__ float cdot_chunk(const uint4 a, const uint4 b, float acc, uint16_t) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cbf16x2_t, a.x), __builtin_bit_cast(cbf16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cbf16x2_t, a.y), __builtin_bit_cast(cbf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cbf16x2_t, a.z), __builtin_bit_cast(cbf16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(cbf16x2_t, a.w), __builtin_bit_cast(cbf16x2_t, b.w), acc, false);
    return acc;
}
__device__ __forceinline__ float cdot_chunk(const uint4 a, const uint4 b, float acc, float) {
    acc = __builtin_fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
    acc = __builtin_fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
    acc = __builtin_fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
    acc = __builtin_fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
    return acc;
}

__device__ __forceinline__ float creduce8(float v) {    // sum over aligned groups of 8 lanes
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}

__device__ __forceinline__ float creduce4(float v) {    // sum over aligned groups of 4 lanes
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return v;
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in order; only the compiler must not reorder across this.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct __attribute__((aligned(8))) CGmmPair { float mu0, sg0, mu1, sg1; };   // two adjacent padded texels

constexpr uint32_t KEY_CLOSED = 0xffffffffu;

// DL   = candidates per pixel group inside a wave (8,16,32,64); PPW = 64/DL pixels per iteration
// CPL  = 16-byte channel chunks per lane in the correlation (F*sizeof(FeatT)/16 <= 8*CPL), FULL = exactly
// MINW = waves per SIMD to compile for; VAR = compile-time specialisation: 0 generic matcher (run-time options), 1 est_costvolume_F
// mode, 2 production matcher (candidates sampled in-kernel, no stats counters, no dev ablations): fewer scalar tests and live registers
template <typename FeatT, int DL, int CPL, bool FULL, int MINW, int VAR, int LPU = 8>
__global__ __launch_bounds__(256, MINW) void cv_cand_kernel(const CvParams p) {
    constexpr int IPP = 64 / (4 * LPU);                   // items per correlation pass: LPU lanes per (item, tap) unit
    constexpr int CSTR = LPU * 16;                        // byte stride between a lane's channel chunks
    constexpr bool MODEF = VAR == 1;                      // est_costvolume_F semantics
    constexpr bool FASTV = VAR == 2;                      // production matcher: sampled candidates, no stats, no dev ablations
    const int abl = FASTV ? 0 : CV_DEV(p);
    uint32_t* const stats = FASTV ? nullptr : p.stats;
    const float* const d_volume = FASTV ? nullptr : p.d_volume;
    constexpr int PPW = 64 / DL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tile, b;
    tile_of_block(p, tile, b);
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int y = ty * TILE_H + wv;                       // this wave's pixel row
    const int yc = min(y, p.h - 1);
    const int x_base = tx * TILE_W;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const float fw = (float)p.w, fh = (float)p.h;
    const int JB = (p.D + DL - 1) / DL;                   // candidate blocks per pixel

    // ---- wave-private LDS ----
    constexpr int OUT_PX = 8;                            // pixels staged before a coalesced flush (32-byte row segments)
    const int wave_bytes = p.V * 512 + 1024 + 272 + OUT_PX * DL * 4;
    unsigned char* wbase = smem + wv * wave_bytes;
    float4*   pvtab = reinterpret_cast<float4*>(wbase);                          // [V][16 px][2]
    float*    ctab  = reinterpret_cast<float*>(wbase + p.V * 512);               // [64 items][4 taps]
    uint32_t* items = reinterpret_cast<uint32_t*>(wbase + p.V * 512 + 1024);     // [64 + pad]
    float*    outb  = reinterpret_cast<float*>(wbase + p.V * 512 + 1024 + 272);  // [OUT_PX][DL] results of one block

    // ---- depth-linear projection terms for the wave's 16 pixels x V views (once per tile row) ----
    for (int e = lane; e < 16 * p.V; e += 64) {
        const int q = e & 15, v = e >> 4;
        const int xc = min(x_base + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, yc, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        pvtab[e * 2 + 0] = make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz);
        pvtab[e * 2 + 1] = make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz);
    }
    wave_lds_fence();

    const GridConst gc = grid_const(p);
    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    const int nchunk = (int)(texel_bytes / 16);
    // lane roles
    const int g = lane / DL, j0 = lane % DL;                                      // geometry: pixel group, candidate
    const int sub = lane & (LPU - 1), tap = (lane / LPU) & 3, upair = lane / (4 * LPU);   // correlation: chunk, tap, item of the pass
    const uint32_t lane_src_off = (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes + (uint32_t)sub * 16u;
    const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) +
        ((size_t)b * hw + (size_t)yc * p.w) * texel_bytes;                        // reference features of this pixel row
    const float fV = (float)p.V;
    // view validity (homography.py:97) as a bitmask read ONCE: a load inside the loop cannot be hoisted past the
    // kernel's global stores and costs a full memory round trip per (pixel, view) iteration
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    // (mu, sigma) of the wave's 16 reference pixels, lane q holds pixel q: read once per wave instead of a dependent global
    // load at the head of every pixel's view loop
    float mu_row = 0.f, sg_row = 0.f;
    if (!d_volume && !MODEF) {
        const size_t pixr = (size_t)yc * p.w + min(x_base + (lane & 15), p.w - 1);
        mu_row = p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr];
        sg_row = p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr];
    }

    for (int jb = 0; jb < JB; ++jb) {                                             // candidate block of DL candidates
        const int j = jb * DL + j0;
        const int jc = min(j, p.D - 1);
        const float kj = d_volume ? 0.f : p.k[jc];
        for (int qb = 0; qb < 16 / PPW; ++qb) {
            const int q = qb * PPW + g;                                           // pixel within the wave's row
            const int x = x_base + q;
            const bool pinb = (x < p.w) && (y < p.h);
            const size_t pix = (size_t)yc * p.w + min(x, p.w - 1);
            const bool live = pinb && (j < p.D);
            // PPW == 1: every correlation unit of this iteration belongs to this one pixel, so each lane
            // keeps its 16-byte chunk(s) of the pixel's reference vector in registers for all views
            uint4 rvp[CPL];
            if (PPW == 1) {
                const unsigned char* rp = ref_row + (__umul24((uint32_t)min(x, p.w - 1), texel_bytes) + (uint32_t)sub * 16u);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc)
                    rvp[cc] = (FULL || (sub + LPU * cc < nchunk)) ? *reinterpret_cast<const uint4*>(rp + cc * CSTR)
                                                                : make_uint4(0, 0, 0, 0);
            }
            float d;
            if (d_volume) d = d_volume[((size_t)b * p.D + jc) * hw + pix];
            else if (MODEF) d = kj;                                            // fixed depth bin (homography.py:54)
            else {
                const float mu = __shfl(mu_row, q);
                const float sg = __shfl(sg_row, q);
                const float sk = sg * kj; d = mu + sk;                            // MAGNET.py:155 (mul, then add)
            }
            d = live ? d : __builtin_nanf("");                                    // dead lane -> out of image below
            double acc = 0.0;
            float accf = 0.f;                                                     // mode 1: fp32 view sum (homography.py:42)

            for (int v = 0; v < p.V; ++v) {
                if (!((vmask >> v) & 1ull)) continue;                            // homography.py:97 (wave-uniform)
                const size_t sidx = (size_t)v * p.B + b;                         // view-major, homography.py:105
                const unsigned char* __restrict__ src =
                    reinterpret_cast<const unsigned char*>(p.src_feat) + sidx * (size_t)Hp * Wp * texel_bytes;
                const unsigned char* __restrict__ sgm =
                    reinterpret_cast<const unsigned char*>(p.src_gmm) + sidx * (size_t)Hp * Wp * 8;
                // ---------------- geometry + gate (bit-identical to the oracle) ----------------
                const float4 pa = pvtab[(v * 16 + q) * 2 + 0], pb = pvtab[(v * 16 + q) * 2 + 1];
                PixelView pv;
                pv.rpx = pa.x; pv.rpy = pa.y; pv.rpz = pa.z; pv.rcz = pa.w;
                pv.kt0 = pb.x; pv.kt1 = pb.y; pv.kt2 = pb.z; pv.tz = pb.w;
                float ix, iy, zw;
                project(pv, gc, d, ix, iy, zw);
                const float x0f = __builtin_floorf(ix), y0f = __builtin_floorf(iy);
                const float x1 = x0f + 1.0f, y1 = y0f + 1.0f;
                const float ax = x1 - ix, bx = ix - x0f, ay = y1 - iy, by = iy - y0f;
                Taps t;                                                          // ATen's weights, homography.py:150-152
                t.nw = ax * ay; t.ne = bx * ay; t.sw = ax * by; t.se = bx * by;
                // floor(ix) in [-1, w-1] <=> -1 <= ix < w (false for NaN): some tap may be inside the image
                const bool inwin = (ix >= -1.0f) && (ix < fw) && (iy >= -1.0f) && (iy < fh);
                const int x0 = (int)x0f, y0 = (int)y0f;
                const uint32_t qi = inwin ? (uint32_t)(__mul24(y0 + 1, Wp) + (x0 + 1)) : 0u;
                // (mu,sigma) taps.  Candidates are sorted along the epipolar segment, so lanes on the same
                // quad form runs: only the first lane of a run (its LEADER) loads the 2 x 16 bytes, the others
                // fetch them from the leader's registers (ds_bpermute: LDS crossbar, no LDS memory).  The
                // texture addresser spends one cycle per ACTIVE lane on such gathers — 64 -> ~7 per instruction.
                const uint32_t tkey = inwin ? qi : KEY_CLOSED;
                uint32_t tprev = (uint32_t)__builtin_amdgcn_update_dpp((int)KEY_CLOSED, (int)tkey, 0x138, 0xf, 0xf, false);  // wave_shr:1
                if (j0 == 0) tprev = KEY_CLOSED;                                  // first candidate of a pixel group
                const bool lead = inwin && (tkey != tprev);
                // tap registers: only leader lanes' values are ever read (through ds_bpermute); declared per iteration so that
                // they are not carried around the loop (8 VGPRs less during the correlation)
                CGmmPair g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
                if (lead && !(abl & 2) && !MODEF) {
                    g0 = *reinterpret_cast<const CGmmPair*>(sgm + qi * 8u);
                    g1 = *reinterpret_cast<const CGmmPair*>(sgm + (qi + (uint32_t)Wp) * 8u);
                }
                // gate = inwin && |z - mu_w| < kappa sigma_w (homography.py:157-158), evaluated from the leaders' tap registers.
                // Dev option (path bit 16 << 8): SPECULATIVE order — correlate every distinct IN-IMAGE quad and apply the gate
                // when the result is accumulated, so the feature loads do not wait for the (mu,sigma) taps (one memory round
                // trip per iteration instead of two).  Same results; measured SLOWER (C2 1.97 vs 1.89 ms, D=5 0.88 vs 0.67,
                // C4 2.10 vs 1.65): the extra (item, tap) dot products cost more than the round trip saves.
                bool gate = false;
                auto eval_gate = [&]() {
                    const unsigned long long lb = __ballot(lead);
                    const unsigned long long le = lb & (~0ull >> (63 - lane));    // leaders at or below this lane
                    const int ldr = 63 - __builtin_clzll(le | 1ull);              // nearest one (lane 0 if none: unused)
                    const int sel = ldr << 2;
                    g0.mu0 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g0.mu0)));
                    g0.sg0 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g0.sg0)));
                    g0.mu1 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g0.mu1)));
                    g0.sg1 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g0.sg1)));
                    g1.mu0 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g1.mu0)));
                    g1.sg0 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g1.sg0)));
                    g1.mu1 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g1.mu1)));
                    g1.sg1 = __int_as_float(__builtin_amdgcn_ds_bpermute(sel, __float_as_int(g1.sg1)));
                    const float mu_w = bilerp(g0.mu0, g0.mu1, g1.mu0, g1.mu1, t);
                    const float sg_w = bilerp(g0.sg0, g0.sg1, g1.sg0, g1.sg1, t);
                    gate = inwin && (__builtin_fabsf(zw - mu_w) < sg_w * p.kappa);    // homography.py:157-158
                    if (abl & 2) gate = inwin && ((j0 & 3) != 0);                // dev: taps skipped, ~75 % open
                    if (abl & 8) gate = false;                                   // dev: geometry only
                    if (MODEF) gate = inwin;                                       // est_costvolume_F has no gate
                };
                const bool spec = (abl & 16) && !(abl & 8);
                if (!spec) eval_gate();
                if (!FASTV && !MODEF && p.gate_bits && live)                      // debug output of the gate bits (parity tests)
                    p.gate_bits[(((size_t)b * p.V + v) * p.D + j) * hw + (size_t)y * p.w + x] = gate ? 1 : 0;
                const bool open = spec ? inwin : gate;                               // lanes whose quad becomes an item

                // ---------------- distinct open quads of the wave -> items ----------------
                const uint32_t key = open ? qi : KEY_CLOSED;
                uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)KEY_CLOSED, (int)key, 0x138, 0xf, 0xf, false);  // wave_shr:1
                if (j0 == 0) prev = KEY_CLOSED;                                   // first candidate of a pixel group
                const bool fresh = open && (key != prev);
                const unsigned long long bal = __ballot(fresh);
                const int nitems = __popcll(bal);
                if (nitems == 0) continue;                                        // wave-uniform: nothing open in this view
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                       __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                const int myitem = below + (fresh ? 1 : 0) - 1;                  // the item covering this lane (if open)
                if (fresh) items[below] = ((uint32_t)q << 26) | qi;
                if (lane == 0) items[nitems] = 0u;                                // pad to a whole pass: pixel 0, texel 0
                if (stats && lane == 0) atomicAdd(stats + 2, (unsigned)nitems);
                wave_lds_fence();

                // ---------------- correlation: unit = (item, tap), 8 lanes x 16 B per unit ----------------
                const int passes = (abl & 1) ? 0 : (nitems + IPP - 1) / IPP;  // IPP items (4*IPP units) per pass
                for (int ps = 0; ps < passes; ps += 2) {
                    uint4 sv[2][CPL], rv[2][CPL];
                    // second pass of the pair only if it holds an item (wave-uniform): at ~3.5 items per (pixel, view) a third of
                    // the iterations need one pass
                    const bool second = (IPP * (ps + 1) < nitems) || (abl & 32);
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        if (a == 1 && !second) break;
                        const int it = min(IPP * (ps + a) + upair, nitems);       // tail of the last pass: the pad item
                        const uint32_t item = items[it];
                        const unsigned char* sp = src + (__umul24(item & 0xffffffu, texel_bytes) + lane_src_off);
                        const int xr = min(x_base + (int)(item >> 26), p.w - 1);
                        const unsigned char* rp = ref_row + (__umul24((uint32_t)xr, texel_bytes) + (uint32_t)sub * 16u);
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) {
                            const bool okc = FULL || (sub + LPU * cc < nchunk);
                            sv[a][cc] = okc ? *reinterpret_cast<const uint4*>(sp + cc * CSTR) : make_uint4(0, 0, 0, 0);
                            if (PPW == 1) rv[a][cc] = rvp[cc];
                            else rv[a][cc] = okc ? *reinterpret_cast<const uint4*>(rp + cc * CSTR) : make_uint4(0, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        if (a == 1 && !second) break;
                        float part = 0.f;
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) part = cdot_chunk(rv[a][cc], sv[a][cc], part, FeatT());
                        part = LPU == 8 ? creduce8(part) : creduce4(part);
                        const int it = IPP * (ps + a) + upair;
                        if (sub == 0 && it < nitems) ctab[it * 4 + tap] = part;
                    }
                }
                wave_lds_fence();

                // ---------------- gate (speculative order), bilinear combine + fp64 view accumulation ----------------
                if (spec) eval_gate();
                if (gate) {
                    const float4 c4 = *reinterpret_cast<const float4*>(ctab + myitem * 4);
                    const float c = bilerp(c4.x, c4.y, c4.z, c4.w, t);
                    if (MODEF) accf = accf + c; else acc += (double)c;         // homography.py:42 / :159,116
                }
                wave_lds_fence();                                                 // ctab/items are rewritten by the next view
            }
            const float cval = (MODEF ? accf : (float)acc) / fV;               // homography.py:46 / :118,120
            if (p.cost_hi) {
                // split-bf16 channel-last output for the conv kernel: lanes = consecutive channels of one row
                if (live) {
                    const uint16_t hi = f32_to_bf16_rne(cval);
                    const uint16_t lo = f32_to_bf16_rne(cval - bf16_to_f32(hi));
                    const size_t e = (((size_t)b * Hp + (y + 1)) * Wp + (x + 1)) * (size_t)p.cost_ld + j;
                    p.cost_hi[e] = hi; p.cost_lo[e] = lo;
                }
                continue;
            }
            outb[(q & (OUT_PX - 1)) * DL + j0] = cval;
            if ((((qb + 1) * PPW) & (OUT_PX - 1)) == 0) {
                // ---- OUT_PX px x DL results: LDS -> coalesced row segments of cost[b, j, y, :] ----
                const int q_base = (qb + 1) * PPW - OUT_PX;
                wave_lds_fence();
                if (y < p.h) {
                    for (int e = lane; e < OUT_PX * DL; e += 64) {
                        const int qq = e & (OUT_PX - 1), jj = e / OUT_PX;
                        const int jo = jb * DL + jj, xo = x_base + q_base + qq;
                        if (jo < p.D && xo < p.w)
                            p.cost[(size_t)b * p.cost_bstride + (size_t)jo * hw + (size_t)y * p.w + xo] = outb[qq * DL + jj];
                    }
                }
                wave_lds_fence();
            }
        }
    }
    if (stats && tid == 0) atomicAdd(stats + 0, 1u);
}

template <int DL>
static size_t cand_lds_bytes(const CvParams& p) { return (size_t)4 * (p.V * 512 + 1024 + 272 + 8 * DL * 4); }

template <typename FeatT, int DL, int CPL, bool FULL, int MINW, int LPU = 8>
static hipError_t launch_cand(const CvParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    if (p.mode_f) hipLaunchKernelGGL((cv_cand_kernel<FeatT, DL, CPL, FULL, MINW, 1, LPU>), grid, block, cand_lds_bytes<DL>(p), stream, p);
    else if (!p.d_volume && !p.stats && !CV_DEV(p) && !p.gate_bits)
        hipLaunchKernelGGL((cv_cand_kernel<FeatT, DL, CPL, FULL, MINW, 2, LPU>), grid, block, cand_lds_bytes<DL>(p), stream, p);
    else hipLaunchKernelGGL((cv_cand_kernel<FeatT, DL, CPL, FULL, MINW, 0, LPU>), grid, block, cand_lds_bytes<DL>(p), stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL, int LPU = 8>
static hipError_t launch_cand_c(const CvParams& p, hipStream_t stream) {
    if (p.D <= 8)       return launch_cand<FeatT, 8, CPL, FULL, 4, LPU>(p, stream);
    else if (p.D <= 16) return launch_cand<FeatT, 16, CPL, FULL, 4, LPU>(p, stream);
    else if (p.D <= 32) return launch_cand<FeatT, 32, CPL, FULL, 4, LPU>(p, stream);
    // fp32 features carry twice the chunk registers: 6 waves/SIMD spills 76 B/lane there, 5 does not
    if (sizeof(FeatT) == 4) return launch_cand<FeatT, 64, CPL, FULL, 5, LPU>(p, stream);
    return launch_cand<FeatT, 64, CPL, FULL, 6, LPU>(p, stream);
}

hipError_t launch_cv_cand(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 24)) return hipSuccess;               // 24-bit texel index
    if ((size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets
    if (cand_lds_bytes<64>(p) > 64 * 1024) return hipSuccess;                                 // absurd V
    const int nchunk = (int)(p.F * esz / 16);
    *handled = true;
    if (p.feat_bf16) {
        if (nchunk == 8)  return launch_cand_c<uint16_t, 2, true, 4>(p, stream);     // F = 64: 4 lanes x 32 B per (item, tap) unit
        if (nchunk <= 8)  return launch_cand_c<uint16_t, 1, false>(p, stream);
        if (nchunk <= 16) return launch_cand_c<uint16_t, 2, false>(p, stream);
    } else {
        if (nchunk == 16) return launch_cand_c<float, 2, true>(p, stream);           // F = 64
        if (nchunk <= 8)  return launch_cand_c<float, 1, false>(p, stream);
        if (nchunk <= 16) return launch_cand_c<float, 2, false>(p, stream);
        if (nchunk <= 32) return launch_cand_c<float, 4, false>(p, stream);
    }
    *handled = false;                                                                 // very wide F: generic kernel
    return hipSuccess;
}

}  // namespace magnet
