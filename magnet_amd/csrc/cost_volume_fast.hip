// cost_volume_fast.hip — the PRODUCTION matcher: fused sampling + warp + gate + score with TOLERANCE parity.
//
// Same mapping as cost_volume_cand.hip (lane = depth candidate, a wave walks over 16 reference pixels of one
// tile row, distinct gate-open quads of a (pixel, view) become ITEMS, one F-channel dot product per (item, tap)),
// but the per-candidate arithmetic is no longer the reference's rounding sequence operation for operation.
// north_star's contract is abs_rel < 1e-4 on the final depth and SURVEY.md §7 allows a ~1e-5 fraction of flipped
// consistency gates; what this kernel computes instead of homography.py:131-148,157-159 / MAGNET.py:155:
//   d      = fma(sigma, k_j, mu)                         (reference: mul, then add)
//   P      = fma(r_pix, d, t_pix), z = fma(r_z, d, t_z)  (reference: mul, then add)
//   1/P_z  = v_rcp_f32 (1 ulp)                           (reference: IEEE division by P_z + 1e-10; the 1e-10 only
//            matters for |P_z| < ~1e-3, where the sample is far outside the image either way)
//   texel position directly in PADDED-map coordinates: ixs = fma(P_x, 1/P_z, 0.5) (= u - 0.5 + 1), no
//            normalise -> clamp(+-10) -> unnormalise round trip (homography.py:141-148 + ATen's unnormalise): every
//            clamped coordinate is out of the image, and so is every coordinate this kernel rejects by its window test
//   window test as one unsigned compare of the float's bit pattern per axis (0 <= ixs < w + 1)
//   views summed in fp32 (reference: fp64, homography.py:116-118,159), division by V as a multiplication by 1/V
// Modelled on the CPU against the oracle's gate bits (tools/flip_model.py): 4e-6 of the gates differ at C2; measured
// on the device by tests/test_gpu_fast_matcher.py (gate bits through the `gate_bits` debug output, value tolerance
// 2e-5 + 2e-5 |oracle| elsewhere, abs_rel of the refinement loop).
//
// Instruction diet relative to cost_volume_cand.hip (per (pixel, view) wave-iteration, bf16 F = 64, D = 64):
//   * frame / view base addresses are wave-uniform (readfirstlane) -> scalar address arithmetic, 32-bit lane offsets
//   * (mu,sigma) taps: every lane loads its own 2 x 16 B (no leader election, no ds_bpermute)
//   * no fp64, no exec-mask branches around the gate / combine
// Everything that needs the reference's exact rounding (explicit d_volume, est_costvolume_F mode, stats) stays in
// cost_volume_cand.hip / cost_volume.hip.
#include "cv_runs.hpp"

namespace magnet {

// Round 5: TEXEL-PAIR ITEMS (TX; see cost_volume_fast64.hip's header).  An item is a pair of texels across the direction in which the
// candidates of a (pixel, view) travel through the source map; a run's quad is two consecutive pairs, and a leader whose quad is the
// previous run's quad moved one step along the direction of travel re-uses that run's second pair.  Here several pixels share a wave
// (D <= 32), so mode and direction are per-lane values and the pairs are numbered in travel order (the combine mirrors its
// interpolation weight for a segment travelling towards smaller coordinates).
//
// DL   = candidates per pixel group inside a wave (8,16,32,64); PPW = 64/DL pixels per iteration
// CPL  = 16-byte channel chunks per lane in the VALU correlation (F*sizeof(FeatT)/16 <= LPU*CPL), FULL = exactly
// MINW = waves per SIMD to compile for; LPU = lanes per texel of the VALU correlation
// OPT  = bit 1: write gate bits (debug / parity tests); bit 2: texel-pair items (the product form; quad items are kept for dev A/B)
template <typename FeatT, int DL, int CPL, bool FULL, int MINW, int LPU, int OPT>
__global__ __launch_bounds__(256, MINW) void cv_fast_kernel(const CvParams p) {
    constexpr bool GBITS = (OPT & 2) != 0;
    constexpr bool TX = (OPT & 4) != 0;
    constexpr int TPI = TX ? 2 : 4;                       // texels per item
    constexpr int IPP = 64 / (TPI * LPU);                 // items per correlation pass
    constexpr int CSTR = LPU * 16;                        // byte stride between a lane's channel chunks (VALU correlation)
    constexpr int PPW = 64 / DL;
    constexpr int CT_BYTES = TX ? (128 + 2) * 8 : 16 + 1024;          // TX: [2 zero pairs | 128 pairs] x {c0, c1}; else [zero | 64 items] x 4 taps
    constexpr int IT_BYTES = TX ? (128 + 2) * 8 : 272;                // TX: {byte offset of texel 0, of texel 1} (+ pixel in the low bits); else texel index (+ pixel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_v, b_v;
    tile_of_block(p, tile_v, b_v);
    const int tile = __builtin_amdgcn_readfirstlane(tile_v), b = __builtin_amdgcn_readfirstlane(b_v);   // wave-uniform
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int y = ty * TILE_H + wv;                       // this wave's pixel row
    const int yc = min(y, p.h - 1);
    const int x_base = tx * TILE_W;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const int JB = (p.D + DL - 1) / DL;                   // candidate blocks per pixel

    // ---- wave-private LDS ----
    constexpr int OUT_PX = 8;                             // pixels staged before a coalesced flush (32-byte row segments)
    const int out_bytes = p.cost_hi ? 0 : OUT_PX * DL * 4;
    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    // PPW > 1 (D <= 32: several pixels per wave iteration): the reference vectors of the wave's 16 pixels live in LDS.  Read
    // from global memory per correlation pass they were half of this path's L1 traffic, and the path is bound by the L1 gather rate
    // (D = 5, fp32 features: ~40 B/clk/CU during the correlation phase against the 50 - 55 a pure L2-resident gather reaches)
    // Round 3: only the 8 pixels the next iterations work on are staged (re-staged once in the middle of the wave's 16): with 16 the
    // fp32 instances sat at 5 workgroups per CU by LDS, and the path is occupancy-sensitive (capped at 4 / 3: +11 % / +36 % time)
    const int ref_lds = PPW > 1 ? 8 * (int)texel_bytes : 0;
    const int md_bytes = TX ? p.V * 64 : 0;
    const int wave_bytes = p.V * 512 + CT_BYTES + IT_BYTES + out_bytes + md_bytes + ref_lds;
    unsigned char* wbase = smem + wv * wave_bytes;
    unsigned char* refl = wbase + (wave_bytes - ref_lds);                             // [8 px][texel_bytes]
    float4*   pvtab = reinterpret_cast<float4*>(wbase);                               // [V][16 px][2]
    float4*   ctab  = reinterpret_cast<float4*>(wbase + p.V * 512);                   // quad items: [zero slot for closed lanes | 64 items] x 4 taps
    float2*   ctab2 = reinterpret_cast<float2*>(wbase + p.V * 512);                   // pair items: [2 zero slots | 128 pairs] x 2 texels
    uint32_t* items = reinterpret_cast<uint32_t*>(wbase + p.V * 512 + CT_BYTES);      // quad items: [64 + pad]; pair items: [128 + pad] x 2
    float*    outb  = reinterpret_cast<float*>(wbase + p.V * 512 + CT_BYTES + IT_BYTES);   // [OUT_PX][DL] results of one block
    uint32_t* mtab  = reinterpret_cast<uint32_t*>(wbase + p.V * 512 + CT_BYTES + IT_BYTES + out_bytes);   // TX: [V][16 px] travel mode | signed step << 2

    // ---- depth-linear projection terms for the wave's 16 pixels x V views (once per tile row) ----
    for (int e = lane; e < 16 * p.V; e += 64) {
        const int q = e & 15, v = e >> 4;
        const int xc = min(x_base + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, yc, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        pvtab[e * 2 + 0] = make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz);
        pvtab[e * 2 + 1] = make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz);
        if (TX) {
            const uint32_t md = travel_mode(pv);
            const int step = ((md & 1u) ? Wp : 1) * ((md & 2u) ? -1 : 1);             // key of the next quad along the direction of travel - this quad's key
            mtab[e] = md | ((uint32_t)step << 2);
        }
    }
    if (TX) { if (lane < 2) ctab2[lane] = make_float2(0.f, 0.f); }
    else if (lane == 0) ctab[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    fwave_lds_fence();

    const int nchunk = (int)(texel_bytes / 16);
    // window 0 <= ixs < w + 1 (padded-map coordinates) as an unsigned compare of the float bits: negative values have the
    // sign bit set, NaNs are above every finite pattern
    const uint32_t xlim = __float_as_uint((float)(p.w + 1)), ylim = __float_as_uint((float)(p.h + 1));
    // lane roles
    const int g = lane / DL, j0 = lane % DL;                                      // geometry: pixel group, candidate
    const int sub = lane & (LPU - 1), tap = (lane / LPU) & (TPI - 1), upair = lane / (TPI * LPU);   // correlation: chunk, texel of the item, item of the pass
    const uint32_t lane_src_off = (TX ? 0u : (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes) + (uint32_t)sub * 16u;
    const uint32_t row_bytes = (uint32_t)Wp * texel_bytes;
    const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) +
        ((size_t)b * hw + (size_t)yc * p.w) * texel_bytes;                        // reference features of this pixel row
    const float invV = 1.0f / (float)p.V;
    // view validity (homography.py:97) as a bitmask read ONCE
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    // (mu, sigma) of the wave's 16 reference pixels, lane q holds pixel q
    float mu_row, sg_row;
    {
        const size_t pixr = (size_t)yc * p.w + min(x_base + (lane & 15), p.w - 1);
        mu_row = p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr];
        sg_row = p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr];
    }
    const size_t map_texels = (size_t)Hp * Wp;
    // source view v of frame b is image v*B + b (view-major, homography.py:105): walk the views by a constant stride
    const size_t src_vstride = (size_t)p.B * map_texels * texel_bytes, sgm_vstride = (size_t)p.B * map_texels * 8;
    // (round 4) frame bases pinned into SGPRs as GLOBAL pointers: base + 32-bit lane offset selects the scalar-base addressing mode; the
    // 64-bit multiply above runs on the vector unit, so these bases used to be per-lane register pairs (one v_lshl_add_u64 per load)
    const cvr_gptr src_b = (cvr_gptr)(unsigned long long)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_feat) + (size_t)b * map_texels * texel_bytes);
    const cvr_gptr sgm_b = (cvr_gptr)(unsigned long long)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_gmm) + (size_t)b * map_texels * 8);
    const float kappa = p.kappa;

    for (int jb = 0; jb < JB; ++jb) {                                             // candidate block of DL candidates
        const int j = jb * DL + j0;
        const float kj = p.k[min(j, p.D - 1)];
        for (int qb = 0; qb < 16 / PPW; ++qb) {
            if (PPW > 1 && ((qb * PPW) & 7) == 0) {
                // reference vectors of the next 8 pixels (contiguous in the channel-last row) -> LDS
                fwave_lds_fence();                                                // the previous 8 pixels' correlation reads are done
                const int q8 = qb * PPW;
                for (int e = lane; e < 8 * nchunk; e += 64) {
                    const int qq = e / nchunk, c = e - qq * nchunk;
                    const int xr = min(x_base + q8 + qq, p.w - 1);
                    *reinterpret_cast<uint4*>(refl + e * 16) = *reinterpret_cast<const uint4*>(ref_row + (__umul24((uint32_t)xr, texel_bytes) + (uint32_t)c * 16u));
                }
                fwave_lds_fence();
            }
            const int q = qb * PPW + g;                                           // pixel within the wave's row
            const int x = x_base + q;
            const bool live = (x < p.w) && (y < p.h) && (j < p.D);
            // PPW == 1: every correlation unit of this iteration belongs to this one pixel: its reference vector (the lane's chunks)
            // stays in registers for all views
            uint4 rvp[CPL];
            if (PPW == 1) {
                const unsigned char* rp = ref_row + (__umul24((uint32_t)min(x, p.w - 1), texel_bytes) + (uint32_t)sub * 16u);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc)
                    rvp[cc] = (FULL || (sub + LPU * cc < nchunk)) ? *reinterpret_cast<const uint4*>(rp + cc * CSTR) : make_uint4(0, 0, 0, 0);
            }
            float d;
            {
                const float mu = PPW == 1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu_row), qb)) : __shfl(mu_row, q);
                const float sg = PPW == 1 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg_row), qb)) : __shfl(sg_row, q);
                d = __builtin_fmaf(sg, kj, mu);                                   // MAGNET.py:155
            }
            d = live ? d : __builtin_nanf("");                                    // dead lane -> out of image below
            float acc = 0.f;

            for (int v = 0; v < p.V; ++v) {
                if (!((vmask >> v) & 1ull)) continue;                            // homography.py:97 (wave-uniform)
                // the view's bases, pinned again: the 64-bit view stride is a vector-unit product
                const cvr_gptr src = (cvr_gptr)(unsigned long long)v4_uniform_ptr((const void*)(src_b + (size_t)v * src_vstride));
                const cvr_gptr sgm = (cvr_gptr)(unsigned long long)v4_uniform_ptr((const void*)(sgm_b + (size_t)v * sgm_vstride));
                // ---------------- geometry ----------------
                const float4 pa = pvtab[(v * 16 + q) * 2 + 0], pb = pvtab[(v * 16 + q) * 2 + 1];
                const uint32_t md = TX ? mtab[v * 16 + q] : 0u;
                const float Px = __builtin_fmaf(pa.x, d, pb.x);                  // homography.py:132
                const float Py = __builtin_fmaf(pa.y, d, pb.y);
                const float Pz = __builtin_fmaf(pa.z, d, pb.z);
                const float zw = __builtin_fmaf(pa.w, d, pb.w);                  // homography.py:137-138
                const float rz = __builtin_amdgcn_rcpf(Pz);                      // homography.py:133
                const float ixs = __builtin_fmaf(Px, rz, 0.5f);                  // = (u - 0.5) + 1: padded-map texel coordinate
                const float iys = __builtin_fmaf(Py, rz, 0.5f);
                const float x0f = __builtin_floorf(ixs), y0f = __builtin_floorf(iys);
                const float bx = ixs - x0f, by = iys - y0f;
                const float ax = 1.0f - bx, ay = 1.0f - by;
                const float wnw = ax * ay, wne = bx * ay, wsw = ax * by, wse = bx * by;   // homography.py:150-152
                const bool inwin = (__float_as_uint(ixs) < xlim) && (__float_as_uint(iys) < ylim);
                const uint32_t qi = inwin ? (uint32_t)__mul24((int)y0f, Wp) + (uint32_t)(int)x0f : 0u;   // quad origin, padded map
                // ---------------- (mu,sigma) taps + consistency gate ----------------
                const float4 g0 = v3_gld_f4(sgm + qi * 8u);                      // (mu,sg) x0, x0+1 of row y0
                const float4 g1 = v3_gld_f4(sgm + (qi + (uint32_t)Wp) * 8u);
                float mu_w = g0.x * wnw, sg_w = g0.y * wnw;
                mu_w = __builtin_fmaf(g0.z, wne, mu_w); sg_w = __builtin_fmaf(g0.w, wne, sg_w);
                mu_w = __builtin_fmaf(g1.x, wsw, mu_w); sg_w = __builtin_fmaf(g1.y, wsw, sg_w);
                mu_w = __builtin_fmaf(g1.z, wse, mu_w); sg_w = __builtin_fmaf(g1.w, wse, sg_w);
                const bool gate = inwin && (__builtin_fabsf(zw - mu_w) < sg_w * kappa);          // homography.py:157-158
                if (GBITS && live)
                    p.gate_bits[(((size_t)b * p.V + v) * p.D + j) * hw + (size_t)y * p.w + x] = gate ? 1 : 0;

                // ---------------- distinct open quads of the wave -> items ----------------
                const uint32_t key = gate ? qi : FKEY_CLOSED;
                uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)FKEY_CLOSED, (int)key, 0x138, 0xf, 0xf, false);  // wave_shr:1
                if (PPW > 1 && j0 == 0) prev = FKEY_CLOSED;                       // first candidate of a pixel group
                const bool fresh = gate && (key != prev);
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(fresh);
                if (bal == 0ull) continue;                                        // wave-uniform: nothing open in this view
                int nitems = __popcll(bal);
                int incl = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                __builtin_amdgcn_mbcnt_lo((uint32_t)bal, fresh ? 1u : 0u));         // leaders at or below this lane
                float fmin = 0.f, fmaj = 0.f;
                if (TX) {
                    const bool rowm = (md & 1u) != 0, neg = (md & 2u) != 0;
                    // a leader whose quad is the previous run's quad moved one step along the direction of travel shares that run's
                    // second pair (FKEY_CLOSED + a step is no valid key)
                    const bool shr = fresh && (prev + (uint32_t)((int)md >> 2) == key);
                    const unsigned long long sbal = __builtin_amdgcn_ballot_w64(shr);
                    const int ns = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sbal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sbal, shr ? 1u : 0u));
                    nitems = 2 * nitems - __popcll(sbal);                         // pairs of this view
                    incl = 2 * (incl - 1) - ns;                                   // the lane's first pair, numbered in travel order
                    const uint32_t oJ = rowm ? row_bytes : texel_bytes, oM = rowm ? texel_bytes : row_bytes;
                    const uint32_t o0 = __umul24(qi, texel_bytes) + (PPW > 1 ? (uint32_t)(q & 7) : 0u);   // (the pixel -> its reference vector in LDS)
                    const uint32_t oF = o0 + (neg ? oJ : 0u), oS = o0 + (neg ? 0u : oJ);           // first / second pair in travel order
                    if (fresh && !shr) *reinterpret_cast<uint2*>(items + incl * 2) = make_uint2(oF, oF + oM);
                    if (fresh) *reinterpret_cast<uint2*>(items + (incl + 1) * 2) = make_uint2(oS, oS + oM);
                    if (lane == 0) *reinterpret_cast<uint2*>(items + nitems * 2) = make_uint2(0u, 0u);   // pad to a whole pass: pixel 0, texel 0
                    incl = gate ? incl + 2 : 0;                                   // closed lanes: the zero pairs
                    fmin = rowm ? bx : by; fmaj = rowm ? by : bx;
                    fmaj = neg ? 1.0f - fmaj : fmaj;                              // pairs are numbered in travel order
                } else {
                    if (fresh) items[incl - 1] = PPW == 1 ? qi : (((uint32_t)q << 26) | qi);
                    if (lane == 0) items[nitems] = 0u;                            // pad to a whole pass: pixel 0, texel 0
                }
                fwave_lds_fence();

                // ---------------- correlation ----------------
                {
                    const int passes = (nitems + IPP - 1) / IPP;
                    for (int ps = 0; ps < passes; ps += 2) {
                        uint4 sv[2][CPL], rv[2][CPL];
                        const bool second = IPP * (ps + 1) < nitems;             // second pass of the pair only if it holds an item
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            if (a == 1 && !second) break;
                            const int it = min(IPP * (ps + a) + upair, nitems);   // tail of the last pass: the pad item
                            const uint32_t item = TX ? items[it * 2 + tap] : items[it];
                            const cvr_gptr sp = src + ((TX ? (item & ~15u) : __umul24(item & 0xffffffu, texel_bytes)) + lane_src_off);
                            const unsigned char* rp = refl + (__umul24(TX ? (item & 7u) : ((item >> 26) & 7u), texel_bytes) + (uint32_t)sub * 16u);   // LDS (PPW > 1 only)
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) {
                                const bool okc = FULL || (sub + LPU * cc < nchunk);
                                sv[a][cc] = okc ? v3_gld_u4(sp + cc * CSTR) : make_uint4(0, 0, 0, 0);
                                if (PPW == 1) rv[a][cc] = rvp[cc];
                                else rv[a][cc] = okc ? *reinterpret_cast<const uint4*>(rp + cc * CSTR) : make_uint4(0, 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            if (a == 1 && !second) break;
                            float part = 0.f;
#pragma unroll
                            for (int cc = 0; cc < CPL; ++cc) part = fdot_chunk(rv[a][cc], sv[a][cc], part, FeatT());
                            part = LPU == 8 ? freduce8(part) : freduce4(part);
                            const int it = IPP * (ps + a) + upair;
                            if (sub == 0 && it < nitems) {
                                if (TX) reinterpret_cast<float*>(ctab2 + it + 2)[tap] = part;
                                else reinterpret_cast<float*>(ctab + it + 1)[tap] = part;
                            }
                        }
                    }
                }
                fwave_lds_fence();

                // ---------------- bilinear combine + view accumulation ----------------
                if (TX) {
                    const float2 cA = ctab2[incl], cB = ctab2[incl + 1];
                    const float t = __builtin_fmaf(fmin, cA.y - cA.x, cA.x), l = __builtin_fmaf(fmin, cB.y - cB.x, cB.x);
                    const float c = __builtin_fmaf(fmaj, l - t, t);               // homography.py:150,155 (grid_sample's bilinear weights, factored)
                    acc += gate ? c : 0.f;                                        // homography.py:159,116 (fp32 here); the fractions of a closed lane may be NaN
                } else {
                    const float4 c4 = ctab[gate ? incl : 0];                      // closed lanes: the zero slot
                    float c = c4.x * wnw;
                    c = __builtin_fmaf(c4.y, wne, c);
                    c = __builtin_fmaf(c4.z, wsw, c);
                    c = __builtin_fmaf(c4.w, wse, c);
                    acc += gate ? c : 0.f;                                        // homography.py:159,116 (fp32 here); the weights of a closed lane may be NaN
                }
                fwave_lds_fence();                                                // ctab/items are rewritten by the next view
            }
            const float cval = acc * invV;                                        // homography.py:118,120
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
                fwave_lds_fence();
                if (y < p.h) {
                    for (int e = lane; e < OUT_PX * DL; e += 64) {
                        const int qq = e & (OUT_PX - 1), jj = e / OUT_PX;
                        const int jo = jb * DL + jj, xo = x_base + q_base + qq;
                        if (jo < p.D && xo < p.w)
                            p.cost[(size_t)b * p.cost_bstride + (size_t)jo * hw + (size_t)y * p.w + xo] = outb[qq * DL + jj];
                    }
                }
                fwave_lds_fence();
            }
        }
    }
}

template <int DL>
static size_t fast_lds_bytes(const CvParams& p, bool tx = true) {
    const size_t tables = tx ? (size_t)(130 * 8 + 130 * 8 + p.V * 64) : (size_t)(16 + 1024 + 272);
    return (size_t)4 * (p.V * 512 + tables + (p.cost_hi ? 0 : 8 * DL * 4) + (DL < 64 ? 8 * p.F * (p.feat_bf16 ? 2 : 4) : 0));
}

template <typename FeatT, int DL, int CPL, bool FULL, int MINW, int LPU>
static hipError_t launch_fast(const CvParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
#ifdef MAGNET_DEV
    if (CV_DEV(p) & 0x400) {                                                       // dev: quad items (rounds 2 - 4) for same-box A/B
        const size_t lq = fast_lds_bytes<DL>(p, false);
        if (p.gate_bits) hipLaunchKernelGGL((cv_fast_kernel<FeatT, DL, CPL, FULL, MINW, LPU, 2>), grid, block, lq, stream, p);
        else hipLaunchKernelGGL((cv_fast_kernel<FeatT, DL, CPL, FULL, MINW, LPU, 0>), grid, block, lq, stream, p);
        return hipGetLastError();
    }
#endif
    size_t lds = fast_lds_bytes<DL>(p);
#ifdef MAGNET_DEV
    {   // dev: cap the workgroups per CU by asking for more LDS than the kernel uses (occupancy sensitivity)
        const int cap = (CV_DEV(p) & 0x300000) == 0x300000 ? 3 : (CV_DEV(p) & 0x200000) ? 4 : 0;
        if (cap) { const size_t need = (size_t)160 * 1024 / (cap + 1) + 512; if (lds < need) lds = need; }
    }
#endif
    if (p.gate_bits) hipLaunchKernelGGL((cv_fast_kernel<FeatT, DL, CPL, FULL, MINW, LPU, 6>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((cv_fast_kernel<FeatT, DL, CPL, FULL, MINW, LPU, 4>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL, int LPU>
static hipError_t launch_fast_c(const CvParams& p, hipStream_t stream) {
    if (p.D <= 8)       return launch_fast<FeatT, 8, CPL, FULL, 4, LPU>(p, stream);
    else if (p.D <= 16) return launch_fast<FeatT, 16, CPL, FULL, 4, LPU>(p, stream);
    else if (p.D <= 32) return launch_fast<FeatT, 32, CPL, FULL, 4, LPU>(p, stream);
    if constexpr (sizeof(FeatT) == 4) return launch_fast<FeatT, 64, CPL, FULL, 6, LPU>(p, stream);
    else return launch_fast<FeatT, 64, CPL, FULL, 6, LPU>(p, stream);          // (8 waves per SIMD needed scratch)
}

// Production matcher: fused candidate sampling only (no explicit d_volume), mode 0, no stats counters.
hipError_t launch_cv_fast(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    if (p.d_volume || p.mode_f || p.stats) return hipSuccess;
    const bool have_gmm = p.src_gmm != nullptr;               // the kernels below read the interleaved map
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 24)) return hipSuccess;               // 24-bit texel index
    if ((size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets
    if (fast_lds_bytes<64>(p) > 64 * 1024 || (p.D <= 32 && fast_lds_bytes<32>(p) > 64 * 1024)) return hipSuccess;   // absurd V (D <= 32: + the reference vectors)
    if (p.src_gmq && !(CV_DEV(p) & 0x100)) {                                                  // D > 32 with the quad-form (mu, sigma) map: the round-3 kernel (dev bit 0x100: the round-2 kernels)
#ifdef MAGNET_DEV
        // round 4's two measured experiments, dev library only (profiles/r4/NOTES.md): 0x8 = quads AND texels staged in LDS by DMA, correlation on
        // the matrix pipe (cost_volume_v4.hip: 1.24 - 1.35 ms); 0x20 = round 3's kernel with the quads prefetched one unit ahead (cost_volume_v5.hip: 1.11 ms)
        if (CV_DEV(p) & 0x8) {
            const hipError_t e4 = launch_cv_v4(p, stream, handled);
            if (e4 != hipSuccess || *handled) return e4;
        }
        if (CV_DEV(p) & 0x20) {
            const hipError_t e5 = launch_cv_v5(p, stream, handled);
            if (e5 != hipSuccess || *handled) return e5;
        }
#endif
        const hipError_t e = launch_cv_v3(p, stream, handled);
        if (e != hipSuccess || *handled) return e;
    }
    if (!have_gmm) return hipSuccess;                                                         // (api.hip reports MAGNET_E_SHAPE)
    if (!(CV_DEV(p) & 0x800)) {                                                                // D > 32: views batched per pixel (cost_volume_fast64.hip); dev 0x800: the per-view kernel below
        const hipError_t e = launch_cv_fast64(p, stream, handled);
        if (e != hipSuccess || *handled) return e;
    }
    const int nchunk = (int)(p.F * esz / 16);
    *handled = true;
    if (p.feat_bf16) {
        if (nchunk == 8)  return launch_fast_c<uint16_t, 2, true, 4>(p, stream);             // F = 64: 4 lanes x 32 B per texel
        if (nchunk <= 8)  return launch_fast_c<uint16_t, 1, false, 8>(p, stream);
        if (nchunk <= 16) return launch_fast_c<uint16_t, 2, false, 8>(p, stream);
    } else {
        if (nchunk == 16) return launch_fast_c<float, 2, true, 8>(p, stream);                // F = 64
        if (nchunk <= 8)  return launch_fast_c<float, 1, false, 8>(p, stream);
        if (nchunk <= 16) return launch_fast_c<float, 2, false, 8>(p, stream);
        if (nchunk <= 32) return launch_fast_c<float, 4, false, 8>(p, stream);
    }
    *handled = false;                                                                         // very wide F: exact kernels
    return hipSuccess;
}

}  // namespace magnet
