// cost_volume_fast64.hip — production matcher for D > 32 (one reference pixel per wave iteration, lane = candidate):
// the source VIEWS of a pixel are processed in groups of VG, phase by phase, instead of one after the other.
//
// Why.  Counters and ablations of the per-view kernel (cost_volume_fast.hip, profiles/r2/): it is neither VALU- nor
// HBM-bound — 86 % of the wave cycles are s_waitcnt.  Each (pixel, view) iteration is a chain of dependent round trips
// (projection table from LDS -> geometry -> (mu,sigma) taps from L2 -> gate -> item list through LDS -> feature texels
// from L2 -> dot products -> result table through LDS -> combine) and a wave has exactly one of them in flight; with the
// synthetic inputs of SURVEY.md §8d (per-pixel independent depths) every (item, tap) is a different 128-byte line, so the
// kernel moves ~8.8 GB per launch from the L2s to the L1s and needs many more loads in flight than 8 waves x 1 provide
// ("no correlation" ablation: 0.46 ms of 1.09 ms).  Here the VG views of a pixel run their geometry / tap loads / gates
// back to back (independent chains, interleaved by the compiler), their open quads go into ONE item list, the (item, tap)
// dot products of the whole group are fetched in batches of up to 16 items (8 x 1 KiB loads in flight per wave), then the
// VG bilinear combines run: two memory round trips per GROUP instead of two per view.
//
// Arithmetic, tolerance contract, layouts: exactly cost_volume_fast.hip (see its header).
//
// Round 5: TEXEL-PAIR ITEMS (TX).  The candidates of a pixel walk along the view's epipolar segment, so the quads of consecutive
// gate-open runs are neighbours and share two of their four texels (homography.py:150: grid_sample's 2 x 2 footprint): with quad
// items 4 texels per run were fetched and correlated where ~2.2 distinct ones exist on the full-resolution grids (14 runs per
// (pixel, view) at C2L).  Now an item is a PAIR of texels across the segment's direction of travel — a column {(y0, x), (y0+1, x)}
// when the segment runs along x, a row {(y, x0), (y, x0+1)} when it runs along y.  The direction is the sign of the d-independent
// numerator of d(P_x/P_z)/dd = (r_x t_z - t_x r_z) / P_z^2 (likewise y), evaluated once per (pixel, view) in the prologue.  A run's
// quad is two consecutive pairs; a leader whose quad is the previous run's quad + one step along the direction of travel re-uses
// that run's second pair and emits only one.  Slot numbering runs along increasing x (y), whatever the direction of travel, so the
// combine is direction-free: c = lerp(lerp(pair s), lerp(pair s + 1)).  The mode only decides how many pairs are shared, never the
// result: any (mode, direction) gives the same texels to the same quads.
#include <stdlib.h>
#include "cv_runs.hpp"

namespace magnet {

// CPL / FULL / LPU: as in cv_fast_kernel (VALU correlation units of LPU lanes x CPL 16-byte chunks)
// VG = views per group (1..4); OPT bit 0 = write the gate bits (debug / parity tests); bit 2 = texel-pair items (TX; the product form —
// quad items are kept for dev A/B only)
template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG, int OPT>
__global__ __launch_bounds__(256, MINW) void cv_fast64_kernel(const CvParams p) {
    constexpr int DL = 64;
    constexpr bool GBITS = (OPT & 1) != 0;
    constexpr bool TX = (OPT & 4) != 0;
    constexpr int TPI = TX ? 2 : 4;                       // texels per item
    constexpr int IPP = 64 / (TPI * LPU);                 // items per correlation pass
    constexpr int NPASS = TX ? (32 / IPP > 4 ? 4 : 32 / IPP) : (16 / IPP > 4 ? 4 : 16 / IPP);   // passes fetched together (<= 4)
    constexpr int CSTR = LPU * 16;                        // byte stride between a lane's channel chunks
    constexpr int CAP = TX ? 128 : 64;                    // item capacity of the LDS tables (one view alone never needs more)
    constexpr int CT_BYTES = TX ? (CAP + 2) * 8 : (CAP + 1) * 16;     // TX: [2 zero pairs | CAP pairs] x {c0, c1}; else [zero | CAP] x 4 taps
    constexpr int IT_BYTES = TX ? (CAP + 2) * 8 : (CAP + 4) * 4;      // TX: {byte offset of texel 0, of texel 1}; else byte offset of the quad
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_v, b_v;
    tile_of_block(p, tile_v, b_v);
    const int tile = __builtin_amdgcn_readfirstlane(tile_v), b = __builtin_amdgcn_readfirstlane(b_v);   // wave-uniform
    // workgroup = 4 waves side by side in ONE pixel row, NPX consecutive pixels each: blocks walk the frame in raster order, so
    // the pixels an XCD works on at any time are a band of a few rows whose source footprint (band + disparity halo, all
    // views) stays inside its 4 MiB L2 (see the launcher)
    const int NPX = p.npx;
    int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    if (p.strip_tx > 0) {                                 // vertical strips of strip_tx tiles, each walked in raster order (the last one may be narrower)
        const int per_strip = p.strip_tx * p.tiles_y;
        const int s = min(tile / per_strip, (p.tiles_x - 1) / p.strip_tx), r = tile - s * per_strip;
        const int sw = min(p.strip_tx, p.tiles_x - s * p.strip_tx);
        ty = r / sw; tx = s * p.strip_tx + (r - ty * sw);
    }
    const int y = ty;                                     // this wave's pixel row
    const int yc = min(y, p.h - 1);
    const int x_base = (tx * 4 + wv) * NPX;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const int JB = (p.D + DL - 1) / DL;                   // candidate blocks per pixel

    // ---- wave-private LDS ----
    const int OUT_PX = min(8, NPX);                       // pixels staged before a flush of row segments (NCHW fp32 output only)
    const int out_bytes = p.cost_hi ? 0 : OUT_PX * DL * 4;
    const int pv_bytes = p.V * NPX * 32;
    const int md_bytes = TX ? (p.V * NPX * 4 + 15) / 16 * 16 : 0;
    const int wave_bytes = pv_bytes + CT_BYTES + IT_BYTES + out_bytes + md_bytes;
    unsigned char* wbase = smem + wv * wave_bytes;
    float4*   pvtab = reinterpret_cast<float4*>(wbase);                                         // [V][NPX px][2]
    float4*   ctab  = reinterpret_cast<float4*>(wbase + pv_bytes);                              // quad items: [zero slot | CAP items] x 4 taps
    float2*   ctab2 = reinterpret_cast<float2*>(wbase + pv_bytes);                              // pair items: [2 zero slots | CAP pairs] x 2 texels
    uint32_t* items = reinterpret_cast<uint32_t*>(wbase + pv_bytes + CT_BYTES);                 // quad items: byte offsets; pair items: 2 per pair
    float*    outb  = reinterpret_cast<float*>(wbase + pv_bytes + CT_BYTES + IT_BYTES);         // [OUT_PX][DL]
    uint32_t* mtab  = reinterpret_cast<uint32_t*>(wbase + pv_bytes + CT_BYTES + IT_BYTES + out_bytes);   // TX: [V][NPX] bit 0 = travels along y, bit 1 = towards smaller coordinates

    // ---- depth-linear projection terms for the wave's 16 pixels x V views (once per tile row) ----
    for (int e = lane; e < NPX * p.V; e += 64) {
        const int q = e % NPX, v = e / NPX;
        const int xc = min(x_base + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, yc, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        pvtab[e * 2 + 0] = make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz);
        pvtab[e * 2 + 1] = make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz);
        if (TX) mtab[e] = travel_mode(pv);
    }
    if (TX) { if (lane < 2) ctab2[lane] = make_float2(0.f, 0.f); }
    else if (lane == 0) ctab[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    fwave_lds_fence();

    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    const int nchunk = (int)(texel_bytes / 16);
    const uint32_t xlim = __float_as_uint((float)(p.w + 1)), ylim = __float_as_uint((float)(p.h + 1));   // see cost_volume_fast.hip
    const int j0 = lane;
    const int sub = lane & (LPU - 1), tap = (lane / LPU) & (TPI - 1), upair = lane / (TPI * LPU);   // correlation: chunk, texel of the item, item of the pass
    const uint32_t lane_src_off = (TX ? 0u : (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes) + (uint32_t)sub * 16u;
    const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) +
        ((size_t)b * hw + (size_t)yc * p.w) * texel_bytes;
    const float invV = 1.0f / (float)p.V;
    unsigned long long vmask = 0ull;                                              // homography.py:97, read once
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    float mu_row, sg_row;                                                         // lane q holds reference pixel q of the row
    {
        const size_t pixr = (size_t)yc * p.w + min(x_base + (lane & 15), p.w - 1);      // NPX <= 16
        mu_row = p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr];
        sg_row = p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr];
    }
    const size_t map_texels = (size_t)Hp * Wp;
    // source view v of frame b is image v*B + b (view-major, homography.py:105); byte offsets inside the frame's views fit
    // 32 bits (checked by the launcher)
    const uint32_t src_vstride = (uint32_t)((size_t)p.B * map_texels * texel_bytes);
    const size_t sgm_vstride = (size_t)p.B * map_texels * 8;
    // (round 4) frame bases pinned into SGPRs as GLOBAL pointers (scalar-base addressing mode of the vector loads; see cost_volume_v3.hip)
    const cvr_gptr src_b = (cvr_gptr)(unsigned long long)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_feat) + (size_t)b * map_texels * texel_bytes);
    const cvr_gptr sgm_b = (cvr_gptr)(unsigned long long)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_gmm) + (size_t)b * map_texels * 8);
    const float kappa = p.kappa;
    const uint32_t row_bytes = (uint32_t)Wp * texel_bytes;

    for (int jb = 0; jb < JB; ++jb) {                                             // candidate block of 64 candidates
        const int j = jb * DL + j0;
        const float kj = p.k[min(j, p.D - 1)];
        for (int q = 0; q < NPX; ++q) {                                           // pixel within the wave's row segment
            const int x = x_base + q;
            const bool live = (x < p.w) && (y < p.h) && (j < p.D);
            // the pixel's reference vector: this lane's chunk(s), kept in registers for all views
            uint4 rvp[CPL];
            {
                const unsigned char* rp = ref_row + (__umul24((uint32_t)min(x, p.w - 1), texel_bytes) + (uint32_t)sub * 16u);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc)
                    rvp[cc] = (FULL || (sub + LPU * cc < nchunk)) ? *reinterpret_cast<const uint4*>(rp + cc * CSTR) : make_uint4(0, 0, 0, 0);
            }
            float d;
            {
                const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mu_row), q));
                const float sg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sg_row), q));
                d = __builtin_fmaf(sg, kj, mu);                                   // MAGNET.py:155
            }
            d = live ? d : __builtin_nanf("");                                    // dead lane -> out of image below
            float acc = 0.f;

            // texel dot products of items [0, n) of the LDS list -> their slots (quad items: ctab[1 + item][tap]; pair items: ctab2[2 + item].{x, y})
            auto correlate = [&](const int n) {
                for (int ps = 0; ps < n; ps += IPP * NPASS) {
                    uint4 sv[NPASS][CPL];
                    uint32_t off[NPASS];
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {                             // tail: the pad item
                        const int it = min(ps + IPP * a + upair, n);
                        off[a] = TX ? items[it * 2 + tap] : items[it];
                    }
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {
                        if (a > 0 && ps + IPP * a >= n) break;                    // wave-uniform: this pass holds no item
                        const cvr_gptr sp = src_b + (off[a] + lane_src_off);
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc)
                            sv[a][cc] = (FULL || (sub + LPU * cc < nchunk)) ? v3_gld_u4(sp + cc * CSTR) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {
                        if (a > 0 && ps + IPP * a >= n) break;
                        float part = 0.f;
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) part = fdot_chunk(rvp[cc], sv[a][cc], part, FeatT());
                        part = LPU == 8 ? freduce8(part) : freduce4(part);
                        const int it = ps + IPP * a + upair;
                        if (sub == 0 && it < n) {
                            if (TX) reinterpret_cast<float*>(ctab2 + it + 2)[tap] = part;
                            else reinterpret_cast<float*>(ctab + it + 1)[tap] = part;
                        }
                    }
                }
            };

            for (int v0 = 0; v0 < p.V; v0 += VG) {
                // ---------------- phase A: geometry, (mu,sigma) taps, gate, distinct open quads — VG independent chains ----------------
                float wa[VG], wb[VG], wc[VG], wd[VG];                              // quad items: the four bilinear weights; pair items: {bx, by}
                uint32_t qoff[VG];                                                // byte offset of the quad's first texel in the frame's views
                int incl[VG], cnt[VG];                                            // quad items: items at or below the lane / of the view; pair items: first slot (travel order) / pairs of the view
                bool gate[VG], fresh[VG], shr[VG];
                uint32_t md[VG];
                float zw[VG];
                uint32_t qi[VG];
                bool inwin[VG];
                float4 g0[VG], g1[VG];
                {
                    // A1: the VG projection-table reads, then A2: geometry and the (mu,sigma) tap loads of all VG views — issued
                    // unconditionally (quad origin clamped into the padded map, so out-of-window lanes read valid memory) so that
                    // the compiler does not wrap each view's loads into an exec-masked region with its own vmcnt(0)
                    float4 pa[VG], pb[VG];
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        const int vv = min(v0 + u, p.V - 1);                      // tail group: clamped view, masked below
                        pa[u] = pvtab[(vv * NPX + q) * 2 + 0]; pb[u] = pvtab[(vv * NPX + q) * 2 + 1];
                        if (TX) md[u] = __builtin_amdgcn_readfirstlane(mtab[vv * NPX + q]);
                    }
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        const int vv = min(v0 + u, p.V - 1);
                        const float Px = __builtin_fmaf(pa[u].x, d, pb[u].x);     // homography.py:132
                        const float Py = __builtin_fmaf(pa[u].y, d, pb[u].y);
                        const float Pz = __builtin_fmaf(pa[u].z, d, pb[u].z);
                        zw[u] = __builtin_fmaf(pa[u].w, d, pb[u].w);              // homography.py:137-138
                        const float rz = __builtin_amdgcn_rcpf(Pz);               // homography.py:133
                        const float ixs = __builtin_fmaf(Px, rz, 0.5f);           // = (u - 0.5) + 1: padded-map texel coordinate
                        const float iys = __builtin_fmaf(Py, rz, 0.5f);
                        const float x0f = __builtin_floorf(ixs), y0f = __builtin_floorf(iys);
                        const float bx = ixs - x0f, by = iys - y0f;
                        const float ax = 1.0f - bx, ay = 1.0f - by;
                        wa[u] = ax * ay; wb[u] = bx * ay; wc[u] = ax * by; wd[u] = bx * by;   // homography.py:150-152
                        inwin[u] = (__float_as_uint(ixs) < xlim) && (__float_as_uint(iys) < ylim);
                        // clamp in float first (v_med3_f32; NaN -> 0): the float -> unsigned conversion is then defined
                        const uint32_t xq = (uint32_t)__builtin_amdgcn_fmed3f(x0f, 0.0f, (float)p.w);
                        const uint32_t yq = (uint32_t)__builtin_amdgcn_fmed3f(y0f, 0.0f, (float)p.h);
                        qi[u] = __umul24(yq, (uint32_t)Wp) + xq;                 // quad origin in the padded map (exact when inwin)
                        const cvr_gptr sgm = (cvr_gptr)(unsigned long long)v4_uniform_ptr((const void*)(sgm_b + (size_t)vv * sgm_vstride));
                        g0[u] = v3_gld_f4(sgm + qi[u] * 8u);                     // (mu,sg) x0, x0+1 of row y0
                        g1[u] = v3_gld_f4(sgm + (qi[u] + (uint32_t)Wp) * 8u);
                        if (TX) {                                                 // the gate below needs the four weights, the combine only the fractions
                            const float mu_w = __builtin_fmaf(g1[u].z, wd[u], __builtin_fmaf(g1[u].x, wc[u], __builtin_fmaf(g0[u].z, wb[u], g0[u].x * wa[u])));
                            const float sg_w = __builtin_fmaf(g1[u].w, wd[u], __builtin_fmaf(g1[u].y, wc[u], __builtin_fmaf(g0[u].w, wb[u], g0[u].y * wa[u])));
                            wc[u] = mu_w; wd[u] = sg_w; wa[u] = bx; wb[u] = by;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    // A3: consistency gate, distinct open quads
                    const int vv = min(v0 + u, p.V - 1);
                    const bool vok = (v0 + u < p.V) && ((vmask >> vv) & 1ull);    // homography.py:97 (wave-uniform)
                    float mu_w, sg_w;
                    if (TX) { mu_w = wc[u]; sg_w = wd[u]; }
                    else {
                        mu_w = g0[u].x * wa[u]; sg_w = g0[u].y * wa[u];
                        mu_w = __builtin_fmaf(g0[u].z, wb[u], mu_w); sg_w = __builtin_fmaf(g0[u].w, wb[u], sg_w);
                        mu_w = __builtin_fmaf(g1[u].x, wc[u], mu_w); sg_w = __builtin_fmaf(g1[u].y, wc[u], sg_w);
                        mu_w = __builtin_fmaf(g1[u].z, wd[u], mu_w); sg_w = __builtin_fmaf(g1[u].w, wd[u], sg_w);
                    }
                    gate[u] = vok & inwin[u] & (__builtin_fabsf(zw[u] - mu_w) < sg_w * kappa);     // homography.py:157-158
                    if (GBITS && live && vok)
                        p.gate_bits[(((size_t)b * p.V + vv) * p.D + j) * hw + (size_t)y * p.w + x] = gate[u] ? 1 : 0;
                    qoff[u] = (uint32_t)vv * src_vstride + __umul24(qi[u], texel_bytes);
                    const uint32_t key = gate[u] ? qi[u] : FKEY_CLOSED;
                    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)FKEY_CLOSED, (int)key, 0x138, 0xf, 0xf, false);  // wave_shr:1
                    fresh[u] = gate[u] && (key != prev);
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(fresh[u]);
                    const int nl = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                        __builtin_amdgcn_mbcnt_lo((uint32_t)bal, fresh[u] ? 1u : 0u));   // leaders at or below this lane
                    if (TX) {
                        // a leader whose quad is the previous run's quad moved one step along the direction of travel shares that run's
                        // second pair (FKEY_CLOSED +- a step is no valid key)
                        const uint32_t step = ((md[u] & 1u) ? (uint32_t)Wp : 1u) * ((md[u] & 2u) ? 0xffffffffu : 1u);
                        shr[u] = fresh[u] && (prev + step == key);
                        const unsigned long long sbal = __builtin_amdgcn_ballot_w64(shr[u]);
                        const int ns = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sbal >> 32),
                                            __builtin_amdgcn_mbcnt_lo((uint32_t)sbal, shr[u] ? 1u : 0u));
                        cnt[u] = 2 * v4_popc(bal) - v4_popc(sbal);               // pairs of this view
                        incl[u] = 2 * (nl - 1) - ns;                              // the lane's first pair, numbered in travel order
                    } else {
                        cnt[u] = __popcll(bal);
                        incl[u] = nl;
                    }
                }
                int n_tot = 0;
#pragma unroll
                for (int u = 0; u < VG; ++u) n_tot += cnt[u];
                if (n_tot == 0) continue;                                         // wave-uniform: nothing open in this group
                if (TX) {
                    // ---------------- pair items: phase B (lists), correlation, phase C (combine) — the whole group, or view by view if the tables overflow ----------------
                    const bool whole = n_tot <= CAP;
                    int base = 2;                                                 // slots 0, 1: the zero pairs of closed lanes
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        if (!whole) base = 2;
                        const bool neg = (md[u] & 2u) != 0, rowm = (md[u] & 1u) != 0;
                        // slots are numbered along increasing x (y): travelling towards smaller coordinates reverses the view's numbering
                        const int s = neg ? base + cnt[u] - 2 - incl[u] : base + incl[u];
                        const uint32_t oJ = rowm ? row_bytes : texel_bytes, oM = rowm ? texel_bytes : row_bytes;
                        // the pair a sharing leader re-uses: its first (in coordinate order) when travelling up, its second when travelling down
                        if (fresh[u] && !(shr[u] && !neg)) *reinterpret_cast<uint2*>(items + (s - 2) * 2) = make_uint2(qoff[u], qoff[u] + oM);
                        if (fresh[u] && !(shr[u] && neg))  *reinterpret_cast<uint2*>(items + (s - 1) * 2) = make_uint2(qoff[u] + oJ, qoff[u] + oJ + oM);
                        incl[u] = gate[u] ? s : 0;
                        base += cnt[u];
                        if (!whole) {
                            if (cnt[u] == 0) continue;                            // wave-uniform
                            if (lane == 0) *reinterpret_cast<uint2*>(items + cnt[u] * 2) = make_uint2(0u, 0u);
                            fwave_lds_fence();
                            correlate(cnt[u]);
                            fwave_lds_fence();
                            const float2 cA = ctab2[incl[u]], cB = ctab2[incl[u] + 1];
                            const float fmin = rowm ? wa[u] : wb[u], fmaj = rowm ? wb[u] : wa[u];
                            const float t = __builtin_fmaf(fmin, cA.y - cA.x, cA.x), l = __builtin_fmaf(fmin, cB.y - cB.x, cB.x);
                            const float c = __builtin_fmaf(fmaj, l - t, t);       // homography.py:150,155 (grid_sample's bilinear weights, factored)
                            acc += gate[u] ? c : 0.f;
                            fwave_lds_fence();
                        }
                    }
                    if (whole) {
                        if (lane == 0) *reinterpret_cast<uint2*>(items + n_tot * 2) = make_uint2(0u, 0u);   // pad item: view 0, texel 0
                        fwave_lds_fence();
                        correlate(n_tot);
                        fwave_lds_fence();
#pragma unroll
                        for (int u = 0; u < VG; ++u) {
                            const bool rowm = (md[u] & 1u) != 0;
                            const float2 cA = ctab2[incl[u]], cB = ctab2[incl[u] + 1];
                            const float fmin = rowm ? wa[u] : wb[u], fmaj = rowm ? wb[u] : wa[u];
                            const float t = __builtin_fmaf(fmin, cA.y - cA.x, cA.x), l = __builtin_fmaf(fmin, cB.y - cB.x, cB.x);
                            const float c = __builtin_fmaf(fmaj, l - t, t);       // homography.py:150,155
                            acc += gate[u] ? c : 0.f;                             // homography.py:159,116 (fp32 here)
                        }
                        fwave_lds_fence();                                        // the tables are rewritten by the next group
                    }
                } else if (n_tot <= CAP) {
                    // ---------------- phase B: one item list for the group ----------------
                    int base = 0;
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        if (fresh[u]) items[base + incl[u] - 1] = qoff[u];
                        incl[u] = gate[u] ? base + incl[u] : 0;                   // -> slot of this lane's item (0 = the zero slot)
                        base += cnt[u];
                    }
                    if (lane == 0) items[n_tot] = 0u;                             // pad item: view 0, texel 0
                    fwave_lds_fence();
                    correlate(n_tot);
                    fwave_lds_fence();
                    // ---------------- phase C: bilinear combine + view accumulation ----------------
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        const float4 c4 = ctab[incl[u]];
                        float c = c4.x * wa[u];
                        c = __builtin_fmaf(c4.y, wb[u], c);
                        c = __builtin_fmaf(c4.z, wc[u], c);
                        c = __builtin_fmaf(c4.w, wd[u], c);
                        acc += gate[u] ? c : 0.f;                                 // homography.py:159,116 (fp32 here)
                    }
                    fwave_lds_fence();                                            // ctab/items are rewritten by the next group
                } else {
                    // more distinct open quads than the tables hold (long epipolar segments): one view at a time
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        if (cnt[u] == 0) continue;                                // wave-uniform
                        if (fresh[u]) items[incl[u] - 1] = qoff[u];
                        if (lane == 0) items[cnt[u]] = 0u;
                        fwave_lds_fence();
                        correlate(cnt[u]);
                        fwave_lds_fence();
                        const float4 c4 = ctab[gate[u] ? incl[u] : 0];
                        float c = c4.x * wa[u];
                        c = __builtin_fmaf(c4.y, wb[u], c);
                        c = __builtin_fmaf(c4.z, wc[u], c);
                        c = __builtin_fmaf(c4.w, wd[u], c);
                        acc += gate[u] ? c : 0.f;
                        fwave_lds_fence();
                    }
                }
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
            outb[(q & (OUT_PX - 1)) * DL + j0] = cval;                            // OUT_PX and NPX are powers of two
            if (((q + 1) & (OUT_PX - 1)) == 0) {
                // ---- OUT_PX px x 64 results: LDS -> coalesced row segments of cost[b, j, y, :] ----
                const int q_base = q + 1 - OUT_PX;
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

static size_t fast64_lds_bytes(const CvParams& p, bool tx) {
    const size_t tables = tx ? (size_t)(130 * 8 + 130 * 8 + (p.V * p.npx * 4 + 15) / 16 * 16) : (size_t)(65 * 16 + 68 * 4);
    return (size_t)4 * (p.V * p.npx * 32 + tables + (p.cost_hi ? 0 : (p.npx < 8 ? p.npx : 8) * 64 * 4));
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG>
static hipError_t launch_fast64_v(const CvParams& p0, hipStream_t stream) {
    // Pixels per wave.  A wave owns its pixels for NPX x V iterations, and everything the resident waves of an XCD
    // (~900) own at one time must keep its source footprint in that XCD's 4 MiB L2, or texels shared by neighbouring pixels
    // are fetched from HBM again and again (measured with 16 x 4 tiles: 2.3-3 GB per launch for 0.85 GB of inputs).
    // Footprint ~ (900 * NPX / w + 2 * halo) rows x (w + 2) x V x texel bytes.  Measured at C2 (profiles/r2): NPX = 8 in raster
    // order cuts the L2 misses 2.6x (FETCH 2.1 -> 0.83 GB per launch = the compulsory bytes); smaller NPX pays more per-wave
    // prologue (projection table, validity mask) than it saves.
    CvParams p = p0;
    p.npx = 8;
    p.tiles_x = (p.w + 4 * p.npx - 1) / (4 * p.npx);
    p.tiles_y = p.h;
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    // Full-resolution grids (w > 512): the resident blocks of an XCD walk a narrow vertical strip instead of whole rows — with candidate
    // segments ~40 texels long in both directions a 640 x 11 band of reference pixels touches twice the source texels of a 32 x 200
    // strip.  Measured per strip width (profiles/r5/strip_sweep.log, kernel alone, warm): 480 x 640 ScanNet grids 32-pixel strips
    // (C2L 2.30 -> 1.90 ms, fp32 features 2.25 -> 1.40 ms; 64 / 128 / 256 pixels: 1.93 / 1.97 / 2.00); the 352 x 1216 KITTI grid (forward
    // motion: radial segments, no preferred direction) hardly cares: 1.53 raster, 1.57 / 1.56 / 1.53 / 1.49 ms at 32 / 64 / 128 / 256 pixels
    p.strip_tx = p.w > 1024 ? 8 : (p.w > 512 ? 1 : 0);
    // Texel-pair items where they pay: bf16 features (C2L 1.97 -> 1.90 ms); with fp32 features the per-view list bookkeeping costs more
    // than the texels it saves (C4L 2.79 -> 3.09 ms), and at 3.5 items per (pixel, view) (C2, C4 grids, which cost_volume_v3.hip serves)
    // it loses 15 - 20 %: profiles/r5/ablate_tx.log
    bool tx = sizeof(FeatT) == 2;
#ifdef MAGNET_DEV
    { static const int strip = getenv("MAGNET_STRIP") ? atoi(getenv("MAGNET_STRIP")) : -1; if (strip >= 0) p.strip_tx = strip; }   // dev: block order A/B
    if (CV_DEV(p) & 0x400) tx = !tx;                                               // dev: the other item form, same box
#endif
    if (!tx) {
        if (p.gate_bits) hipLaunchKernelGGL((cv_fast64_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 1>), grid, block, fast64_lds_bytes(p, false), stream, p);
        else hipLaunchKernelGGL((cv_fast64_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 0>), grid, block, fast64_lds_bytes(p, false), stream, p);
        return hipGetLastError();
    }
    if (p.gate_bits) hipLaunchKernelGGL((cv_fast64_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 5>), grid, block, fast64_lds_bytes(p, true), stream, p);
    else hipLaunchKernelGGL((cv_fast64_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 4>), grid, block, fast64_lds_bytes(p, true), stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU>
static hipError_t launch_fast64(const CvParams& p, hipStream_t stream) {
    // views per group: as many as possible (<= 4) without idle slots in the last group (round 5, C2L, same box: 4 views at 5 waves per
    // SIMD 1.908 ms; 2 views at 6 waves 1.946; 1 view 1.939; compiled for 8 waves (spills) 2.415: profiles/r5/f64_occupancy.log)
    int vg = p.V >= 4 ? 4 : p.V;
    if (p.V > 4 && p.V % 4 != 0 && (p.V % 3 == 0 || p.V % 4 < p.V % 3)) vg = 3;
    switch (vg) {
        case 1: return launch_fast64_v<FeatT, CPL, FULL, MINW, LPU, 1>(p, stream);
        case 2: return launch_fast64_v<FeatT, CPL, FULL, MINW, LPU, 2>(p, stream);
        case 3: return launch_fast64_v<FeatT, CPL, FULL, MINW, LPU, 3>(p, stream);
        default: return launch_fast64_v<FeatT, CPL, FULL, MINW, LPU, 4>(p, stream);
    }
}

// D > 32 with candidates sampled in the kernel; called by launch_cv_fast, which has checked everything else.
hipError_t launch_cv_fast64(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if (p.D <= 32) return hipSuccess;
    if ((size_t)p.V * p.B * (size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets over all views
    if ((size_t)p.V * 512 * 4 > 48 * 1024) return hipSuccess;
    const int nchunk = (int)(p.F * esz / 16);
    *handled = true;
    if (p.feat_bf16) {
        if (nchunk == 8)  return launch_fast64<uint16_t, 2, true, 5, 4>(p, stream);       // F = 64: 4 lanes x 32 B per texel
        if (nchunk <= 8)  return launch_fast64<uint16_t, 1, false, 5, 8>(p, stream);
        if (nchunk <= 16) return launch_fast64<uint16_t, 2, false, 5, 8>(p, stream);
    } else {
        if (nchunk == 16) return launch_fast64<float, 2, true, 5, 8>(p, stream);          // F = 64
        if (nchunk <= 8)  return launch_fast64<float, 1, false, 5, 8>(p, stream);
        if (nchunk <= 16) return launch_fast64<float, 2, false, 5, 8>(p, stream);
        if (nchunk <= 32) return launch_fast64<float, 4, false, 4, 8>(p, stream);
    }
    *handled = false;
    return hipSuccess;
}

}  // namespace magnet
