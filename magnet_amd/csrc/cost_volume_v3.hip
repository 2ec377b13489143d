// cost_volume_v3.hip — production matcher for D > 32, round 3 (one reference pixel per wave iteration, lane = candidate).
//
// Same mapping as cost_volume_fast64.hip (round 2): the VG views of a pixel run geometry / (mu, sigma) quad loads / gate back to
// back, the distinct gate-open quads of the group become ONE item list, their (item, tap) dot products are fetched in blocks,
// then the VG bilinear combines run.  What round 3 measured and changed (profiles/r3/):
//   * Instruction cost (issue_rate.txt): on gfx950 only v_add/sub/mul_f32, v_add_u32 (2.6 cycles per wave64 instruction) and
//     v_fma_f32 (3.3) run at the full rate; compares, selects, floor / cvt, shifts, DPP, v_mbcnt, v_dot2c cost 4.3 - 4.5,
//     v_rcp_f32 8.2 — and a SCALAR instruction costs its SIMD 4.4 cycles of issue too (own pipe, but a wave issues one
//     instruction at a time).  The round-2 kernel issued 90 vector + 55 scalar instructions per (pixel, view).
//   * Occupancy is what hides this kernel's memory latency (4 -> 5 -> 6 waves per SIMD: 1.25 -> 1.11 -> 0.95 ms on the same
//     code): the register budget is a first-order performance parameter.
// Hence:
//   1. QUAD FORM.  The source (mu, sigma) map is stored per quad origin as {v00, v10-v00, v01-v00, (v11-v01)-(v10-v00)}
//      (magnet_pack_gmm_quad): a bilinear sample = 3 fma per channel, no weights (was 2 sub + 4 mul + 8 fma); the combine
//      derives its four weights from bx, by, bx*by in 3 subtractions.
//   2. SCALAR MASKS.  Lane predicates live as 64-bit ballots: compares write SGPR pairs, the logic runs on the scalar unit,
//      masked LDS stores (s_and_saveexec around the store) and v_cndmask take the mask as an operand — no per-lane booleans,
//      no v_cndmask 0/1 + v_cmp pairs to rebuild a ballot.
//   3. ONE run detection per view, after the gate: DPP compare with the previous lane + ballot + one v_mbcnt pair number the
//      distinct open quads; the same number is the item index and the slot of the run's four tap correlations.
//   4. Float index arithmetic (fma + one cvt instead of cvt / min / mad chains); block -> (frame, row, segment) by multiplication
//      with host-computed reciprocals on the scalar unit (the generic integer division is ~25 vector instructions each);
//      reference vectors and (mu, sigma) of the wave's pixels staged once in LDS; invalid views compacted away up front.
//   5. REGISTER DIET.  Views are software-pipelined by hand (geometry + quad loads of view u, then the gate of view u - 1):
//      two views' quads and one view's projection terms live at a time; two correlation passes of loads in flight; the split
//      output form is its own instance: 78 registers -> 6 waves per SIMD (round 2: 70 -> 7, but 45 % more instructions).
// Measured and rejected in round 3: loader lanes that fetch one (mu, sigma) quad per RUN through LDS slots (a
// quarter of the load instructions, but two more LDS round trips in the dependency chain: 1.11 vs 0.95 ms) and a two-stage
// software pipeline over groups of two views on top of it (register pressure: 1.38 ms).
// Round 5 (profiles/r5/NOTES.md): (a) the split-output F = 64 instances correlate the view groups of TWO
// neighbouring pixels in one batch (PX2 below: ~11 instead of ~5.5 items per pass sequence of 4-item passes, a rolling two-pass load
// pipeline, reference vectors per pass from LDS; 64 registers = 8 waves; bit-identical; C2 0.823 -> 0.796 ms); (b) grids wider than 256
// walk 32-pixel-wide vertical strips (C4 0.696 -> 0.682 ms); (c) texel-pair items (cost_volume_fast64.hip's item form) were built into this kernel
// too and measured 3.3 % SLOWER at C2's item counts (0.851 vs 0.824 ms, profiles/r5/ablate_v3_pair_items.log): removed in round 6 (the code is
// in the history at 0e28a09); this kernel has quad items only.
// Arithmetic and tolerance contract: as cost_volume_fast.hip (fma-contracted geometry, one v_rcp_f32, padded-map texel
// coordinates, fp32 view sum); the (mu, sigma) and correlation interpolations use the quad form / difference-form weights,
// which changes results by fp32 rounding only (homography.py:150-152,155-159).
#include <stdlib.h>
#include "cv_runs.hpp"

namespace magnet {

// fixed offsets inside a wave's LDS region
constexpr int V3_CAP = 64;                         // open runs (= items) of one view group
constexpr int V3_NPASS_DEFAULT = 2;                 // correlation passes whose loads are in flight together (registers: 8 per pass)
constexpr int V3_NPX = 8;                          // reference pixels per wave (raster-order row segments: cost_volume_fast64.hip)
constexpr int V3_CT = 0;                           // [CAP + 1] x 16 B: the 4 tap correlations of each open run; slot CAP = dump
constexpr int V3_IT = 65 * 16;                     // [CAP + 4] x 8 B: open runs {feature byte offset, LDS address of the run's slot}
constexpr int V3_MS = (V3_IT + 68 * 8 + 15) / 16 * 16;   // [NPX] x 8 B: (mu, sigma) of the wave's reference pixels
constexpr int V3_FIX = V3_MS + V3_NPX * 8;         // then: view table [Vr] x 8 B, projection table [Vr][NPX] x 32 B, reference vectors, output stage

// CPL / FULL / LPU: VALU correlation units of LPU lanes x CPL 16-byte chunks (as cv_fast_kernel); VG = views per group;
// OPT bit 0: write the gate bits (debug / parity tests); bit 1 (dev builds): no dot products; bit 6: split output form only;
// bits 8..11: correlation passes whose loads are in flight together
// dev A/B (HALFQ): q0 holds 8 fp16 {mu quad form, sigma quad form}; widen to the two fp32 quads
__device__ __forceinline__ void v3_unpack_quad16(float4& q0, float4& q1) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    const h2_t a = __builtin_bit_cast(h2_t, q0.x), b = __builtin_bit_cast(h2_t, q0.y), c = __builtin_bit_cast(h2_t, q0.z), d = __builtin_bit_cast(h2_t, q0.w);
    q0 = make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
    q1 = make_float4((float)c[0], (float)c[1], (float)d[0], (float)d[1]);
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG, int OPT>
__global__ __launch_bounds__(256, MINW) void cv_v3_kernel(const CvParams p) {
    constexpr bool GBITS = (OPT & 1) != 0;
    constexpr bool NO_CORR = (OPT & 2) != 0;              // dev: no feature loads / dot products
    constexpr bool SPLIT = (OPT & 64) != 0;               // the split-bf16 channel-last output form only (cost_hi given): no NCHW staging code, fewer live scalars
    constexpr int NPX = V3_NPX;
    constexpr int IPP = 64 / (4 * LPU);                   // items per correlation pass
    constexpr int HALFQ = (OPT >> 14) & 3;                // dev A/B (round 6, tools/ablate.py ABLATE_HALFQ): the quad-form (mu, sigma) entry as 8 fp16 = ONE 16-byte
                                                          // load per candidate from a 16-byte-stride map (the tool converts the map); gates differ from the fp32 map's
                                                          // at the 1e-3 level (outside the contract), the amount of work does not: what the map's bytes cost
    constexpr bool PX2 = (OPT & 0x2000) != 0 && SPLIT && FULL;   // two pixels per correlation batch (round 5; split output form, F = 64 instances)
    constexpr bool QF = LPU == 4 && !(OPT & 128);                         // an item's four taps share a 16-lane row: correlations stored in quad form (cv_runs.hpp)
    constexpr int O_IT = V3_IT, O_MS = V3_MS, O_FIX = V3_FIX;
    constexpr int NPASS = ((OPT >> 8) & 15) ? ((OPT >> 8) & 15) : V3_NPASS_DEFAULT;   // passes fetched together
    constexpr int CSTR = LPU * 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block -> (frame b, pixel row y, segment tx): the XCD-aware remap of cv_common.hpp with the two runtime divisions done by
    // multiplication with host-computed reciprocals (all scalar: blockIdx is wave-uniform; the generic integer division is ~25
    // vector instructions each)
    int tile, b;
    {
        const unsigned n = gridDim.x, bid = blockIdx.x;
        const unsigned qn = n / NUM_XCD, rn = n % NUM_XCD;
        const unsigned xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
        const unsigned start = (xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        const unsigned logical = start + idx;
        // (round 4: readfirstlane — the high multiply runs on the vector unit, and the frame's base pointers derived from b were carried
        // as per-lane 64-bit values: one v_lshl_add_u64 per vector load instead of the scalar-base addressing mode)
        b = __builtin_amdgcn_readfirstlane(p.magic_tiles ? (int)__umulhi(logical, p.magic_tiles) : (int)logical);        // magic 0 = division by 1
        tile = (int)(logical - (unsigned)b * (unsigned)(p.tiles_x * p.tiles_y));
    }
    int y = __builtin_amdgcn_readfirstlane(p.magic_tiles_x ? (int)__umulhi((unsigned)tile, p.magic_tiles_x) : tile);   // raster order: see cost_volume_fast64.hip's launcher
    int tx = tile - y * p.tiles_x;
    if (p.strip_tx == 1) {                                  // column-major tile order: 32-pixel-wide vertical strips (grids wider than 256: the launcher)
        tx = __builtin_amdgcn_readfirstlane(p.magic_tiles_y ? (int)__umulhi((unsigned)tile, p.magic_tiles_y) : tile);
        y = tile - tx * p.tiles_y;
    }
    const int yc = min(y, p.h - 1);
    const int x_base = (tx * 4 + wv) * NPX;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const int JB = (p.D + 63) / 64;
    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    const int nchunk = (int)(texel_bytes / 16);
    const uint32_t map_texels = (uint32_t)(Hp * Wp);
    const uint32_t vstride = (uint32_t)p.B * map_texels;                                    // texels between the views of a frame (view-major)

    // ---- valid views (homography.py:97), compacted: the loop below never sees an invalid view ----
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    const int nval = __popcll(vmask);
    const int Vr = (p.V + VG - 1) / VG * VG;                                                // table rows: views rounded up to whole groups

    // ---- wave-private LDS ----
    const int vt_bytes = Vr * 8, pv_bytes = Vr * NPX * 32, rf_bytes = NPX * (int)texel_bytes;
    const int out_bytes = (SPLIT || p.cost_hi) ? 0 : NPX * 64 * 4;
    const int wave_bytes = O_FIX + (vt_bytes + 15) / 16 * 16 + pv_bytes + rf_bytes + out_bytes;
    const uint32_t wb = (uint32_t)(uintptr_t)(v3_lds_u8*)smem + (uint32_t)(wv * wave_bytes);
    const uint32_t vtb = wb + O_FIX, pvb = vtb + (vt_bytes + 15) / 16 * 16, rfb = pvb + pv_bytes, outb = rfb + rf_bytes;

    // ---- per (compact view c, pixel q): depth-linear projection terms; per compact view: {texel offset of the view, view index} ----
    for (int e = lane; e < NPX * Vr; e += 64) {
        const int q = e % NPX, c = min(e / NPX, max(nval - 1, 0));                          // padding rows repeat the last valid view (masked in the loop)
        // view index of the c-th valid view: c-th set bit of vmask
        int v = 0;
        {
            unsigned long long m = vmask;
            for (int i = 0; i < c; ++i) m &= m - 1;
            v = m ? __builtin_ctzll(m) : 0;
        }
        const int xc = min(x_base + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, yc, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        v3_st_f4(pvb + e * 32, make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz));
        v3_st_f4(pvb + e * 32 + 16, make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz));
        if (q == 0) v3_st_u2(vtb + (e / NPX) * 8, make_uint2((uint32_t)v * vstride, (uint32_t)v));
    }
    // ---- reference vectors and (mu, sigma) of the wave's pixels (contiguous in their rows): staged once ----
    {
        const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) +
            ((size_t)b * hw + (size_t)yc * p.w) * texel_bytes;
        // the NPX pixels of the segment are contiguous in the channel-last row: a plain copy of NPX * texel_bytes bytes, clamped
        // at the row's end (pixels past the row are never processed)
        const uint32_t row_bytes = (uint32_t)p.w * texel_bytes, seg0 = (uint32_t)x_base * texel_bytes;
        for (uint32_t e = (uint32_t)lane * 16u; e < (uint32_t)NPX * texel_bytes; e += 1024u)
            v3_st_u4(rfb + e, *reinterpret_cast<const uint4*>(ref_row + min(seg0 + e, row_bytes - 16u)));
        if (lane < NPX) {
            const size_t pixr = (size_t)yc * p.w + min(x_base + lane, p.w - 1);
            v3_st_u2(wb + O_MS + lane * 8, make_uint2(__float_as_uint(p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr]),
                                                      __float_as_uint(p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr])));
        }
    }
    fwave_lds_fence();

    const uint32_t xlim = __float_as_uint((float)(p.w + 1)), ylim = __float_as_uint((float)(p.h + 1));   // see cost_volume_fast.hip
    const int sub = lane & (LPU - 1), tap = (lane / LPU) & 3, upair = lane / (4 * LPU);   // correlation roles
    const uint32_t lane_src_off = (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes + (uint32_t)sub * 16u;
    const uint32_t tap4 = (uint32_t)tap * 4u;
    const float invV = 1.0f / (float)p.V;
    // frame bases pinned into SGPRs and typed as GLOBAL pointers: base + zero-extended 32-bit lane offset then selects the
    // scalar-base addressing mode of the vector loads (the 64-bit multiply above runs on the vector unit, so the compiler carried
    // these bases as per-lane register pairs and added them with one v_lshl_add_u64 per load)
    typedef cvr_gptr v3_gptr;
    auto uniform_base = [](const void* q) { return (v3_gptr)(unsigned long long)v4_uniform_ptr(q); };
    const v3_gptr src_b = uniform_base(reinterpret_cast<const unsigned char*>(p.src_feat) + (size_t)b * map_texels * texel_bytes);
    // quad-form (mu, sigma) map of frame b over all views, as a buffer: the hardware bounds check replaces the clamp of the quad key
    const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(
        (void*)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_gmq) + (size_t)b * map_texels * (HALFQ ? 16 : 32)), 0,
        (int)(((uint32_t)(p.V - 1) * vstride + map_texels) * (HALFQ ? 16u : 32u)), 0x00020000);
    const float kappa = p.kappa;
    // split output: bases of the wave's first pixel pinned into SGPRs, per pixel a 32-bit byte offset (round 4; was 64-bit per-lane math)
    v4_gu8* const hi_base = (SPLIT || p.cost_hi) ? v4_uniform_gptr(p.cost_hi + (((size_t)b * Hp + (y + 1)) * Wp + (x_base + 1)) * (size_t)p.cost_ld) : nullptr;
    v4_gu8* const lo_base = (SPLIT || p.cost_hi) ? v4_uniform_gptr(p.cost_lo + (((size_t)b * Hp + (y + 1)) * Wp + (x_base + 1)) * (size_t)p.cost_ld) : nullptr;
    const uint32_t ld2 = (uint32_t)p.cost_ld * 2u;

    const int npix = min(NPX, p.w - x_base);                                                // pixels of the segment inside the row (may be <= 0)
    const uint32_t it_lane = wb + O_IT + (uint32_t)upair * 8u;                              // correlation unit -> its item entry of pass 0
    const uint32_t row_bytes = (uint32_t)Wp * texel_bytes;
    const uint32_t rf_lane = rfb + (uint32_t)sub * 16u;

    for (int jb = 0; jb < JB; ++jb) {
        const int j = jb * 64 + lane;
        const float kj = p.k[min(j, p.D - 1)];
        const unsigned long long jmask = __builtin_amdgcn_ballot_w64(j < p.D);               // lanes that hold a candidate
        if constexpr (PX2) {
            // ---- two pixels per correlation batch ---------------------------------------------------------------------------------
            // A view group of one pixel opens ~5.5 distinct quads: 1.4 correlation passes of 4 items, i.e. ~2.4 issued (a pass costs
            // its two wave-loads whatever it holds).  The groups of TWO neighbouring pixels share one item list and one pass sequence
            // (~11 items: 2.75 passes for both): a third fewer feature wave-loads and dot / reduce instructions, half the LDS round
            // trips per pixel.  What stays live per (pixel, view) across the batch is 3 registers (bx, by, slot address) and two scalar
            // masks; the reference vector of an item's pixel comes from the wave's LDS copy per pass instead of from registers.
            // Arithmetic per candidate and per item is unchanged: results are bit-identical to the one-pixel loop.
            int pstep = 2;
            for (int q = 0; q < npix;) {
                const int pn = min(pstep, npix - q);                                      // pixels of this batch (wave-uniform)
                float dpx[2], acc2[2] = {0.f, 0.f};
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const uint2 ms = v3_ld_u2(wb + O_MS + (uint32_t)min(q + pp, NPX - 1) * 8u);
                    dpx[pp] = __builtin_fmaf(__uint_as_float(ms.y), kj, __uint_as_float(ms.x));   // MAGNET.py:155
                }
                const uint32_t rf_q = rf_lane + (uint32_t)q * texel_bytes;
                auto correlate2 = [&](const int n) {
                    // two passes of loads in flight, ROLLING: as soon as a pass's dot products are done its registers take the loads of the
                    // pass after next (the one-pixel loop fetches in blocks of NPASS passes and drains between blocks)
                    const int npass = (n + IPP - 1) / IPP;
                    uint4 sv[2][CPL];
                    uint2 ent[2];
                    auto issue = [&](const int a, const int slot) {
                        ent[slot] = v3_ld_u2(it_lane + (uint32_t)a * (IPP * 8));                    // past the list: pad entries (dump slot)
                        const v3_gptr sp = src_b + ((ent[slot].x & ~1u) + lane_src_off);
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) sv[slot][cc] = v3_gld_u4(sp + cc * CSTR);
                    };
                    auto reduce = [&](const int slot) {
                        const uint32_t ra = rf_q + (ent[slot].x & 1u) * texel_bytes;       // the item's pixel: bit 0 of its entry
                        float part = 0.f;
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) part = fdot_chunk(v3_ld_u4(ra + cc * CSTR), sv[slot][cc], part, FeatT());
                        part = LPU == 8 ? freduce8(part) : v3_reduce4(part);
                        if (QF) part = v3_quadform16(part);
                        if (sub == 0) v3_st_f1(ent[slot].y + tap4, part);
                    };
                    issue(0, 0);
                    if (npass > 1) issue(1, 1);
                    for (int a = 0; a < npass; a += 2) {
                        reduce(0);
                        if (a + 2 < npass) issue(a + 2, 0);
                        if (a + 1 < npass) {
                            reduce(1);
                            if (a + 3 < npass) issue(a + 3, 1);
                        }
                    }
                };
                int gstep = VG;
                bool redo1 = false;
                for (int g0 = 0; g0 < nval;) {
                    const int nact = min(gstep, nval - g0);
                    float bx[2][VG], by[2][VG];                            // (bx * by is recomputed at the combine: one register less per (pixel, view))
                    uint32_t keyf[2][VG], raddr[2][VG];
                    unsigned long long Gb[2][VG], Lb[2][VG];
                    int n_items = 0;
                    const uint32_t vta = vtb + (uint32_t)(g0 * 8);
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        if (pp >= pn) {
#pragma unroll
                            for (int u = 0; u < VG; ++u) { Gb[pp][u] = 0ull; Lb[pp][u] = 0ull; bx[pp][u] = by[pp][u] = 0.f; keyf[pp][u] = 0u; }
                            continue;
                        }
                        const float d = dpx[pp];
                        const int x = x_base + q + pp;
                        float zw[VG], fq[VG];
                        unsigned long long Wb[VG];
                        float4 q0[VG], q1[VG];
                        uint32_t vidx[VG];
                        const uint32_t pva = pvb + (uint32_t)((g0 * NPX + q + pp) * 32);
                        auto gate_view = [&](const int u) {
                            const float mu_w = __builtin_fmaf(fq[u], q0[u].w, __builtin_fmaf(by[pp][u], q0[u].z, __builtin_fmaf(bx[pp][u], q0[u].y, q0[u].x)));   // homography.py:151
                            const float sg_w = __builtin_fmaf(fq[u], q1[u].w, __builtin_fmaf(by[pp][u], q1[u].z, __builtin_fmaf(bx[pp][u], q1[u].y, q1[u].x)));   // homography.py:152
                            Gb[pp][u] = __builtin_amdgcn_ballot_w64(__builtin_fabsf(zw[u] - mu_w) < sg_w * kappa) & Wb[u];   // homography.py:157-158
                            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)keyf[pp][u], 0x138, 0xf, 0xf, true);
                            Lb[pp][u] = (__builtin_amdgcn_ballot_w64(keyf[pp][u] != prev) | 1ull | ~(Gb[pp][u] << 1)) & Gb[pp][u];
                            n_items += v4_popc(Lb[pp][u]);
                            if (GBITS) {
                                if (j < p.D && u < nact)
                                    p.gate_bits[(((size_t)b * p.V + vidx[u]) * p.D + j) * hw + (size_t)y * p.w + x] = (uint8_t)v3_sel_u(Gb[pp][u], 1u, 0u);
                            }
                        };
#pragma unroll
                        for (int u = 0; u < VG; ++u) {
                            const float4 pa = v3_ld_f4(pva + u * (NPX * 32)), pb = v3_ld_f4(pva + u * (NPX * 32) + 16);
                            const uint2 vt = v3_ld_u2(vta + u * 8);
                            const float Px = __builtin_fmaf(pa.x, d, pb.x);               // homography.py:132
                            const float Py = __builtin_fmaf(pa.y, d, pb.y);
                            const float Pz = __builtin_fmaf(pa.z, d, pb.z);
                            zw[u] = __builtin_fmaf(pa.w, d, pb.w);                        // homography.py:137-138
                            const float rz = __builtin_amdgcn_rcpf(Pz);                   // homography.py:133
                            const float ixs = __builtin_fmaf(Px, rz, 0.5f);
                            const float iys = __builtin_fmaf(Py, rz, 0.5f);
                            bx[pp][u] = __builtin_amdgcn_fractf(ixs); by[pp][u] = __builtin_amdgcn_fractf(iys);
                            fq[u] = bx[pp][u] * by[pp][u];
                            const unsigned long long wx = __builtin_amdgcn_ballot_w64(__float_as_uint(ixs) < xlim);
                            const unsigned long long wy = __builtin_amdgcn_ballot_w64(__float_as_uint(iys) < ylim);
                            Wb[u] = (u < nact) ? (wx & wy & jmask) : 0ull;
                            keyf[pp][u] = __umul24(v3_cvt_u32_sat(iys), (uint32_t)Wp) + v3_cvt_u32_sat(ixs) + vt.x;
                            const int qo = (int)(keyf[pp][u] << (HALFQ ? 4 : 5));
                            q0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, qo, 0, 0));
                            if constexpr (HALFQ != 0) v3_unpack_quad16(q0[u], q1[u]);
                            else q1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, qo + 16, 0, 0));
                            if (GBITS) vidx[u] = vt.y;
                            __builtin_amdgcn_sched_barrier(0);
                            if (u > 0) { gate_view(u - 1); __builtin_amdgcn_sched_barrier(0); }
                        }
                        gate_view(VG - 1);
                    }
                    if (n_items == 0) { g0 += nact; continue; }
                    if (n_items > V3_CAP) {
                        if (gstep > 1) { gstep = 1; continue; }                           // view by view
                        redo1 = true; break;                                              // still too many: this pixel alone
                    }
                    {
                        int base = 0;
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                            for (int u = 0; u < VG; ++u) {
                                const unsigned long long Ls = Lb[pp][u] >> 1;
                                const int sb = base + (int)(Lb[pp][u] & 1ull) - 1;
                                const uint32_t cnt = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ls, 0u));
                                raddr[pp][u] = (cnt << 4) + (wb + V3_CT + (uint32_t)(sb * 16));
                                v3_st2_mask(Lb[pp][u], (cnt << 3) + (wb + V3_IT + (uint32_t)(sb * 8)), __umul24(keyf[pp][u], texel_bytes) | (uint32_t)pp, raddr[pp][u]);
                                base += v4_popc(Lb[pp][u]);
                            }
                    }
                    v3_st2_mask((1ull << (IPP - 1)) - 1ull, wb + V3_IT + ((uint32_t)n_items + (uint32_t)lane) * 8u, 0u, wb + V3_CT + V3_CAP * 16);
                    fwave_lds_fence();
                    correlate2(n_items);
                    fwave_lds_fence();
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int u = 0; u < VG; ++u) {
                            const float4 c4 = v3_ld_f4(raddr[pp][u]);
                            const float fqc = bx[pp][u] * by[pp][u];
                            float c;
                            if (QF) {                                                      // homography.py:150,155
                                c = __builtin_fmaf(fqc, c4.w, __builtin_fmaf(by[pp][u], c4.z, __builtin_fmaf(bx[pp][u], c4.y, c4.x)));
                            } else {
                                const float w10 = bx[pp][u] - fqc, w01 = by[pp][u] - fqc;
                                const float w00 = (1.0f - bx[pp][u]) - w01;
                                c = c4.x * w00;
                                c = __builtin_fmaf(c4.y, w10, c);
                                c = __builtin_fmaf(c4.z, w01, c);
                                c = __builtin_fmaf(c4.w, fqc, c);
                            }
                            acc2[pp] += v3_sel_f(Gb[pp][u], c, 0.f);                       // homography.py:159,116 (fp32 here)
                        }
                    fwave_lds_fence();
                    g0 += nact;
                }
                if (redo1) { pstep = 1; continue; }
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    if (pp < pn) {
                        const float cval = acc2[pp] * invV;                               // homography.py:118,120
                        const uint32_t off = (uint32_t)(q + pp) * ld2 + (uint32_t)j * 2u;
                        if (j < p.D) {
                            const uint16_t hi = f32_to_bf16_rne(cval);
                            const uint16_t lo = f32_to_bf16_rne(cval - bf16_to_f32(hi));
                            *reinterpret_cast<v4_gu16*>(hi_base + off) = hi; *reinterpret_cast<v4_gu16*>(lo_base + off) = lo;
                        }
                    }
                }
                q += pn;
                pstep = 2;
            }
            continue;
        }
        for (int q = 0; q < npix; ++q) {
            const int x = x_base + q;
            uint4 rvp[CPL];                                                               // this lane's chunk(s) of the pixel's reference vector
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc)
                rvp[cc] = (FULL || (sub + LPU * cc < nchunk)) ? v3_ld_u4(rf_lane + q * texel_bytes + cc * CSTR) : make_uint4(0, 0, 0, 0);
            float d;
            {
                const uint2 ms = v3_ld_u2(wb + O_MS + q * 8);
                d = __builtin_fmaf(__uint_as_float(ms.y), kj, __uint_as_float(ms.x));     // MAGNET.py:155
            }
            float acc = 0.f;

            // (item, tap) dot products of the n open runs listed in the item table -> the runs' slots
            auto correlate = [&](const int n) {
                for (int ps = 0; ps < n; ps += IPP * NPASS) {
                    uint4 sv[NPASS][CPL];
                    uint2 ent[NPASS];
                    const uint32_t ita = it_lane + (uint32_t)ps * 8u;
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) ent[a] = v3_ld_u2(ita + a * (IPP * 8));             // past the list: pad entries (dump slot)
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {
                        if (a > 0 && ps + IPP * a >= n) break;                            // wave-uniform: this pass holds no item
                        const v3_gptr sp = src_b + (ent[a].x + lane_src_off);
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
                        part = LPU == 8 ? freduce8(part) : v3_reduce4(part);
                        if (QF) part = v3_quadform16(part);                                 // slot = {c00, dc/dx, dc/dy, d2c/dxdy}
                        if (sub == 0) v3_st_f1(ent[a].y + tap4, part);                     // pad units write the dump slot
                    }
                }
            };

            int gstep = VG;                                                               // views per group (1 after an overflow)
            for (int g0 = 0; g0 < nval;) {
                const int nact = min(gstep, nval - g0);
                float bx[VG], by[VG], fxy[VG], zw[VG];
                uint32_t keyf[VG], raddr[VG];
                unsigned long long Wb[VG], Gb[VG], Lb[VG];
                float4 q0[VG], q1[VG];
                uint32_t vidx[VG];
                const uint32_t pva = pvb + (uint32_t)((g0 * NPX + q) * 32);
                const uint32_t vta = vtb + (uint32_t)(g0 * 8);
                int n_items = 0;
                // geometry + quad loads of view u, then the gate of view u - 1 while those loads are in flight: at most two views' quads
                // and one view's projection terms are live at a time (the register budget decides the waves per SIMD, and with them
                // how well the kernel hides its memory latency)
                auto gate_view = [&](const int u) {
                    const float mu_w = __builtin_fmaf(fxy[u], q0[u].w, __builtin_fmaf(by[u], q0[u].z, __builtin_fmaf(bx[u], q0[u].y, q0[u].x)));   // homography.py:151
                    const float sg_w = __builtin_fmaf(fxy[u], q1[u].w, __builtin_fmaf(by[u], q1[u].z, __builtin_fmaf(bx[u], q1[u].y, q1[u].x)));   // homography.py:152
                    Gb[u] = __builtin_amdgcn_ballot_w64(__builtin_fabsf(zw[u] - mu_w) < sg_w * kappa) & Wb[u];   // homography.py:157-158
                    // runs of equal quads among the OPEN lanes: leader = open and (lane 0, or another quad than the previous lane, or the previous lane closed)
                    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)keyf[u], 0x138, 0xf, 0xf, true);   // wave_shr:1, lane 0 <- 0
                    Lb[u] = (__builtin_amdgcn_ballot_w64(keyf[u] != prev) | 1ull | ~(Gb[u] << 1)) & Gb[u];
                    n_items += __popcll(Lb[u]);
                    if (GBITS) {
                        if (j < p.D && u < nact)
                            p.gate_bits[(((size_t)b * p.V + vidx[u]) * p.D + j) * hw + (size_t)y * p.w + x] = (uint8_t)v3_sel_u(Gb[u], 1u, 0u);
                    }
                };
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    const float4 pa = v3_ld_f4(pva + u * (NPX * 32)), pb = v3_ld_f4(pva + u * (NPX * 32) + 16);
                    const uint2 vt = v3_ld_u2(vta + u * 8);
                    const float Px = __builtin_fmaf(pa.x, d, pb.x);                       // homography.py:132
                    const float Py = __builtin_fmaf(pa.y, d, pb.y);
                    const float Pz = __builtin_fmaf(pa.z, d, pb.z);
                    zw[u] = __builtin_fmaf(pa.w, d, pb.w);                                // homography.py:137-138
                    const float rz = __builtin_amdgcn_rcpf(Pz);                           // homography.py:133
                    const float ixs = __builtin_fmaf(Px, rz, 0.5f);
                    const float iys = __builtin_fmaf(Py, rz, 0.5f);
                    bx[u] = __builtin_amdgcn_fractf(ixs); by[u] = __builtin_amdgcn_fractf(iys);     // = ixs - floor(ixs), exact inside the window
                    fxy[u] = bx[u] * by[u];
                    const unsigned long long wx = __builtin_amdgcn_ballot_w64(__float_as_uint(ixs) < xlim);
                    const unsigned long long wy = __builtin_amdgcn_ballot_w64(__float_as_uint(iys) < ylim);
                    Wb[u] = (u < nact) ? (wx & wy & jmask) : 0ull;                          // (jmask: lanes that hold a candidate)
                    // quad index relative to (frame b, view 0): truncation = floor inside the window (ixs, iys >= 0); garbage outside it
                    keyf[u] = __umul24(v3_cvt_u32_sat(iys), (uint32_t)Wp) + v3_cvt_u32_sat(ixs) + vt.x;
                    // bounds-checked buffer loads (round 4): the key of a lane outside the window is garbage and reads as zero; was min + global load
                    const int qo = (int)(keyf[u] << (HALFQ ? 4 : 5));
                    q0[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, qo, 0, 0));
                    if constexpr (HALFQ != 0) v3_unpack_quad16(q0[u], q1[u]);
                    else q1[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_q, qo + 16, 0, 0));
                    if (GBITS) vidx[u] = vt.y;
                    __builtin_amdgcn_sched_barrier(0);
                    if (u > 0) { gate_view(u - 1); __builtin_amdgcn_sched_barrier(0); }
                }
                gate_view(VG - 1);
                if (n_items == 0) { g0 += nact; continue; }                               // wave-uniform: nothing open in this group
                if (n_items > V3_CAP) { gstep = 1; continue; }                                     // (only with nact > 1) redo view by view
                {
                    int base = 0;
#pragma unroll
                    for (int u = 0; u < VG; ++u) {
                        // slot of a lane = leaders at or below it - 1 = mbcnt(L >> 1) + (L & 1) - 1 (+ the slots of the views before)
                        const unsigned long long Ls = Lb[u] >> 1;
                        const int sb = base + (int)(Lb[u] & 1ull) - 1;
                        const uint32_t cnt = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ls, 0u));
                        raddr[u] = (cnt << 4) + (wb + V3_CT + (uint32_t)(sb * 16));
                        v3_st2_mask(Lb[u], (cnt << 3) + (wb + V3_IT + (uint32_t)(sb * 8)), __umul24(keyf[u], texel_bytes), raddr[u]);
                        base += __popcll(Lb[u]);
                    }
                }
                if (IPP > 1)
                    v3_st2_mask((1ull << (IPP - 1)) - 1ull, wb + V3_IT + ((uint32_t)n_items + (uint32_t)lane) * 8u, 0u, wb + V3_CT + V3_CAP * 16);
                fwave_lds_fence();
                if (!NO_CORR) correlate(n_items);
                fwave_lds_fence();
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    float c;
                    const float4 c4 = v3_ld_f4(raddr[u]);
                    if (QF) {                                                              // homography.py:150,155 (grid_sample's bilinear weights, factored)
                        c = __builtin_fmaf(fxy[u], c4.w, __builtin_fmaf(by[u], c4.z, __builtin_fmaf(bx[u], c4.y, c4.x)));
                    } else {
                        const float w10 = bx[u] - fxy[u], w01 = by[u] - fxy[u];
                        const float w00 = (1.0f - bx[u]) - w01;
                        c = c4.x * w00;
                        c = __builtin_fmaf(c4.y, w10, c);
                        c = __builtin_fmaf(c4.z, w01, c);
                        c = __builtin_fmaf(c4.w, fxy[u], c);
                    }
                    acc += v3_sel_f(Gb[u], c, 0.f);                                        // homography.py:159,116 (fp32 here)
                }
                fwave_lds_fence();
                g0 += nact;
            }
            const float cval = acc * invV;                                                // homography.py:118,120
            if (SPLIT || p.cost_hi) {
                // split-bf16 channel-last output for the conv kernel: lanes = 64 consecutive channels of one padded-grid row
                const uint32_t off = (uint32_t)q * ld2 + (uint32_t)j * 2u;
                if (j < p.D) {
                    const uint16_t hi = f32_to_bf16_rne(cval);
                    const uint16_t lo = f32_to_bf16_rne(cval - bf16_to_f32(hi));
                    *reinterpret_cast<v4_gu16*>(hi_base + off) = hi; *reinterpret_cast<v4_gu16*>(lo_base + off) = lo;
                }
                continue;
            }
            if (SPLIT) continue;
            v3_st_f1(outb + (uint32_t)(q * 64 + lane) * 4u, cval);
            if (q == npix - 1) {
                // ---- npix px x 64 results: LDS -> coalesced row segments of cost[b, j, y, :] ----
                fwave_lds_fence();
                const unsigned char* gbase = reinterpret_cast<const unsigned char*>(p.cost + (size_t)b * p.cost_bstride + (size_t)(jb * 64) * hw + (size_t)y * p.w + x_base);
                // lane -> (candidate jj = it * 8 + lane / 8, pixel qq = lane % 8) of flush iteration `it`
                const uint32_t out_ld_lane = outb + (uint32_t)(((lane & 7) * 64 + (lane >> 3)) * 4);
                const size_t out_g_lane = ((size_t)(lane >> 3) * hw + (size_t)(lane & 7)) * 4;
                const bool px_ok = (lane & 7) < npix;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const float v = __uint_as_float(v3_ld_u1(out_ld_lane + it * 32));
                    if (px_ok && jb * 64 + it * 8 + (lane >> 3) < p.D)
                        *reinterpret_cast<float*>(const_cast<unsigned char*>(gbase) + (size_t)it * 8 * hw * 4 + out_g_lane) = v;
                }
                fwave_lds_fence();
            }
        }
    }
}

static size_t v3_lds_bytes(const CvParams& p, int vg) {
    const size_t esz = p.feat_bf16 ? 2 : 4;
    const int Vr = (p.V + vg - 1) / vg * vg;
    return (size_t)4 * (V3_FIX + (Vr * 8 + 15) / 16 * 16 + Vr * V3_NPX * 32 + V3_NPX * p.F * esz + (p.cost_hi ? 0 : V3_NPX * 64 * 4));
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG>
static hipError_t launch_v3_v(const CvParams& p0, hipStream_t stream) {
    CvParams p = p0;
    p.npx = V3_NPX;
    p.tiles_x = (p.w + 4 * p.npx - 1) / (4 * p.npx);
    p.tiles_y = p.h;
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    // n / d = umulhi(n, ceil(2^32 / d)) exactly while n * d < 2^32 (n < grid size; checked by launch_cv_v3); 0 = division by 1
    const uint64_t nt = (uint64_t)p.tiles_x * p.tiles_y;
    p.magic_tiles = nt > 1 ? (uint32_t)((((uint64_t)1 << 32) + nt - 1) / nt) : 0u;
    p.magic_tiles_x = p.tiles_x > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)p.tiles_x - 1) / (uint64_t)p.tiles_x) : 0u;
    p.magic_tiles_y = p.tiles_y > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)p.tiles_y - 1) / (uint64_t)p.tiles_y) : 0u;
    // block order: raster rows up to 256-pixel-wide grids; wider ones (C4's 304) walk 32-pixel-wide vertical strips, whose source
    // footprint per XCD is smaller (profiles/r5/v3_strip.log: C2 0.821 = 0.820 ms, C5 0.882 -> 0.877, C4 0.696 -> 0.682 ms)
    p.strip_tx = p.w > 256 ? 1 : 0;
    size_t lds = v3_lds_bytes(p, VG);
#ifdef MAGNET_DEV
    { static const int strip = getenv("MAGNET_STRIP") ? atoi(getenv("MAGNET_STRIP")) : -1; if (strip >= 0) p.strip_tx = strip; }   // dev: block order A/B
    {   // dev: cap the workgroups per CU (waves per SIMD) by asking for more LDS than the kernel uses
        const int cap = (CV_DEV(p) & 0x300000) == 0x300000 ? 3 : (CV_DEV(p) & 0x200000) ? 4 : (CV_DEV(p) & 0x100000) ? 5 : 0;
        if (cap) { const size_t need = (size_t)160 * 1024 / (cap + 1) + 512; if (lds < need) lds = need; }
    }
#endif
    constexpr int NP = (CPL >= 4 ? 1 : V3_NPASS_DEFAULT) << 8;     // 4 chunks per lane: one pass already has 4 wave-loads in flight
    constexpr int MW2 = CPL >= 4 ? 4 : (MINW > 5 ? 5 : MINW);             // the NCHW-output and gate-bit instances carry more live values: one wave per SIMD less instead of scratch
#ifdef MAGNET_DEV
    if (CV_DEV(p) & 0x200) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MW2, LPU, VG, NP | 2>), grid, block, lds, stream, p); return hipGetLastError(); }      // no dot products (timing only)
    if (p.cost_hi && (CV_DEV(p) & 0x20000)) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, NP | 64 | 128>), grid, block, lds, stream, p); return hipGetLastError(); }   // dev: four-weight combine instead of the quad-form slots
    if (p.cost_hi && (CV_DEV(p) & 0x4000)) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 64 | 0x200>), grid, block, lds, stream, p); return hipGetLastError(); }
    if (p.cost_hi && (CV_DEV(p) & 0x40000)) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 64 | 0x300>), grid, block, lds, stream, p); return hipGetLastError(); }
    if (p.cost_hi && (CV_DEV(p) & 0x80000)) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, 64 | 0x400>), grid, block, lds, stream, p); return hipGetLastError(); }
#endif
    // The split-output form of the F = 64 instances (what MAGNET.forward runs) takes TWO pixels per correlation batch, compiled for 8 waves
    // per SIMD (63 - 64 registers, no scratch): bit-identical to the one-pixel loop, C2 0.822 -> 0.796 ms (profiles/r5/ablate_px2.log).
    // dev flag 0x10: the one-pixel loop, same box.
#ifdef MAGNET_DEV
    if constexpr (FULL && CPL == 2 && VG <= 2) {       // dev A/B 0x80: the product instance reading an fp16 quad-form map (16 B per candidate, 16-byte stride; see HALFQ)
        if (p.cost_hi && !p.gate_bits && (CV_DEV(p) & 0x80)) { hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, 8, LPU, VG, NP | 64 | 0x2000 | 0x4000>), grid, block, lds, stream, p); return hipGetLastError(); }
    }
#endif
    if constexpr (FULL && CPL == 2 && VG <= 2) {
        if (p.cost_hi && !p.gate_bits && !(CV_DEV(p) & 0x10)) {
            hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, 8, LPU, VG, NP | 64 | 0x2000>), grid, block, lds, stream, p);
            return hipGetLastError();
        }
    }
    if (p.gate_bits) hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MW2, LPU, VG, NP | 1>), grid, block, lds, stream, p);
    else if (p.cost_hi) hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MINW, LPU, VG, NP | 64>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((cv_v3_kernel<FeatT, CPL, FULL, MW2, LPU, VG, NP>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU>
static hipError_t launch_v3(const CvParams& p, hipStream_t stream) {
    // views in flight per pixel.  Two: 68 registers = 7 waves per SIMD; four views cost 78 registers = 6 waves and lose 3 % both
    // alone (0.906 vs 0.879 ms per 64 C2 frames, warm) and inside the step; one view (8 waves) has too little to overlap: 0.975
    int vg = p.V == 1 ? 1 : (p.V % 2 == 0 ? 2 : (p.V % 3 == 0 ? 3 : 2));
#ifdef MAGNET_DEV
    if (CV_DEV(p) & 0x400000) vg = 4;                                                     // dev: views per group
    if (CV_DEV(p) & 0x800000) vg = 1;
#endif
    switch (vg) {
        case 1: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 1>(p, stream);
        case 2: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 2>(p, stream);
        case 3: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 3>(p, stream);
        default: return launch_v3_v<FeatT, CPL, FULL, MINW, LPU, 4>(p, stream);
    }
}

// D > 32, candidates sampled in the kernel, quad-form (mu, sigma) map given; called by launch_cv_fast, which has checked the rest.
hipError_t launch_cv_v3(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if (p.D <= 32 || !p.src_gmq) return hipSuccess;
    // Full-resolution matching grids (w > 512: the grid-stress shapes C2L / C4L, 14 instead of 3.5 items per (pixel, view)) are
    // decided by the correlation of long item lists, where the round-2 kernel (16 items per batch, one list for all views) is
    // better: C2L 2.33 vs 2.84 ms, C4L 1.54 vs 1.63 ms (same session, warm); at w <= 304 this kernel wins (C4 0.75 vs 0.95 ms)
    if (p.w > 512 && !(CV_DEV(p) & 0x4)) return hipSuccess;                                 // dev 0x4: this kernel anyway
    if ((size_t)p.V * p.B * (size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets over all views
    {   // quad keys carry the view offset (v * B + b's views: up to (V - 1) * B * map + map - 1) and are multiplied by the texel size
        // with a 24-bit multiply; the (mu, sigma) quad address is key << 5 in 32 bits.  Larger batches / grids go to the round-2 kernel,
        // whose keys are per map (tests/test_gpu_fast_matcher.py: test_large_batch_key_range)
        const size_t map = (size_t)(p.h + 2) * (p.w + 2);
        if ((size_t)(p.V - 1) * p.B * map + map >= ((size_t)1 << 24)) return hipSuccess;
        if ((size_t)p.V * p.B * map * 32 >= ((size_t)1 << 32)) return hipSuccess;
    }
    const int nchunk = (int)(p.F * esz / 16);
    if (v3_lds_bytes(p, 4) > 64 * 1024) return hipSuccess;
    {   // the scalar block -> tile divisions by reciprocal multiplication are exact while grid * tiles < 2^32
        const uint64_t tiles = (uint64_t)((p.w + 4 * V3_NPX - 1) / (4 * V3_NPX)) * (uint64_t)p.h;
        if (tiles * (uint64_t)p.B * tiles >= ((uint64_t)1 << 32)) return hipSuccess;
    }
    *handled = true;
    // MINW = waves per SIMD the instance is compiled for: the largest that needs no scratch (78 registers for bf16 F = 64)
    if (p.feat_bf16) {
#ifdef MAGNET_DEV
        if (nchunk == 8 && (CV_DEV(p) & 0x1000)) return launch_v3<uint16_t, 2, true, 8, 4>(p, stream);    // dev: occupancy A/B
        if (nchunk == 8 && (CV_DEV(p) & 0x2000)) return launch_v3<uint16_t, 2, true, 4, 4>(p, stream);
        if (nchunk == 8 && (CV_DEV(p) & 0x10000)) return launch_v3<uint16_t, 2, true, 7, 4>(p, stream);
#endif
        if (nchunk == 8)  return launch_v3<uint16_t, 2, true, 6, 4>(p, stream);          // F = 64: 4 lanes x 32 B per (item, tap) unit
        if (nchunk <= 8)  return launch_v3<uint16_t, 1, false, 6, 8>(p, stream);
        if (nchunk <= 16) return launch_v3<uint16_t, 2, false, 5, 8>(p, stream);
    } else {
#ifdef MAGNET_DEV
        // dev: 4 lanes x 64 B per unit (4 items per pass, quad-form slots) — 74 instead of 68 registers, 6 instead of 7 waves:
        // C4 0.828 vs 0.748 ms, C2 with fp32 features 1.274 vs 1.213 ms
        if (nchunk == 16 && (CV_DEV(p) & 0x8000)) return launch_v3<float, 4, true, 5, 4>(p, stream);
#endif
        if (nchunk == 16) return launch_v3<float, 2, true, 5, 8>(p, stream);             // F = 64: 8 lanes x 32 B per unit
        if (nchunk <= 8)  return launch_v3<float, 1, false, 6, 8>(p, stream);
        if (nchunk <= 16) return launch_v3<float, 2, false, 5, 8>(p, stream);
        if (nchunk <= 32) return launch_v3<float, 4, false, 4, 8>(p, stream);
    }
    *handled = false;
    return hipSuccess;
}

}  // namespace magnet
