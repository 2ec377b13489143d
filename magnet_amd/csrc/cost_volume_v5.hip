// cost_volume_v5.hip — production matcher for D > 32, round 4: cost_volume_v3.hip with the (mu, sigma) quads PREFETCHED one unit ahead
// through LDS-DMA (homography.py:124-161 + MAGNET.py:153-156; lane = depth candidate, a wave owns 8 consecutive pixels of a row).
//
// What round 4 measured (profiles/r4/): the round-3 kernel is LATENCY-bound, not issue-bound — its time over the
// waves per SIMD w fits T = 0.33 ms + 3.9 ms / w (w = 4, 5, 7: 1.30, 1.10, 0.88 ms), i.e. at 7 waves 0.55 of the 0.88 ms are the two
// DEPENDENT memory round trips of a (pixel, view pair) iteration: candidate position -> (mu, sigma) quad load -> gate -> open quads ->
// feature loads -> dot products.  Round 4's first experiment (cost_volume_v4.hip: quads AND texels staged in LDS, correlation on the
// matrix pipe) removed instructions but holds 5 KB of LDS per unit in flight: 5 - 6 units per SIMD, 1.24 - 1.35 ms.  This kernel removes
// the FIRST round trip from the chain at 0.75 KB of LDS per unit in flight:
//   stage A (unit i = a pixel's next pair of views): geometry; one run detection over the IN-WINDOW lanes numbers the distinct 2x2
//            source quads of each view (slots); the leaders leave their quad keys in the slots; two LDS-DMA wave-loads
//            (buffer_load_dword ... lds: 8 lanes x 4 B per slot, 8 slots per instruction, ~4.5 address-unit cycles each against 2 x 16
//            for round 3's per-candidate dwordx4 pair) fetch the slots' quad-form entries into one of two slot buffers.  Not waited for.
//   stage B (unit i - 1, AFTER stage A of unit i was issued): s_waitcnt vmcnt(2) — everything older than the two DMAs just issued has
//            landed, i.e. unit i - 1's quads, which had the whole previous stage B to arrive; every candidate reads ITS slot from LDS,
//            gate, then round 3's machinery unchanged: runs of equal open quads -> items, (item, tap) dot products from VGPR loads,
//            quad-form bilinear combine.
// The loop is unrolled by two so that the two units in flight live in two fixed register sets and two fixed slot buffers.
// A unit with more than 24 distinct in-window quads (measured 3 % at C2) fetches its quads again, view by view, in synchronous rounds.
// Arithmetic and tolerance contract: exactly cost_volume_v3.hip's.
// (compiled into the DEV library only: a measured experiment, not a product path)
#ifdef MAGNET_DEV
#include "cv_runs.hpp"

namespace magnet {

constexpr int V5_CAP = 64;                         // open runs (= items) of one view group
constexpr int V5_NPASS_DEFAULT = 2;                // correlation passes whose loads are in flight together
constexpr int V5_NPX = 8;                          // reference pixels per wave
constexpr int V5_NSU = 24;                         // quad slots of one unit (both views): 3 DMA wave-loads of 8 slots
constexpr int V5_CT = 0;                           // [CAP + 1] x 16 B: the 4 tap correlations of each open run; slot CAP = dump
constexpr int V5_IT = 65 * 16;                     // [CAP + 4] x 8 B: open runs {feature byte offset, LDS address of the run's slot}
constexpr int V5_MS = (V5_IT + 68 * 8 + 15) / 16 * 16;   // [NPX] x 8 B: (mu, sigma) of the wave's reference pixels
constexpr int V5_QB = V5_MS + V5_NPX * 8;          // 2 buffers x [NSU] x 32 B: quad-form (mu, sigma) of the slots of the two units in flight
constexpr int V5_FIX = V5_QB + 2 * V5_NSU * 32;    // then: view table [Vr] x 8 B, projection table [Vr][NPX] x 32 B, reference vectors, output stage

// CPL / FULL / LPU / VG / OPT: as cv_v3_kernel
template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG, int OPT>
__global__ __launch_bounds__(256, MINW) void cv_v5_kernel(const CvParams p) {
    constexpr bool GBITS = (OPT & 1) != 0;
    constexpr bool NO_CORR = (OPT & 2) != 0;              // dev: no feature loads / dot products
    constexpr bool SPLIT = (OPT & 64) != 0;               // the split-bf16 channel-last output form only
    constexpr int NPX = V5_NPX, NSU = V5_NSU;
    constexpr int IPP = 64 / (4 * LPU);                   // items per correlation pass
    constexpr bool QF = LPU == 4;                         // an item's four taps share a 16-lane row: correlations stored in quad form
    constexpr int NPASS = ((OPT >> 8) & 15) ? ((OPT >> 8) & 15) : V5_NPASS_DEFAULT;
    constexpr int CSTR = LPU * 16;
    static_assert(VG <= 2, "two DMA wave-loads cover the first 16 slots of a unit: two views");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile, b;
    {
        const unsigned n = gridDim.x, bid = blockIdx.x;
        const unsigned qn = n / NUM_XCD, rn = n % NUM_XCD;
        const unsigned xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
        const unsigned start = (xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        const unsigned logical = start + idx;
        b = __builtin_amdgcn_readfirstlane(p.magic_tiles ? (int)__umulhi(logical, p.magic_tiles) : (int)logical);
        tile = (int)(logical - (unsigned)b * (unsigned)(p.tiles_x * p.tiles_y));
    }
    const int y = __builtin_amdgcn_readfirstlane(p.magic_tiles_x ? (int)__umulhi((unsigned)tile, p.magic_tiles_x) : tile);
    const int tx = tile - y * p.tiles_x;
    const int x_base = (tx * 4 + wv) * NPX;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const int JB = (p.D + 63) / 64;
    const uint32_t texel_bytes = (uint32_t)p.F * (uint32_t)sizeof(FeatT);
    const int nchunk = (int)(texel_bytes / 16);
    const uint32_t map_texels = (uint32_t)(Hp * Wp);
    const uint32_t vstride = (uint32_t)p.B * map_texels;

    // ---- valid views (homography.py:97), compacted ----
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    const int nval = v4_popc(vmask);
    const int Vr = (p.V + VG - 1) / VG * VG;

    // ---- wave-private LDS ----
    const int vt_bytes = Vr * 8, pv_bytes = Vr * NPX * 32, rf_bytes = NPX * (int)texel_bytes;
    const int out_bytes = (SPLIT || p.cost_hi) ? 0 : NPX * 64 * 4;
    const int wave_bytes = V5_FIX + (vt_bytes + 15) / 16 * 16 + pv_bytes + rf_bytes + out_bytes;
    const uint32_t wb = (uint32_t)(uintptr_t)(v3_lds_u8*)smem + (uint32_t)(wv * wave_bytes);
    const uint32_t vtb = wb + V5_FIX, pvb = vtb + (vt_bytes + 15) / 16 * 16, rfb = pvb + pv_bytes, outb = rfb + rf_bytes;

    for (int e = lane; e < NPX * Vr; e += 64) {
        const int q = e % NPX, c = min(e / NPX, max(nval - 1, 0));
        int v = 0;
        {
            unsigned long long m = vmask;
            for (int i = 0; i < c; ++i) m &= m - 1;
            v = m ? __builtin_ctzll(m) : 0;
        }
        const int xc = min(max(x_base, 0) + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, y, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        v3_st_f4(pvb + e * 32, make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz));
        v3_st_f4(pvb + e * 32 + 16, make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz));
        if (q == 0) v3_st_u2(vtb + (e / NPX) * 8, make_uint2((uint32_t)v * vstride, (uint32_t)v));
    }
    {
        const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) + ((size_t)b * hw + (size_t)y * p.w) * texel_bytes;
        const uint32_t row_bytes = (uint32_t)p.w * texel_bytes, seg0 = (uint32_t)max(x_base, 0) * texel_bytes;
        for (uint32_t e = (uint32_t)lane * 16u; e < (uint32_t)NPX * texel_bytes; e += 1024u)
            v3_st_u4(rfb + e, *reinterpret_cast<const uint4*>(ref_row + min(seg0 + e, row_bytes - 16u)));
        if (lane < NPX) {
            const size_t pixr = (size_t)y * p.w + min(max(x_base, 0) + lane, p.w - 1);
            v3_st_u2(wb + V5_MS + lane * 8, make_uint2(__float_as_uint(p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr]),
                                                      __float_as_uint(p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr])));
        }
        // slot buffers: zero (a slot without a leader keeps whatever it held: its DMA lanes fetch a valid or an out-of-range address)
        for (uint32_t e = (uint32_t)lane * 16u; e < (uint32_t)(2 * NSU * 32); e += 1024u) v3_st_u4(wb + V5_QB + e, make_uint4(0, 0, 0, 0));
    }
    fwave_lds_fence();
    const int npix = min(NPX, p.w - x_base);
    if (npix <= 0 || nval == 0) {
        // no valid view: the volume is zero (homography.py:97,118)
        if (npix > 0) {
            for (int jb = 0; jb < JB; ++jb)
                for (int q = 0; q < npix; ++q) {
                    const int j = jb * 64 + lane, x = x_base + q;
                    if (j >= p.D) continue;
                    if (SPLIT || p.cost_hi) {
                        const size_t e0 = (((size_t)b * Hp + (y + 1)) * Wp + (x + 1)) * (size_t)p.cost_ld + (size_t)j;
                        p.cost_hi[e0] = 0; p.cost_lo[e0] = 0;
                    } else {
                        p.cost[(size_t)b * p.cost_bstride + (size_t)j * hw + (size_t)y * p.w + x] = 0.f;
                    }
                }
        }
        return;
    }

    const uint32_t xlim = __float_as_uint((float)(p.w + 1)), ylim = __float_as_uint((float)(p.h + 1));
    const int sub = lane & (LPU - 1), tap = (lane / LPU) & 3, upair = lane / (4 * LPU);
    const uint32_t lane_src_off = (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes + (uint32_t)sub * 16u;
    const uint32_t tap4 = (uint32_t)tap * 4u;
    const float invV = 1.0f / (float)p.V;
    const float kappa = p.kappa;
    const cvr_gptr src_b = (cvr_gptr)(unsigned long long)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_feat) + (size_t)b * map_texels * texel_bytes);
    // quad-form (mu, sigma) map of frame b, all views (the view offset is part of the quad key): buffer descriptor for the LDS-DMA
    const uint32_t span = (uint32_t)(p.V - 1) * vstride + map_texels;
    const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(
        (void*)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_gmq) + (size_t)b * map_texels * 32), 0, (int)(span * 32u), 0x00020000);
    const uint32_t it_lane = wb + V5_IT + (uint32_t)upair * 8u;
    const uint32_t rf_lane = rfb + (uint32_t)sub * 16u;
    const uint32_t kq_off = (uint32_t)(lane >> 3) * 32u, qw_lane = (uint32_t)(lane & 7) * 4u;   // DMA e: slot 8 e + (lane >> 3), dword lane & 7
    const uint32_t qb0 = wb + V5_QB, qb1 = qb0 + NSU * 32;
    const int ngrp = (nval + VG - 1) / VG;
    // split output: base of the wave's first pixel pinned into SGPRs, per pixel a 32-bit byte offset
    v4_gu8* const hi_base = (SPLIT || p.cost_hi) ? v4_uniform_gptr(p.cost_hi + (((size_t)b * Hp + (y + 1)) * Wp + (x_base + 1)) * (size_t)p.cost_ld) : nullptr;
    v4_gu8* const lo_base = (SPLIT || p.cost_hi) ? v4_uniform_gptr(p.cost_lo + (((size_t)b * Hp + (y + 1)) * Wp + (x_base + 1)) * (size_t)p.cost_ld) : nullptr;
    const uint32_t ld2 = (uint32_t)p.cost_ld * 2u;

    // a unit in flight between stage A and stage B
    struct Unit {
        float bx[VG], by[VG], zw[VG];
        uint32_t sa[VG], keyf[VG], vidx[VG];
        unsigned long long W[VG];
        int ntot, q, last;
    };

    for (int jb = 0; jb < JB; ++jb) {
        const int j = jb * 64 + lane;
        const float kj = p.k[min(j, p.D - 1)];
        const unsigned long long jmask = __builtin_amdgcn_ballot_w64(j < p.D);
        float acc = 0.f;
        float dA = 0.f;                                                                     // candidate depths of stage A's pixel
        int qA = 0, gA = 0;                                                                 // stage A's next unit: pixel, view group

        // the quad DMAs of slots [8 e, 8 e + 8) of a slot buffer
        auto dma = [&](const uint32_t qbuf, const int e) {
            const uint32_t k = v3_ld_u1(qbuf + (uint32_t)(e * 256) + kq_off);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, V4_LPTR(qbuf + (uint32_t)(e * 256)), 4, (int)((k << 5) + qw_lane), 0, 0, 0);
        };

        // =================================== stage A ===================================
        auto stageA = [&](Unit& s, const uint32_t qbuf) {
            if (gA == 0) {
                const uint2 ms = v3_ld_u2(wb + V5_MS + qA * 8);
                dA = __builtin_fmaf(__uint_as_float(ms.y), kj, __uint_as_float(ms.x));     // MAGNET.py:155
            }
            const int g0 = gA * VG;
            const int nact = min(VG, nval - g0);
            const uint32_t pva = pvb + (uint32_t)((g0 * NPX + qA) * 32);
            const uint32_t vta = vtb + (uint32_t)(g0 * 8);
            int base = 0;
            unsigned long long Lp[VG];
#pragma unroll
            for (int u = 0; u < VG; ++u) {
                const float4 pa = v3_ld_f4(pva + u * (NPX * 32)), pb = v3_ld_f4(pva + u * (NPX * 32) + 16);
                const uint2 vt = v3_ld_u2(vta + u * 8);
                const float Px = __builtin_fmaf(pa.x, dA, pb.x);                           // homography.py:132
                const float Py = __builtin_fmaf(pa.y, dA, pb.y);
                const float Pz = __builtin_fmaf(pa.z, dA, pb.z);
                s.zw[u] = __builtin_fmaf(pa.w, dA, pb.w);                                  // homography.py:137-138
                const float rz = __builtin_amdgcn_rcpf(Pz);                                // homography.py:133
                const float ixs = __builtin_fmaf(Px, rz, 0.5f);
                const float iys = __builtin_fmaf(Py, rz, 0.5f);
                s.bx[u] = __builtin_amdgcn_fractf(ixs); s.by[u] = __builtin_amdgcn_fractf(iys);
                const unsigned long long wx = __builtin_amdgcn_ballot_w64(__float_as_uint(ixs) < xlim);
                const unsigned long long wy = __builtin_amdgcn_ballot_w64(__float_as_uint(iys) < ylim);
                s.W[u] = (u < nact) ? (wx & wy & jmask) : 0ull;
                // quad index relative to (frame b, view 0): truncation = floor inside the window; garbage outside it (never used)
                s.keyf[u] = __umul24(v3_cvt_u32_sat(iys), (uint32_t)Wp) + v3_cvt_u32_sat(ixs) + vt.x;
                if (GBITS) s.vidx[u] = vt.y;
                // runs of equal quads among the IN-WINDOW lanes: leader = in window and (lane 0, or another quad than the previous lane,
                // or the previous lane outside the window); slot = leaders in lanes 1 .. lane (+ the slots of the views before; the
                // view's slot 0 stays unused when lane 0 is outside the window)
                const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s.keyf[u], 0x138, 0xf, 0xf, true);   // wave_shr:1
                Lp[u] = (__builtin_amdgcn_ballot_w64(s.keyf[u] != prev) | 1ull | ~(s.W[u] << 1)) & s.W[u];
                const unsigned long long Ls = Lp[u] >> 1;
                const uint32_t cnt = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ls, 0u));
                s.sa[u] = (cnt << 5) + (qbuf + (uint32_t)(base * 32));
                base += v4_popc(Ls) + 1;
            }
            s.ntot = base; s.q = qA; s.last = (gA == ngrp - 1);
            // leaders leave their quad keys in their slots (first dword; the DMA overwrites it with the slot's data)
            if (base <= NSU) {
#pragma unroll
                for (int u = 0; u < VG; ++u) v4_st1_mask(Lp[u], s.sa[u], s.keyf[u]);
            } else {
#pragma unroll
                for (int u = 0; u < VG; ++u)
                    v4_st1_mask(Lp[u] & __builtin_amdgcn_ballot_w64(s.sa[u] - qbuf < (uint32_t)(NSU * 32)), s.sa[u], s.keyf[u]);
            }
            fwave_lds_fence();
            // exactly TWO wave-loads come last (stage B of the previous unit waits with vmcnt(2)); a third, for slots 16 .. 23, first
            if (base > 16) dma(qbuf, 2);
            dma(qbuf, 0);
            dma(qbuf, 1);
            if (++gA == ngrp) { gA = 0; ++qA; }
        };

        // =================================== stage B ===================================
        auto stageB = [&](Unit& s, const uint32_t qbuf) {
            const int q = s.q, x = x_base + q;
            float fxy[VG];
            unsigned long long Gb[VG], Lb[VG];
            uint32_t raddr[VG];
            int n_items = 0;
#pragma unroll
            for (int u = 0; u < VG; ++u) fxy[u] = s.bx[u] * s.by[u];
            auto gate_at = [&](const int u, const uint32_t sa, const unsigned long long inr) {   // homography.py:151-152,157-158
                const float4 q0 = v3_ld_f4(sa), q1 = v3_ld_f4(sa + 16);
                const float mu_w = __builtin_fmaf(fxy[u], q0.w, __builtin_fmaf(s.by[u], q0.z, __builtin_fmaf(s.bx[u], q0.y, q0.x)));
                const float sg_w = __builtin_fmaf(fxy[u], q1.w, __builtin_fmaf(s.by[u], q1.z, __builtin_fmaf(s.bx[u], q1.y, q1.x)));
                return __builtin_amdgcn_ballot_w64(__builtin_fabsf(s.zw[u] - mu_w) < sg_w * kappa) & inr;
            };
            if (s.ntot <= NSU) {
#pragma unroll
                for (int u = 0; u < VG; ++u) Gb[u] = gate_at(u, s.sa[u], s.W[u]);
            } else {
                // (rare) more distinct in-window quads than the slot buffer holds: fetch them again view by view, NSU slots per round
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int vbase = 0;
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    Gb[u] = 0ull;
                    const uint32_t pv_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s.keyf[u], 0x138, 0xf, 0xf, true);
                    const unsigned long long Lpu = (__builtin_amdgcn_ballot_w64(s.keyf[u] != pv_) | 1ull | ~(s.W[u] << 1)) & s.W[u];   // the in-window leaders again
                    const int nv = v4_popc(Lpu >> 1) + 1;
                    for (int r0 = 0; r0 < nv; r0 += NSU) {
                        const uint32_t sa = s.sa[u] - (uint32_t)((vbase + r0) * 32);
                        const unsigned long long inr = __builtin_amdgcn_ballot_w64(sa - qbuf < (uint32_t)(NSU * 32)) & s.W[u];
                        v4_st1_mask(Lpu & inr, sa, s.keyf[u]);
                        fwave_lds_fence();
                        dma(qbuf, 0);
                        if (nv - r0 > 8) dma(qbuf, 1);
                        if (nv - r0 > 16) dma(qbuf, 2);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        Gb[u] |= gate_at(u, sa, inr);
                        fwave_lds_fence();
                    }
                    vbase += nv;
                }
            }
#pragma unroll
            for (int u = 0; u < VG; ++u) {
                // runs of equal quads among the OPEN lanes: leader = open and (lane 0, or another quad than the previous lane, or the previous lane closed)
                const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s.keyf[u], 0x138, 0xf, 0xf, true);
                Lb[u] = (__builtin_amdgcn_ballot_w64(s.keyf[u] != prev) | 1ull | ~(Gb[u] << 1)) & Gb[u];
                n_items += v4_popc(Lb[u]);
                if (GBITS) {
                    if (j < p.D && s.W[u] != 0ull)
                        p.gate_bits[(((size_t)b * p.V + s.vidx[u]) * p.D + j) * hw + (size_t)y * p.w + x] = (uint8_t)v3_sel_u(Gb[u], 1u, 0u);
                }
            }
            // (item, tap) dot products of the n open runs listed in the item table -> the runs' slots (cost_volume_v3.hip)
            auto correlate = [&](const int n) {
                for (int ps = 0; ps < n; ps += IPP * NPASS) {
                    uint4 sv[NPASS][CPL];
                    uint4 rvp[CPL];                                                         // this lane's chunk(s) of the pixel's reference vector: re-read from
                    uint2 ent[NPASS];                                                       // LDS per batch (8 registers less across the unit in flight)
                    const uint32_t ita = it_lane + (uint32_t)ps * 8u;
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) ent[a] = v3_ld_u2(ita + a * (IPP * 8));
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc)
                        rvp[cc] = (FULL || (sub + LPU * cc < nchunk)) ? v3_ld_u4(rf_lane + q * texel_bytes + cc * CSTR) : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < NPASS; ++a) {
                        if (a > 0 && ps + IPP * a >= n) break;
                        const cvr_gptr sp = src_b + (ent[a].x + lane_src_off);
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
                        if (QF) part = v3_quadform16(part);
                        if (sub == 0) v3_st_f1(ent[a].y + tap4, part);
                    }
                }
            };
            // items of views [ub, ue) -> table, correlate, combine
            auto process = [&](const int ub, const int ue, const int n) {
                int base = 0;
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    if (u < ub || u >= ue) continue;
                    const unsigned long long Ls = Lb[u] >> 1;
                    const int sb = base + (int)(Lb[u] & 1ull) - 1;
                    const uint32_t cnt = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ls, 0u));
                    raddr[u] = (cnt << 4) + (wb + V5_CT + (uint32_t)(sb * 16));
                    v3_st2_mask(Lb[u], (cnt << 3) + (wb + V5_IT + (uint32_t)(sb * 8)), __umul24(s.keyf[u], texel_bytes), raddr[u]);
                    base += v4_popc(Lb[u]);
                }
                if (IPP > 1)
                    v3_st2_mask((1ull << (IPP - 1)) - 1ull, wb + V5_IT + ((uint32_t)n + (uint32_t)lane) * 8u, 0u, wb + V5_CT + V5_CAP * 16);
                fwave_lds_fence();
                if (!NO_CORR) correlate(n);
                fwave_lds_fence();
#pragma unroll
                for (int u = 0; u < VG; ++u) {
                    if (u < ub || u >= ue) continue;
                    const float4 c4 = v3_ld_f4(raddr[u]);
                    float c;
                    if (QF) {                                                              // homography.py:150,155 (grid_sample's bilinear weights, factored)
                        c = __builtin_fmaf(fxy[u], c4.w, __builtin_fmaf(s.by[u], c4.z, __builtin_fmaf(s.bx[u], c4.y, c4.x)));
                    } else {
                        const float w10 = s.bx[u] - fxy[u], w01 = s.by[u] - fxy[u];
                        const float w00 = (1.0f - s.bx[u]) - w01;
                        c = c4.x * w00;
                        c = __builtin_fmaf(c4.y, w10, c);
                        c = __builtin_fmaf(c4.z, w01, c);
                        c = __builtin_fmaf(c4.w, fxy[u], c);
                    }
                    acc += v3_sel_f(Gb[u], c, 0.f);                                        // homography.py:159,116 (fp32 here)
                }
                fwave_lds_fence();
            };
            if (n_items > 0) {
                // all views of the group at once; more than CAP open runs (only with VG > 1): view by view (a view has at most 64 runs)
                const bool split = n_items > V5_CAP;
                for (int ub = 0; ub < VG; ub += split ? 1 : VG) {
                    const int ue = split ? ub + 1 : VG;
                    int n = 0;
#pragma unroll
                    for (int u = 0; u < VG; ++u) if (u >= ub && u < ue) n += v4_popc(Lb[u]);
                    if (n > 0) process(ub, ue, n);
                }
            }
            if (s.last) {
                const float cval = acc * invV;                                             // homography.py:118,120
                acc = 0.f;
                if (SPLIT || p.cost_hi) {
                    const uint32_t off = (uint32_t)q * ld2 + (uint32_t)j * 2u;
                    if (j < p.D) {
                        const uint16_t hi = f32_to_bf16_rne(cval);
                        const uint16_t lo = f32_to_bf16_rne(cval - bf16_to_f32(hi));
                        *reinterpret_cast<v4_gu16*>(hi_base + off) = hi; *reinterpret_cast<v4_gu16*>(lo_base + off) = lo;
                    }
                } else {
                    v3_st_f1(outb + (uint32_t)(q * 64 + lane) * 4u, cval);
                    if (q == npix - 1) {
                        // ---- npix px x 64 results: LDS -> coalesced row segments of cost[b, j, y, :] ----
                        fwave_lds_fence();
                        const unsigned char* gbase = reinterpret_cast<const unsigned char*>(p.cost + (size_t)b * p.cost_bstride + (size_t)(jb * 64) * hw + (size_t)y * p.w + x_base);
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
        };

        // two units in flight in two fixed register sets / slot buffers: A(i + 1) is issued before B(i) runs
        Unit s0, s1;
        const int units = npix * ngrp;
        int itA = 1;
        stageA(s0, qb0);
        for (;;) {
            const bool m1 = itA < units;
            if (m1) { stageA(s1, qb1); ++itA; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stageB(s0, qb0);
            __builtin_amdgcn_sched_barrier(0);
            if (!m1) break;
            const bool m0 = itA < units;
            if (m0) { stageA(s0, qb0); ++itA; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stageB(s1, qb1);
            __builtin_amdgcn_sched_barrier(0);
            if (!m0) break;
        }
    }
}

static size_t v5_lds_bytes(const CvParams& p, int vg) {
    const size_t esz = p.feat_bf16 ? 2 : 4;
    const int Vr = (p.V + vg - 1) / vg * vg;
    return (size_t)4 * (V5_FIX + (Vr * 8 + 15) / 16 * 16 + Vr * V5_NPX * 32 + V5_NPX * p.F * esz + (p.cost_hi ? 0 : V5_NPX * 64 * 4));
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU, int VG>
static hipError_t launch_v5_v(const CvParams& p0, hipStream_t stream) {
    CvParams p = p0;
    p.npx = V5_NPX;
    p.tiles_x = (p.w + 4 * p.npx - 1) / (4 * p.npx);
    p.tiles_y = p.h;
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    const uint64_t nt = (uint64_t)p.tiles_x * p.tiles_y;
    p.magic_tiles = nt > 1 ? (uint32_t)((((uint64_t)1 << 32) + nt - 1) / nt) : 0u;
    p.magic_tiles_x = p.tiles_x > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)p.tiles_x - 1) / (uint64_t)p.tiles_x) : 0u;
    size_t lds = v5_lds_bytes(p, VG);
#ifdef MAGNET_DEV
    {   // dev: cap the workgroups per CU (waves per SIMD) by asking for more LDS than the kernel uses
        const int cap = (CV_DEV(p) & 0x300000) == 0x300000 ? 3 : (CV_DEV(p) & 0x200000) ? 4 : (CV_DEV(p) & 0x100000) ? 5 : 0;
        if (cap) { const size_t need = (size_t)160 * 1024 / (cap + 1) + 512; if (lds < need) lds = need; }
    }
    if (p.cost_hi && (CV_DEV(p) & 0x200)) { hipLaunchKernelGGL((cv_v5_kernel<FeatT, CPL, FULL, (MINW > 5 ? 5 : MINW), LPU, VG, 2 | 64>), grid, block, lds, stream, p); return hipGetLastError(); }   // no dot products (timing only)
#endif
    constexpr int NP = (CPL >= 4 ? 1 : V5_NPASS_DEFAULT) << 8;
    constexpr int MW2 = CPL >= 4 ? 4 : (MINW > 5 ? 5 : MINW);
    if (p.gate_bits) hipLaunchKernelGGL((cv_v5_kernel<FeatT, CPL, FULL, MW2, LPU, VG, NP | 1>), grid, block, lds, stream, p);
    else if (p.cost_hi) hipLaunchKernelGGL((cv_v5_kernel<FeatT, CPL, FULL, MINW, LPU, VG, NP | 64>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((cv_v5_kernel<FeatT, CPL, FULL, MW2, LPU, VG, NP>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <typename FeatT, int CPL, bool FULL, int MINW, int LPU>
static hipError_t launch_v5(const CvParams& p, hipStream_t stream) {
    if (p.V == 1) return launch_v5_v<FeatT, CPL, FULL, MINW, LPU, 1>(p, stream);
    return launch_v5_v<FeatT, CPL, FULL, MINW, LPU, 2>(p, stream);
}

// D > 32, candidates sampled in the kernel, quad-form (mu, sigma) map given; called by launch_cv_fast, which has checked the rest.
hipError_t launch_cv_v5(const CvParams& p, hipStream_t stream, bool* handled) {
    *handled = false;
    const size_t esz = p.feat_bf16 ? 2 : 4;
    if (p.D <= 32 || !p.src_gmq) return hipSuccess;
    if (p.w > 512) return hipSuccess;                                                        // long epipolar segments: cost_volume_fast64.hip (see launch_cv_v3)
    if ((size_t)p.V * p.B * (size_t)(p.h + 2) * (p.w + 2) * p.F * esz >= ((size_t)1 << 32)) return hipSuccess;   // 32-bit byte offsets over all views
    {
        const size_t map = (size_t)(p.h + 2) * (p.w + 2);
        if ((size_t)(p.V - 1) * p.B * map + map >= ((size_t)1 << 24)) return hipSuccess;  // 24-bit keys over all views of a frame
        if ((size_t)p.V * p.B * map * 32 >= ((size_t)1 << 31)) return hipSuccess;          // buffer range of the quad map
    }
    const int nchunk = (int)(p.F * esz / 16);
    if (v5_lds_bytes(p, 2) > 64 * 1024) return hipSuccess;
    {
        const uint64_t tiles = (uint64_t)((p.w + 4 * V5_NPX - 1) / (4 * V5_NPX)) * (uint64_t)p.h;
        if (tiles * (uint64_t)p.B * tiles >= ((uint64_t)1 << 32)) return hipSuccess;
    }
    *handled = true;
    if (p.feat_bf16) {
        if (nchunk == 8)  return launch_v5<uint16_t, 2, true, 6, 4>(p, stream);          // F = 64: 4 lanes x 32 B per (item, tap) unit
        if (nchunk <= 8)  return launch_v5<uint16_t, 1, false, 6, 8>(p, stream);
        if (nchunk <= 16) return launch_v5<uint16_t, 2, false, 5, 8>(p, stream);
    } else {
        if (nchunk == 16) return launch_v5<float, 2, true, 5, 8>(p, stream);             // F = 64: 8 lanes x 32 B per unit
        if (nchunk <= 8)  return launch_v5<float, 1, false, 6, 8>(p, stream);
        if (nchunk <= 16) return launch_v5<float, 2, false, 5, 8>(p, stream);
        if (nchunk <= 32) return launch_v5<float, 4, false, 4, 8>(p, stream);
    }
    *handled = false;
    return hipSuccess;
}

}  // namespace magnet
#endif  // MAGNET_DEV
