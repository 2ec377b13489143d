// cost_volume_v4.hip — production matcher for D > 32 with bf16 F = 64 features, round 4: an ASYNCHRONOUS two-stage form of the
// lane = candidate matcher (homography.py:124-161 + MAGNET.py:153-156), built around what round 3's counters said bounds
// cost_volume_v3.hip: 74 vector instructions, 4.4 vector-memory wave-loads (2 of them the per-candidate 32-byte (mu, sigma) quad
// loads) and two dependent memory round trips per (pixel, view), hidden by occupancy alone (waves waiting 68 % of their cycles).
//
// Decomposition (a wave owns 8 consecutive pixels of a row, a UNIT is one (pixel, valid view); iteration i runs stage B of
// unit i - 1, then stage A of unit i):
//   stage A  geometry of the 64 candidates; ONE run detection over the in-window lanes numbers the distinct 2x2 source quads
//            (slots); the leaders publish their quad keys in LDS; then the wave issues LDS-DMA loads
//            (buffer_load ... lds: global -> LDS, no VGPR, no ds_write) for
//              - the slots' quad-form (mu, sigma) entries: 8 lanes x 4 B per slot, 8 slots per wave-instruction (a dword
//                wave-load costs the CU's address unit ~4.4 cycles against 2 x 16 for the per-candidate dwordx4 pair), and
//              - the 4 x 128-byte feature texels of the first 8 slots, XOR-swizzled on the SOURCE address so that the matrix
//                pipe's operand reads below are bank-conflict-free.
//            Nothing of this is waited for: the only state that crosses to stage B is 5 registers per lane.
//   stage B  (one iteration later, after s_waitcnt vmcnt(0): the DMAs had a whole iteration of every resident wave to land)
//            every candidate reads ITS slot's quad from LDS (2 x ds_read_b128 instead of 2 vector-memory loads), gate =
//            |z - mu_w| < kappa sigma_w; the (slot, tap) x reference-pixel correlations run on the MATRIX pipe:
//            v_mfma_f32_16x16x32_bf16 with A = the wave's 8 reference vectors (registers, loaded once per wave), B = 16 staged
//            texels read from LDS — round 2's matrix-pipe attempt loaded the MFMA operand layout from GLOBAL memory and was
//            L1-tag-bound (16 lines per quarter-wave); from LDS the layout is free.  The 16 lanes that hold the current pixel's
//            row write the 16 correlations to the slots with one ds_write_b32; every candidate reads its slot's four taps and
//            combines them with the bilinear weights.
// No second run detection (the feature texels are fetched for every in-window quad, open or not: 4.5 instead of 2.7 items
// per (pixel, view) at C2, paid in address-unit cycles that the quad path freed), no per-candidate vector-memory load at all,
// no dot2c / DPP reduction.  A view that touches more than 32 distinct quads, or more than 8, takes further ROUNDS / BATCHES
// inside stage B with synchronous DMAs (measured share at C2: 3 % / 14 % of the units).
// Arithmetic and tolerance contract: as cost_volume_v3.hip / cost_volume_fast.hip (fma-contracted geometry, one v_rcp_f32,
// padded-map texel coordinates, fp32 view sum); the channel sum of a tap is the matrix pipe's fp32 accumulation of exact
// bf16 x bf16 products, the bilinear combine uses the difference-form weights (fp32 rounding only: homography.py:150-152,155-159).
// (compiled into the DEV library only: a measured experiment, not a product path)
#ifdef MAGNET_DEV
#include "cv_runs.hpp"

namespace magnet {

constexpr int V4_NPX = 8;                          // reference pixels per wave
constexpr int V4_NS = 32;                          // slots (distinct in-window quads) per round (template parameter NS of the kernel; 16 in a dev variant)
constexpr int V4_NB = 8;                           // slots whose texels are staged per batch (= 2 MFMA blocks of 4 slots x 4 taps)
// wave-private LDS (bytes from the wave's base)
constexpr int V4_ST = 0;                           // feature staging: NB slots x 4 texels x 128 B
constexpr int V4_QS = V4_NB * 512;                 // quad slots: NS x 32 B {mu quad form, sigma quad form}
constexpr int v4_cs(int ns) { return V4_QS + ns * 32; }   // per slot 32 B: {c00, c10, c01, c11, quad key, -, -, -}
constexpr int v4_tb(int ns) { return V4_QS + ns * 64; }   // unit table [NPX x valid views] x 48 B: projection terms, view offset, (mu, sigma) of the pixel
constexpr int V4_UNIT = 48;
constexpr int V4_FPAD = 4096;                      // the feature descriptor starts this many bytes BEFORE the frame's first texel (see dma_feats)

// OPT bit 0: write the gate bits (parity tests); bit 6: split-bf16 channel-last output only
template <int OPT, int MINW, int NS>
__global__ __launch_bounds__(256, MINW) void cv_v4_kernel(const CvParams p) {
    constexpr int V4_CS = v4_cs(NS), V4_TB = v4_tb(NS);
    constexpr bool GBITS = (OPT & 1) != 0;
    constexpr bool SPLIT = (OPT & 64) != 0;
    constexpr int NPX = V4_NPX, NB = V4_NB;
    constexpr uint32_t TB = 128;                                                           // texel bytes: F = 64 bf16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile, b;
    {   // block -> (frame, row, segment): cost_volume_v3.hip's XCD-aware raster order, divisions by host-computed reciprocals
        const unsigned n = gridDim.x, bid = blockIdx.x;
        const unsigned qn = n / NUM_XCD, rn = n % NUM_XCD;
        const unsigned xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
        const unsigned start = (xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
        const unsigned logical = start + idx;
        // (readfirstlane: the high multiply runs on the vector unit, and everything derived from b / y — buffer descriptors, output
        // addresses — would otherwise be carried as per-lane values)
        b = __builtin_amdgcn_readfirstlane(p.magic_tiles ? (int)__umulhi(logical, p.magic_tiles) : (int)logical);
        tile = (int)(logical - (unsigned)b * (unsigned)(p.tiles_x * p.tiles_y));
    }
    const int y = __builtin_amdgcn_readfirstlane(p.magic_tiles_x ? (int)__umulhi((unsigned)tile, p.magic_tiles_x) : tile);
    const int tx = tile - y * p.tiles_x;
    const int x_base = (tx * 4 + wv) * NPX;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const int JB = (p.D + 63) / 64;
    const uint32_t map_texels = (uint32_t)(Hp * Wp);
    const uint32_t vstride = (uint32_t)p.B * map_texels;                                   // texels between the views of a frame (view-major)
    const int npix = min(NPX, p.w - x_base);                                               // pixels of the segment inside the row (may be <= 0)

    // ---- valid views (homography.py:97), compacted ----
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    const int nval = v4_popc(vmask);

    // ---- wave-private LDS ----
    const int wave_bytes = V4_TB + p.V * NPX * V4_UNIT;
    const uint32_t wb = (uint32_t)(uintptr_t)(v3_lds_u8*)smem + (uint32_t)(wv * wave_bytes);
    const uint32_t tbl = wb + V4_TB;
    const uint32_t qbase = wb + V4_QS;

    // unit table, unit = (pixel q, c-th valid view), q-major: {(K R) ray, (R ray)_z | K t, t_z | texel offset of the view, mu, sigma, view}
    for (int e = lane; e < NPX * nval; e += 64) {
        const int q = e / nval, c = e - q * nval;
        int v = 0;
        {
            unsigned long long m = vmask;
            for (int i = 0; i < c; ++i) m &= m - 1;
            v = m ? __builtin_ctzll(m) : 0;
        }
        const int xc = min(x_base + q, p.w - 1);
        float r0, r1, r2;
        load_ray(p, b, hw, xc, y, r0, r1, r2);
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        const size_t pixr = (size_t)y * p.w + xc;
        v3_st_f4(tbl + e * V4_UNIT, make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz));
        v3_st_f4(tbl + e * V4_UNIT + 16, make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz));
        v3_st_u4(tbl + e * V4_UNIT + 32, make_uint4((uint32_t)v * vstride, __float_as_uint(p.ref_gmm[((size_t)b * 2 + 0) * hw + pixr]),
                                                    __float_as_uint(p.ref_gmm[((size_t)b * 2 + 1) * hw + pixr]), (uint32_t)v));
    }
    // slot table: zero keys (a stale key is only ever a key some leader wrote: the DMA lanes past a unit's last slot fetch valid memory)
    if (lane * 16 < NS * 32) v3_st_u4(wb + V4_CS + lane * 16, make_uint4(0, 0, 0, 0));
    // ---- matrix-pipe A operand: the wave's reference vectors.  Row i of the 16 x 16 tile holds pixel 2 * (i >> 2) + (i & 1), so
    // that pixel q's correlations come out in accumulator register q & 1 of the 16 lanes 16 * (q >> 1) .. + 15 (C layout: row =
    // 4 * (lane >> 4) + register, column = lane & 15); rows with (i & 3) >= 2 repeat their pair's pixels and are never read.
    fbf16x8_t aref[2];
    {
        const int i = lane & 15, g = lane >> 4;
        const int xr = min(max(x_base, 0) + 2 * (i >> 2) + (i & 1), p.w - 1);
        const unsigned char* rp = reinterpret_cast<const unsigned char*>(p.ref_feat) + ((size_t)b * hw + (size_t)y * p.w + xr) * TB + (uint32_t)g * 16u;
        aref[0] = __builtin_bit_cast(fbf16x8_t, *reinterpret_cast<const uint4*>(rp));          // channels 8 g .. 8 g + 7
        aref[1] = __builtin_bit_cast(fbf16x8_t, *reinterpret_cast<const uint4*>(rp + 64));     // channels 32 + 8 g ..
    }
    fwave_lds_fence();
    if (npix <= 0) return;

    // ---- buffer descriptors over ALL views of frame b (view-major: view v starts v * vstride texels further; the unit table's view
    // offset is part of the quad key).  The feature descriptor starts V4_FPAD bytes before the frame's first texel: the four DMAs of a
    // batch share ONE M0 (LDS base) and differ by the instruction's immediate offset d * 1024, which the hardware adds to the LDS
    // AND the global address — the per-lane offsets carry + V4_FPAD - d * 1024, so every effective global offset is >= V4_FPAD.
    const uint32_t span = (uint32_t)(p.V - 1) * vstride + map_texels;
    const __amdgpu_buffer_rsrc_t rsrc_f = __builtin_amdgcn_make_buffer_rsrc(
        (void*)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_feat) + (size_t)b * map_texels * TB - V4_FPAD), 0, (int)(span * TB + V4_FPAD), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_q = __builtin_amdgcn_make_buffer_rsrc(
        (void*)v4_uniform_ptr(reinterpret_cast<const unsigned char*>(p.src_gmq) + (size_t)b * map_texels * 32), 0, (int)(span * 32u), 0x00020000);

    // ---- per-lane constants ----
    const uint32_t xlim = __float_as_uint((float)(p.w + 1)), ylim = __float_as_uint((float)(p.h + 1));   // see cost_volume_fast.hip
    const float kappa = p.kappa, invV = 1.0f / (float)p.V;
    // quad DMA e: lane -> (slot 8 e + (lane >> 3), dword lane & 7)
    const uint32_t kq_lane = wb + V4_CS + 16 + (uint32_t)(lane >> 3) * 32u;                // + e * 256
    const uint32_t qw_lane = (uint32_t)(lane & 7) * 4u;
    // feature DMA d: lane -> staged texel row r = 8 d + (lane >> 3) = slot 2 d + (lane >> 5), tap (lane >> 3) & 3; the lane fills
    // 16-byte position lane & 7 of the row with channel chunk (lane & 7) ^ ((r >> 1) & 7)  [(r >> 1) & 7 = 4 (d & 1) + (lane >> 4)]
    const uint32_t kf_lane = wb + V4_CS + 16 + (uint32_t)(lane >> 5) * 32u;                // + d * 64
    const uint32_t tapi = (uint32_t)(lane >> 3) & 3u;
    const uint32_t ft_even = ((tapi & 1u) + (tapi >> 1) * (uint32_t)Wp) * TB + (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);
    const uint32_t ft0 = ft_even + (V4_FPAD - 0), ft1 = (ft_even ^ 64u) + (V4_FPAD - 1024), ft2 = ft_even + (V4_FPAD - 2048), ft3 = (ft_even ^ 64u) + (V4_FPAD - 3072);
    // matrix-pipe B operand: lane -> (column n = lane & 15 = staged texel row, channel chunk g + 4 s, g = lane >> 4)
    const uint32_t bf_lane0 = wb + V4_ST + (uint32_t)(lane & 15) * TB + (uint32_t)(((lane >> 4) ^ (((lane & 15) >> 1) & 7)) * 16);        // channels 8 g ..
    const uint32_t bf_lane1 = wb + V4_ST + (uint32_t)(lane & 15) * TB + (uint32_t)((((lane >> 4) + 4) ^ (((lane & 15) >> 1) & 7)) * 16);  // channels 32 + 8 g ..
    // correlation store: column n -> slot (n >> 2) of the block, tap n & 3
    const uint32_t cw_lane = wb + V4_CS + (uint32_t)((lane & 15) >> 2) * 32u + (uint32_t)(lane & 3) * 4u;

    // LDS-DMA issue (the slot-key reads come first, then the DMAs back to back).  Control flow is kept to plain `if`s: every scalar
    // compare / branch / mask move is an issue slot of the wave, and round 4's first version spent 102 of them per unit.
    auto dma_quads_more = [&](const int nq) {                                                // quad entries of slots [8, nq), 8 < nq <= NS
        const uint32_t k1 = v3_ld_u1(kq_lane + 256);
        uint32_t k2 = 0u, k3 = 0u;
        if (nq > 16) { k2 = v3_ld_u1(kq_lane + 512); k3 = v3_ld_u1(kq_lane + 768); }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, V4_LPTR(qbase), 4, (int)((k1 << 5) + qw_lane - 256u), 0, 256, 0);
        if (nq > 16) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, V4_LPTR(qbase), 4, (int)((k2 << 5) + qw_lane - 512u), 0, 512, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, V4_LPTR(qbase), 4, (int)((k3 << 5) + qw_lane - 768u), 0, 768, 0);
        }
    };
    // quad entries of slots [0, 8) and the texels of slots [slot0, slot0 + 4), + 4 more slots' texels when nf > 4 (a DMA pair per four
    // slots whether or not all four exist: an unused slot's stale key fetches valid memory that nobody reads)
    auto dma_unit = [&](const int nf, const uint32_t slot0_off, const bool quads) {
        const uint32_t ka = kf_lane + slot0_off;
        uint32_t kq0 = 0u;
        if (quads) kq0 = v3_ld_u1(kq_lane);
        const uint32_t k0 = v3_ld_u1(ka), k1 = v3_ld_u1(ka + 64);
        if (quads) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_q, V4_LPTR(qbase), 4, (int)((kq0 << 5) + qw_lane), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, V4_LPTR(wb + V4_ST), 16, (int)((k0 << 7) + ft0), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, V4_LPTR(wb + V4_ST), 16, (int)((k1 << 7) + ft1), 0, 1024, 0);
        if (nf > 4) {
            const uint32_t k2 = v3_ld_u1(ka + 128), k3 = v3_ld_u1(ka + 192);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, V4_LPTR(wb + V4_ST), 16, (int)((k2 << 7) + ft2), 0, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_f, V4_LPTR(wb + V4_ST), 16, (int)((k3 << 7) + ft3), 0, 3072, 0);
        }
    };

    // output bases of the wave's first pixel, pinned into SGPRs; per pixel only a 32-bit byte offset is added
    v4_gu8* const hi_base = (SPLIT || p.cost_hi) ? v4_uniform_gptr(p.cost_hi + (((size_t)b * Hp + (y + 1)) * Wp + (x_base + 1)) * (size_t)p.cost_ld) : nullptr;
    v4_gu8* const lo_base = (SPLIT || p.cost_hi) ? v4_uniform_gptr(p.cost_lo + (((size_t)b * Hp + (y + 1)) * Wp + (x_base + 1)) * (size_t)p.cost_ld) : nullptr;
    v4_gu8* const nchw_base = (SPLIT || p.cost_hi) ? nullptr : v4_uniform_gptr(p.cost + (size_t)b * p.cost_bstride + (size_t)y * p.w + x_base);
    const uint32_t ld2 = (uint32_t)p.cost_ld * 2u, hw4 = (uint32_t)hw * 4u;

    for (int jb = 0; jb < JB; ++jb) {
        const int j = jb * 64 + lane;
        const float kj = p.k[min(j, p.D - 1)];
        const unsigned long long jmask = __builtin_amdgcn_ballot_w64(j < p.D);

        if (nval == 0) {                                                                   // no valid view: the volume is zero (homography.py:97,118)
            for (int q = 0; q < npix; ++q) {
                if (SPLIT || p.cost_hi) {
                    const uint32_t off = (uint32_t)q * ld2 + (uint32_t)j * 2u;
                    if (j < p.D) { *reinterpret_cast<v4_gu16*>(hi_base + off) = 0; *reinterpret_cast<v4_gu16*>(lo_base + off) = 0; }
                } else if (j < p.D) {
                    *reinterpret_cast<v4_gf32*>(nchw_base + ((uint32_t)j * hw4 + (uint32_t)q * 4u)) = 0.f;
                }
            }
            continue;
        }

        // ---- unit state handed from stage A to stage B ----
        float s_bx = 0.f, s_by = 0.f, s_zw = 0.f;
        uint32_t s_sa = 0u, s_key = 0u, s_v = 0u;
        unsigned long long s_W = 0ull, s_L = 0ull;
        int s_n = 0;
        float acc = 0.f;
        uint32_t tva = tbl;                                                                 // stage A's unit-table address (uniform)
        int cB = 0, qB = 0;                                                                 // stage B: views done of its pixel, its pixel
        unsigned long long cmask = 0xffffull;                                               // lanes / accumulator register that hold pixel qB's correlations
        bool csel = false;
        const int units = npix * nval;

        // one MFMA block: 16 staged texels (4 slots x 4 taps) x the wave's reference pixels; pixel qB's row -> the slots' correlation fields
        // (plain C++ select on the accumulator: the compiler knows the MFMA -> VALU read hazard, an inline-asm consumer would not be padded)
        auto corr_block = [&](const int blk, const int slot0) {
            const fbf16x8_t b0 = __builtin_bit_cast(fbf16x8_t, v3_ld_u4(bf_lane0 + (uint32_t)(blk * 2048)));
            const fbf16x8_t b1 = __builtin_bit_cast(fbf16x8_t, v3_ld_u4(bf_lane1 + (uint32_t)(blk * 2048)));
            ff32x4_t c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aref[0], b0, ff32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aref[1], b1, c, 0, 0, 0);
            const float cv = csel ? c.y : c.x;
            v4_st1_mask(cmask, cw_lane + (uint32_t)((slot0 + blk * 4) * 32), __float_as_uint(cv));
        };
        // gate of the lanes `inr` against their slot's quad at sa; returns the open lanes (homography.py:151-152,157-158)
        auto gate = [&](const uint32_t sa, const unsigned long long inr, const float fxy) {
            const float4 q0 = v3_ld_f4(sa), q1 = v3_ld_f4(sa + 16);
            const float mu_w = __builtin_fmaf(fxy, q0.w, __builtin_fmaf(s_by, q0.z, __builtin_fmaf(s_bx, q0.y, q0.x)));
            const float sg_w = __builtin_fmaf(fxy, q1.w, __builtin_fmaf(s_by, q1.z, __builtin_fmaf(s_bx, q1.y, q1.x)));
            return __builtin_amdgcn_ballot_w64(__builtin_fabsf(s_zw - mu_w) < sg_w * kappa) & inr;
        };
        // bilinear combine of the open lanes' four tap correlations (homography.py:150,155,159; fp32 view sum: homography.py:116)
        auto combine = [&](const uint32_t sa, const unsigned long long G, const float fxy) {
            const float4 c4 = v3_ld_f4(sa + NS * 32);
            const float w10 = s_bx - fxy, w01 = s_by - fxy;
            const float w00 = (1.0f - s_bx) - w01;
            float c = c4.x * w00;
            c = __builtin_fmaf(c4.y, w10, c);
            c = __builtin_fmaf(c4.z, w01, c);
            c = __builtin_fmaf(c4.w, fxy, c);
            acc += v3_sel_f(G, c, 0.f);
        };

        // =================================== stage B: the unit stage A issued one iteration ago ===================================
        auto stageB = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // that unit's DMAs have landed
            const float fxy = s_bx * s_by;
            unsigned long long inr = s_W;
            if (s_n > NS) inr &= __builtin_amdgcn_ballot_w64(s_sa - qbase < (uint32_t)(NS * 32));   // (rare) first round: slots [0, NS)
            const unsigned long long G = gate(s_sa, inr, fxy);
            unsigned long long Gall = G;
            if (G != 0ull) {
                corr_block(0, 0);                                                           // slots 0..3 (+ 4..7): prefetched by stage A
                if (s_n > 4) corr_block(1, 0);
                if (s_n > NB) {                                                             // further batches of NB slots: synchronous DMAs
                    const int nr = min(s_n, NS);
                    for (int bb = NB; bb < nr; bb += NB) {
                        dma_unit(nr - bb, (uint32_t)(bb * 32), false);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        corr_block(0, bb);
                        if (nr - bb > 4) corr_block(1, bb);
                    }
                }
                fwave_lds_fence();
                combine(s_sa, G, fxy);
                fwave_lds_fence();
            }
            if (s_n > NS) {                                                                 // (rare) further rounds of NS slots, all synchronous
                for (int r0 = NS; r0 < s_n; r0 += NS) {
                    const uint32_t sa = s_sa - (uint32_t)(r0 * 32);
                    const unsigned long long in2 = __builtin_amdgcn_ballot_w64(sa - qbase < (uint32_t)(NS * 32)) & s_W;
                    const int nr = min(s_n - r0, NS);
                    v4_st1_mask(s_L & in2, sa + (NS * 32 + 16), s_key);                     // keys of this round's slots -> slot table
                    fwave_lds_fence();
                    dma_unit(nr, 0u, true);
                    if (nr > 8) dma_quads_more(nr);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long G2 = gate(sa, in2, fxy);
                    Gall |= G2;
                    if (G2 == 0ull) continue;
                    for (int bb = 0; bb < nr; bb += NB) {
                        if (bb > 0) {
                            dma_unit(nr - bb, (uint32_t)(bb * 32), false);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        corr_block(0, bb);
                        if (nr - bb > 4) corr_block(1, bb);
                    }
                    fwave_lds_fence();
                    combine(sa, G2, fxy);
                    fwave_lds_fence();
                }
            }
            if (GBITS) {
                if (j < p.D)
                    p.gate_bits[(((size_t)b * p.V + s_v) * p.D + j) * hw + (size_t)y * p.w + (x_base + qB)] = (uint8_t)v3_sel_u(Gall, 1u, 0u);
            }
            if (++cB == nval) {                                                             // the pixel's last view: write its 64 costs
                const float cval = acc * invV;                                              // homography.py:118,120
                acc = 0.f;
                if (SPLIT || p.cost_hi) {
                    // split-bf16 channel-last output for the conv kernel: lanes = 64 consecutive channels of one padded-grid row
                    const uint32_t off = (uint32_t)qB * ld2 + (uint32_t)j * 2u;
                    if (j < p.D) {
                        const uint16_t hi = f32_to_bf16_rne(cval);
                        const uint16_t lo = f32_to_bf16_rne(cval - bf16_to_f32(hi));
                        *reinterpret_cast<v4_gu16*>(hi_base + off) = hi; *reinterpret_cast<v4_gu16*>(lo_base + off) = lo;
                    }
                } else if (j < p.D) {
                    *reinterpret_cast<v4_gf32*>(nchw_base + ((uint32_t)j * hw4 + (uint32_t)qB * 4u)) = cval;   // cost[b, j, y, x]
                }
                cB = 0; ++qB;
                cmask = 0xffffull << (16 * (qB >> 1));
                csel = (qB & 1) != 0;
            }
        };
        // =================================== stage A: geometry, slots, DMA issue of the next unit ===================================
        auto stageA = [&]() {
            const float4 pa = v3_ld_f4(tva), pb = v3_ld_f4(tva + 16);
            const uint4 pc = v3_ld_u4(tva + 32);
            tva += V4_UNIT;
            const float d = __builtin_fmaf(__uint_as_float(pc.z), kj, __uint_as_float(pc.y));   // MAGNET.py:155
            const float Px = __builtin_fmaf(pa.x, d, pb.x);                                 // homography.py:132
            const float Py = __builtin_fmaf(pa.y, d, pb.y);
            const float Pz = __builtin_fmaf(pa.z, d, pb.z);
            s_zw = __builtin_fmaf(pa.w, d, pb.w);                                           // homography.py:137-138
            const float rz = __builtin_amdgcn_rcpf(Pz);                                     // homography.py:133
            const float ixs = __builtin_fmaf(Px, rz, 0.5f);
            const float iys = __builtin_fmaf(Py, rz, 0.5f);
            s_bx = __builtin_amdgcn_fractf(ixs); s_by = __builtin_amdgcn_fractf(iys);
            const unsigned long long wx = __builtin_amdgcn_ballot_w64(__float_as_uint(ixs) < xlim);
            const unsigned long long wy = __builtin_amdgcn_ballot_w64(__float_as_uint(iys) < ylim);
            s_W = wx & wy & jmask;
            // quad index over all views of the frame: truncation = floor inside the window (garbage outside it, never used)
            s_key = __umul24(v3_cvt_u32_sat(iys), (uint32_t)Wp) + v3_cvt_u32_sat(ixs) + pc.x;
            // runs of equal quads among the in-window lanes: leader = in window and (lane 0, or another quad than the previous lane, or
            // the previous lane outside the window)
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s_key, 0x138, 0xf, 0xf, true);   // wave_shr:1
            s_L = (__builtin_amdgcn_ballot_w64(s_key != prev) | 1ull | ~(s_W << 1)) & s_W;
            // slot of a lane = leaders in lanes 1 .. lane (slot 0 stays unused when lane 0 is outside the window: one key-less slot
            // instead of three scalar instructions per unit; no lane in the window: one key-less slot and a closed gate)
            const unsigned long long Ls = s_L >> 1;
            const uint32_t cnt = __builtin_amdgcn_mbcnt_hi((uint32_t)(Ls >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Ls, 0u));
            s_sa = (cnt << 5) + qbase;
            if (GBITS) s_v = pc.w;
            s_n = v4_popc(Ls) + 1;
            unsigned long long L0 = s_L;
            if (s_n > NS) L0 &= __builtin_amdgcn_ballot_w64(cnt < (uint32_t)NS);
            v4_st1_mask(L0, s_sa + (NS * 32 + 16), s_key);                                   // leaders publish their quad keys
            fwave_lds_fence();
            dma_unit(s_n, 0u, true);
            if (s_n > 8) dma_quads_more(min(s_n, NS));
        };

        stageA();
        for (int it = 1; it < units; ++it) { stageB(); stageA(); }
        stageB();
    }
}

static size_t v4_lds_bytes(const CvParams& p, int ns = V4_NS) { return (size_t)4 * (v4_tb(ns) + p.V * V4_NPX * V4_UNIT); }

// D > 32, bf16 F = 64 features, candidates sampled in the kernel, quad-form (mu, sigma) map given; called by launch_cv_fast
hipError_t launch_cv_v4(const CvParams& p0, hipStream_t stream, bool* handled) {
    *handled = false;
    if (p0.D <= 32 || !p0.src_gmq || !p0.feat_bf16 || p0.F != 64) return hipSuccess;
    if (p0.w > 512) return hipSuccess;                                                       // long epipolar segments: cost_volume_fast64.hip (see launch_cv_v3)
    const size_t map = (size_t)(p0.h + 2) * (p0.w + 2);
    if (map >= ((size_t)1 << 24)) return hipSuccess;                                       // 24-bit quad keys
    if (((size_t)(p0.V - 1) * p0.B * map + map) * 128 >= ((size_t)1 << 31)) return hipSuccess;   // buffer range / 32-bit byte offsets over all views
    if (v4_lds_bytes(p0) > 64 * 1024) return hipSuccess;
    CvParams p = p0;
    p.npx = V4_NPX;
    p.tiles_x = (p.w + 4 * p.npx - 1) / (4 * p.npx);
    p.tiles_y = p.h;
    const uint64_t nt = (uint64_t)p.tiles_x * p.tiles_y;
    if (nt * (uint64_t)p.B * nt >= ((uint64_t)1 << 32)) return hipSuccess;                  // exact reciprocal divisions
    p.magic_tiles = nt > 1 ? (uint32_t)((((uint64_t)1 << 32) + nt - 1) / nt) : 0u;
    p.magic_tiles_x = p.tiles_x > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)p.tiles_x - 1) / (uint64_t)p.tiles_x) : 0u;
    const dim3 grid((unsigned)(nt * p.B)), block(256);
    size_t lds = v4_lds_bytes(p);
    *handled = true;
#ifdef MAGNET_DEV
    if (p.cost_hi && (CV_DEV(p) & 0x1000)) {   // (with 0x8)                                                  // dev: 16 slots per round (6 workgroups per CU)
        hipLaunchKernelGGL((cv_v4_kernel<64, 6, 16>), grid, block, v4_lds_bytes(p, 16), stream, p); return hipGetLastError();
    }
    {   // dev: cap the workgroups per CU by asking for more LDS than the kernel uses
        const int cap = (CV_DEV(p) & 0x200000) ? 3 : (CV_DEV(p) & 0x100000) ? 4 : 0;
        if (cap) { const size_t need = (size_t)160 * 1024 / (cap + 1) + 512; if (lds < need) lds = need; }
    }
#endif
    if (p.gate_bits) hipLaunchKernelGGL((cv_v4_kernel<1, 4, V4_NS>), grid, block, lds, stream, p);
    else if (p.cost_hi) hipLaunchKernelGGL((cv_v4_kernel<64, 5, V4_NS>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((cv_v4_kernel<0, 5, V4_NS>), grid, block, lds, stream, p);
    return hipGetLastError();
}

}  // namespace magnet
#endif  // MAGNET_DEV
