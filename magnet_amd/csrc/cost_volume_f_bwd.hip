// cost_volume_f_bwd.hip — backward of the plain feature-matching volume (mode 1 of magnet_cost_volume_cw).
//
// Reference: est_costvolume_F / _compute_cost_F (models/submodules/homography.py:10-75) is used while
// TRAINING the F-Net (train_scripts/train_FNET/train.py), where autograd differentiates
//     cost[b,j,p] = 1/V * sum_v valid_v * sum_f ref[f,p] * sum_tap w_tap(j,p,v) * src_v[f, quad(j,p,v)+tap]
// with respect to the two feature maps (depth bins, poses and intrinsics carry no gradient):
//     grad_ref[f,p]        = sum_{j,v,tap}  g[j,p]/V * w_tap * src_v[f, quad+tap]
//     grad_src_v[f, texel] = sum_{(j,p,tap) -> texel}  g[j,p]/V * w_tap * ref[f,p]
// The same "lane = candidate" mapping as the forward kernel (cost_volume_cand.hip): the 64 candidates
// of one pixel land on a handful of distinct quads (ITEMS).  Lanes first reduce g*w per (item, tap) into
// a 4-float LDS cell (ds_add_f32), then (item, tap) units sweep the channel vectors once:
//   * grad_ref is accumulated in registers over all candidates and views of the pixel and written ONCE,
//   * grad_src receives ONE atomic per (item, tap, channel) instead of one per (candidate, tap, channel)
//     (~10x fewer than a per-sample scatter at D = 64..80).
// Geometry is the forward's, bit for bit (warp_math.hpp), so forward and backward agree on every quad.
// Sums are fp32 with atomics: the order is not deterministic; tests compare with the fp64 oracle gradient.
#include "cv_common.hpp"

namespace magnet {

namespace {

__device__ __forceinline__ void bwd_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t BKEY_CLOSED = 0xffffffffu;

}  // namespace

// CPL = 16-byte channel chunks per lane (F*4/16 <= 8*CPL).  fp32 features only.
template <int CPL>
__global__ __launch_bounds__(256, 4) void cvf_bwd_kernel(const CvParams p, const float* __restrict__ gout,
                                                         float* __restrict__ grad_ref, float* __restrict__ grad_src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tile, b;
    tile_of_block(p, tile, b);
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int y = ty * TILE_H + wv;
    const int yc = min(y, p.h - 1);
    const int x_base = tx * TILE_W;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const float fw = (float)p.w, fh = (float)p.h;
    const int JB = (p.D + 63) / 64;

    const int wave_bytes = p.V * 512 + 1088 + 272;
    unsigned char* wbase = smem + wv * wave_bytes;
    float4*   pvtab = reinterpret_cast<float4*>(wbase);                          // [V][16 px][2]
    float*    gtab  = reinterpret_cast<float*>(wbase + p.V * 512);               // [65 items][4 taps] sum of g*w
    uint32_t* items = reinterpret_cast<uint32_t*>(wbase + p.V * 512 + 1088);     // [64 + pad]

    for (int e = lane; e < 16 * p.V; e += 64) {
        const int q = e & 15, v = e >> 4;
        const int xc = min(x_base + q, p.w - 1);
        const size_t pix = (size_t)yc * p.w + xc;
        const float r0 = p.rays[((size_t)b * 3 + 0) * hw + pix];
        const float r1 = p.rays[((size_t)b * 3 + 1) * hw + pix];
        const float r2 = p.rays[((size_t)b * 3 + 2) * hw + pix];
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        pvtab[e * 2 + 0] = make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz);
        pvtab[e * 2 + 1] = make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz);
    }
    bwd_lds_fence();

    const GridConst gc = grid_const(p);
    const uint32_t texel_bytes = (uint32_t)p.F * 4u;
    const int nchunk = (int)(texel_bytes / 16);
    const int sub = lane & 7, tap = (lane >> 3) & 3, upair = lane >> 5;
    const uint32_t lane_src_off = (uint32_t)((tap & 1) + (tap >> 1) * Wp) * texel_bytes + (uint32_t)sub * 16u;
    const unsigned char* __restrict__ ref_row = reinterpret_cast<const unsigned char*>(p.ref_feat) +
        ((size_t)b * hw + (size_t)yc * p.w) * texel_bytes;
    const float fV = (float)p.V;
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);

    for (int q = 0; q < 16; ++q) {
        const int x = x_base + q;
        const bool pinb = (x < p.w) && (y < p.h);                                 // wave-uniform
        if (!pinb) continue;
        const size_t pix = (size_t)y * p.w + x;
        float4 rv[CPL], ga[CPL];                                                  // reference chunk(s); grad_ref partials
        {
            const unsigned char* rp = ref_row + (__umul24((uint32_t)x, texel_bytes) + (uint32_t)sub * 16u);
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                rv[cc] = (sub + 8 * cc < nchunk) ? *reinterpret_cast<const float4*>(rp + cc * 128) : make_float4(0.f, 0.f, 0.f, 0.f);
                ga[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int jb = 0; jb < JB; ++jb) {
            const int j = jb * 64 + lane;
            const bool live = j < p.D;
            const int jc = min(j, p.D - 1);
            const float d = live ? p.k[jc] : __builtin_nanf("");
            const float gj = live ? gout[((size_t)b * p.D + jc) * hw + pix] / fV : 0.f;   // d(cost)/d(view sum), homography.py:46

            for (int v = 0; v < p.V; ++v) {
                if (!((vmask >> v) & 1ull)) continue;
                const size_t sidx = (size_t)v * p.B + b;
                const size_t img_off = sidx * (size_t)Hp * Wp * texel_bytes;
                const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(p.src_feat) + img_off;
                unsigned char* __restrict__ gsrc = reinterpret_cast<unsigned char*>(grad_src) + img_off;
                const float4 pa = pvtab[(v * 16 + q) * 2 + 0], pb = pvtab[(v * 16 + q) * 2 + 1];
                PixelView pv;
                pv.rpx = pa.x; pv.rpy = pa.y; pv.rpz = pa.z; pv.rcz = pa.w;
                pv.kt0 = pb.x; pv.kt1 = pb.y; pv.kt2 = pb.z; pv.tz = pb.w;
                float ix, iy, zw;
                project(pv, gc, d, ix, iy, zw);
                int x0, y0; bool inwin;
                const Taps t = make_taps(ix, iy, fw, fh, x0, y0, inwin);
                const uint32_t qi = inwin ? (uint32_t)(__mul24(y0 + 1, Wp) + (x0 + 1)) : 0u;
                const uint32_t key = inwin ? qi : BKEY_CLOSED;
                uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)BKEY_CLOSED, (int)key, 0x138, 0xf, 0xf, false);  // wave_shr:1
                if (lane == 0) prev = BKEY_CLOSED;
                const bool fresh = inwin && (key != prev);
                const unsigned long long bal = __ballot(fresh);
                const int nitems = __popcll(bal);
                if (nitems == 0) continue;
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                const int myitem = below + (fresh ? 1 : 0) - 1;
                if (fresh) items[below] = qi;
                if (lane == 0) items[nitems] = 0u;
                for (int e = lane; e < (nitems + 1) * 4; e += 64) gtab[e] = 0.f;
                bwd_lds_fence();
                if (inwin) {                                                      // sum of g*w per (item, tap)
                    float* cell = gtab + myitem * 4;
                    __hip_atomic_fetch_add(cell + 0, gj * t.nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(cell + 1, gj * t.ne, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(cell + 2, gj * t.sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(cell + 3, gj * t.se, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                bwd_lds_fence();
                const int passes = (nitems + 1) >> 1;                            // 2 items (8 units) per pass
                for (int ps = 0; ps < passes; ++ps) {
                    const int it = min(2 * ps + upair, nitems);                   // odd tail: the pad item (G = 0)
                    const uint32_t texel = items[it];
                    const float G = gtab[it * 4 + tap];
                    const uint32_t off = __umul24(texel, texel_bytes) + lane_src_off;
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) {
                        if (sub + 8 * cc < nchunk) {
                            const float4 s = *reinterpret_cast<const float4*>(src + off + cc * 128);
                            ga[cc].x = __builtin_fmaf(G, s.x, ga[cc].x);
                            ga[cc].y = __builtin_fmaf(G, s.y, ga[cc].y);
                            ga[cc].z = __builtin_fmaf(G, s.z, ga[cc].z);
                            ga[cc].w = __builtin_fmaf(G, s.w, ga[cc].w);
                            if (G != 0.f && grad_src) {                       // grad_src == nullptr: grad_ref only (cost_volume_f_gather.hip owns grad_src)
                                float* gp = reinterpret_cast<float*>(gsrc + off + cc * 128);
                                unsafeAtomicAdd(gp + 0, G * rv[cc].x);
                                unsafeAtomicAdd(gp + 1, G * rv[cc].y);
                                unsafeAtomicAdd(gp + 2, G * rv[cc].z);
                                unsafeAtomicAdd(gp + 3, G * rv[cc].w);
                            }
                        }
                    }
                }
                bwd_lds_fence();                                                  // gtab/items are rewritten by the next view
            }
        }
        // grad_ref: lanes with equal `sub` hold partial sums of the same channels (8 unit slots)
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) {
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) {
                ga[cc].x += __shfl_xor(ga[cc].x, m);
                ga[cc].y += __shfl_xor(ga[cc].y, m);
                ga[cc].z += __shfl_xor(ga[cc].z, m);
                ga[cc].w += __shfl_xor(ga[cc].w, m);
            }
            if (lane < 8 && (sub + 8 * cc < nchunk))
                *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(grad_ref) +
                                           ((size_t)b * hw + pix) * texel_bytes + (size_t)sub * 16 + cc * 128) = ga[cc];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tile-privatised variant (default when F % 32 == 0).  The atomic-per-(item, tap, channel) kernel above is bound by the
// fp32 atomic rate of the L2 (~80-120 G lane-atomics/s, tools/ubench/atomic_scope.hip): every source texel receives
// ~160 contributions per view.  Here a workgroup (16 x 4 reference pixels, all candidates, ONE source view at a time)
// accumulates grad_src in an LDS hash table keyed by source texel (480 slots x 32 channels; ds_add_f32 is ~100x
// faster than a global atomic) and flushes each touched texel ONCE per (tile, view): the tile's candidates sweep a
// narrow epipolar band, so ~10 000 (item, tap) contributions collapse into a few hundred texels.  Channels are processed
// in slices of 32 (the geometry is recomputed per slice) so that the table is 64 KB and two workgroups fit a CU.
// A texel that finds no slot within 8 probes falls back to direct global atomics (correct, just slower).
// grad_ref (no conflicts between workgroups, but one partial per view and slice) is added with global atomics too:
// 8 lanes x 4 per (pixel, view, slice) — noise next to grad_src.
constexpr int HT_SLOTS = 480;          // x 32 channels x 4 B = 60 KB; with the per-wave tables two workgroups fit a CU
constexpr uint32_t HT_EMPTY = 0xffffffffu;

__global__ __launch_bounds__(256, 2) void cvf_bwd_tile_kernel(const CvParams p, const float* __restrict__ gout,
                                                              float* __restrict__ grad_ref, float* __restrict__ grad_src) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tile, b;
    tile_of_block(p, tile, b);
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int y = ty * TILE_H + wv;
    const int yc = min(y, p.h - 1);
    const int x_base = tx * TILE_W;
    const size_t hw = (size_t)p.h * p.w;
    const int Wp = p.w + 2, Hp = p.h + 2;
    const float fw = (float)p.w, fh = (float)p.h;
    const int JB = (p.D + 63) / 64;

    uint32_t* hkeys = reinterpret_cast<uint32_t*>(smem);                         // [HT_SLOTS]
    float*    hval  = reinterpret_cast<float*>(smem + HT_SLOTS * 4);             // [HT_SLOTS][32]
    uint32_t* hcount = reinterpret_cast<uint32_t*>(smem + HT_SLOTS * 4 + HT_SLOTS * 128);   // occupied slots
    const int wave_bytes = p.V * 512 + 1088 + 272 + 1088;
    unsigned char* wbase = smem + HT_SLOTS * 4 + HT_SLOTS * 128 + 16 + wv * wave_bytes;
    float4*   pvtab = reinterpret_cast<float4*>(wbase);                          // [V][16 px][2]
    float*    gtab  = reinterpret_cast<float*>(wbase + p.V * 512);               // [65 items][4 taps] sum of g*w
    uint32_t* items = reinterpret_cast<uint32_t*>(wbase + p.V * 512 + 1088);     // [64 + pad]
    int*      slots = reinterpret_cast<int*>(wbase + p.V * 512 + 1088 + 272);    // [65 items][4 taps] table slot of the unit's texel

    for (int e = tid; e < HT_SLOTS; e += 256) hkeys[e] = HT_EMPTY;
    for (int e = tid; e < HT_SLOTS * 32; e += 256) hval[e] = 0.f;
    if (tid == 0) *hcount = 0u;
    for (int e = lane; e < 16 * p.V; e += 64) {
        const int q = e & 15, v = e >> 4;
        const int xc = min(x_base + q, p.w - 1);
        const size_t pix = (size_t)yc * p.w + xc;
        const float r0 = p.rays[((size_t)b * 3 + 0) * hw + pix];
        const float r1 = p.rays[((size_t)b * 3 + 1) * hw + pix];
        const float r2 = p.rays[((size_t)b * 3 + 2) * hw + pix];
        const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
        pvtab[e * 2 + 0] = make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz);
        pvtab[e * 2 + 1] = make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz);
    }
    __syncthreads();

    const GridConst gc = grid_const(p);
    const uint32_t texel_bytes = (uint32_t)p.F * 4u;
    const int sub = lane & 7, tap = (lane >> 3) & 3, upair = lane >> 5;
    // flush the table (workgroup-uniform call): one global atomic per touched (texel, channel), then empty it
    auto flush = [&](unsigned char* gdst, uint32_t ch_off_) {
        __syncthreads();
        for (int s2 = tid >> 3; s2 < HT_SLOTS; s2 += 32) {
            const uint32_t texel = hkeys[s2];
            if (texel == HT_EMPTY) continue;
            if (p.stats && sub == 0) atomicAdd(p.stats + 1, 1u);                  // flushed texels
            float* cell = hval + s2 * 32 + 4 * sub;
            const float4 val = *reinterpret_cast<const float4*>(cell);
            *reinterpret_cast<float4*>(cell) = make_float4(0.f, 0.f, 0.f, 0.f);
            float* gp = reinterpret_cast<float*>(gdst + (__umul24(texel, texel_bytes) + ch_off_));
            unsafeAtomicAdd(gp + 0, val.x); unsafeAtomicAdd(gp + 1, val.y);
            unsafeAtomicAdd(gp + 2, val.z); unsafeAtomicAdd(gp + 3, val.w);
        }
        __syncthreads();
        for (int e = tid; e < HT_SLOTS; e += 256) hkeys[e] = HT_EMPTY;
        if (tid == 0) *hcount = 0u;
        __syncthreads();
    };
    const uint32_t tap_texels = (uint32_t)((tap & 1) + (tap >> 1) * Wp);
    const float fV = (float)p.V;
    unsigned long long vmask = 0ull;
    for (int v = 0; v < p.V; ++v) vmask |= (unsigned long long)(p.is_valid[b * p.V + v] == 1) << v;
    vmask = __builtin_amdgcn_readfirstlane((uint32_t)vmask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(vmask >> 32)) << 32);
    const bool row_in = y < p.h;                                                  // wave-uniform

    for (int c0 = 0; c0 < p.F; c0 += 32) {                                        // 32-channel slice; lane: channels c0 + 4*sub .. +3
        const uint32_t ch_off = (uint32_t)(c0 + 4 * sub) * 4u;
        for (int v = 0; v < p.V; ++v) {
            if (!((vmask >> v) & 1ull)) continue;                                 // workgroup-uniform
            const size_t sidx = (size_t)v * p.B + b;
            const size_t img_off = sidx * (size_t)Hp * Wp * texel_bytes;
            const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(p.src_feat) + img_off;
            unsigned char* __restrict__ gsrc = reinterpret_cast<unsigned char*>(grad_src) + img_off;
            for (int q = 0; q < 16; ++q) {
                const int x = x_base + q;
                // the table is flushed early once it is ~2/3 full (checked between pixels, all four waves together), so that
                // long epipolar runs keep merging in LDS instead of spilling to direct global atomics
                if (__syncthreads_or(tid == 0 && *hcount > (uint32_t)(HT_SLOTS * 5 / 8))) flush(gsrc, ch_off);
                if (!row_in || x >= p.w) continue;                                // wave-uniform
                const size_t pix = (size_t)y * p.w + x;
                const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(p.ref_feat) +
                                                                   ((size_t)b * hw + pix) * texel_bytes + ch_off);
                float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int jb = 0; jb < JB; ++jb) {
                    const int j = jb * 64 + lane;
                    const bool live = j < p.D;
                    const int jc = min(j, p.D - 1);
                    const float d = live ? p.k[jc] : __builtin_nanf("");
                    const float gj = live ? gout[((size_t)b * p.D + jc) * hw + pix] / fV : 0.f;
                    const float4 pa = pvtab[(v * 16 + q) * 2 + 0], pb = pvtab[(v * 16 + q) * 2 + 1];
                    PixelView pv;
                    pv.rpx = pa.x; pv.rpy = pa.y; pv.rpz = pa.z; pv.rcz = pa.w;
                    pv.kt0 = pb.x; pv.kt1 = pb.y; pv.kt2 = pb.z; pv.tz = pb.w;
                    float ix, iy, zw;
                    project(pv, gc, d, ix, iy, zw);
                    int x0, y0; bool inwin;
                    const Taps t = make_taps(ix, iy, fw, fh, x0, y0, inwin);
                    const uint32_t qi = inwin ? (uint32_t)(__mul24(y0 + 1, Wp) + (x0 + 1)) : 0u;
                    const uint32_t key = inwin ? qi : BKEY_CLOSED;
                    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)BKEY_CLOSED, (int)key, 0x138, 0xf, 0xf, false);
                    if (lane == 0) prev = BKEY_CLOSED;
                    const bool fresh = inwin && (key != prev);
                    const unsigned long long bal = __ballot(fresh);
                    const int nitems = __popcll(bal);
                    if (nitems == 0) continue;
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    const int myitem = below + (fresh ? 1 : 0) - 1;
                    if (fresh) items[below] = qi;
                    if (lane == 0) items[nitems] = 0u;
                    for (int e = lane; e < (nitems + 1) * 4; e += 64) gtab[e] = 0.f;
                    bwd_lds_fence();
                    if (inwin) {
                        float* cell = gtab + myitem * 4;
                        __hip_atomic_fetch_add(cell + 0, gj * t.nw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        __hip_atomic_fetch_add(cell + 1, gj * t.ne, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        __hip_atomic_fetch_add(cell + 2, gj * t.sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        __hip_atomic_fetch_add(cell + 3, gj * t.se, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    bwd_lds_fence();
                    // table slot of every (item, tap) unit's texel, one unit per lane (the CAS latency is paid once per 64 units,
                    // not once per pass); -1: nothing to add, -2: no slot within 8 probes -> direct global atomics
                    for (int u = lane; u < (nitems + 1) * 4; u += 64) {
                        int slot = -1;
                        if (u < nitems * 4 && gtab[u] != 0.f) {
                            const uint32_t texel = items[u >> 2] + (uint32_t)((u & 1) + ((u >> 1) & 1) * Wp);
                            uint32_t sidx2 = ((texel * 2654435761u) >> 16) % (uint32_t)HT_SLOTS;
                            slot = -2;
                            for (int pr = 0; pr < 8; ++pr) {
                                const uint32_t old = atomicCAS(hkeys + sidx2, HT_EMPTY, texel);
                                if (old == HT_EMPTY) atomicAdd(hcount, 1u);
                                if (old == HT_EMPTY || old == texel) { slot = (int)sidx2; break; }
                                sidx2 = (sidx2 + 1 == (uint32_t)HT_SLOTS) ? 0u : sidx2 + 1;
                            }
                            if (p.stats) atomicAdd(p.stats + (slot >= 0 ? 2 : 3), 1u);    // units merged in LDS / sent to global atomics
                        }
                        slots[u] = slot;
                    }
                    bwd_lds_fence();
                    const int passes = (nitems + 1) >> 1;
                    for (int ps = 0; ps < passes; ++ps) {
                        const int it = min(2 * ps + upair, nitems);
                        const uint32_t texel = items[it] + tap_texels;            // this unit's source texel
                        const float G = gtab[it * 4 + tap];
                        const int slot = slots[it * 4 + tap];
                        const float4 sv = *reinterpret_cast<const float4*>(src + (__umul24(texel, texel_bytes) + ch_off));
                        ga.x = __builtin_fmaf(G, sv.x, ga.x); ga.y = __builtin_fmaf(G, sv.y, ga.y);
                        ga.z = __builtin_fmaf(G, sv.z, ga.z); ga.w = __builtin_fmaf(G, sv.w, ga.w);
                        if (slot >= 0) {
                            // a slot is 32 floats = half of the 64 LDS banks and a unit's 8 lanes are 4 floats apart: issuing
                            // component (i + unit) & 3 in the i-th atomic makes the 8 units of a pass cover all bank residues
                            float* cell = hval + slot * 32 + 4 * sub;
                            const float r[4] = {G * rv.x, G * rv.y, G * rv.z, G * rv.w};
                            const int u0 = lane >> 3;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int c = (i + u0) & 3;
                                const float val = c == 0 ? r[0] : (c == 1 ? r[1] : (c == 2 ? r[2] : r[3]));
                                __hip_atomic_fetch_add(cell + c, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        } else if (slot == -2) {
                            float* gp = reinterpret_cast<float*>(gsrc + (__umul24(texel, texel_bytes) + ch_off));
                            unsafeAtomicAdd(gp + 0, G * rv.x); unsafeAtomicAdd(gp + 1, G * rv.y);
                            unsafeAtomicAdd(gp + 2, G * rv.z); unsafeAtomicAdd(gp + 3, G * rv.w);
                        }
                    }
                    bwd_lds_fence();
                }
                // grad_ref partial of (pixel, view, slice): lanes with equal `sub` hold the same channels
#pragma unroll
                for (int m = 8; m < 64; m <<= 1) {
                    ga.x += __shfl_xor(ga.x, m); ga.y += __shfl_xor(ga.y, m);
                    ga.z += __shfl_xor(ga.z, m); ga.w += __shfl_xor(ga.w, m);
                }
                if (lane < 8) {
                    float* gr = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(grad_ref) + ((size_t)b * hw + pix) * texel_bytes + ch_off);
                    unsafeAtomicAdd(gr + 0, ga.x); unsafeAtomicAdd(gr + 1, ga.y);
                    unsafeAtomicAdd(gr + 2, ga.z); unsafeAtomicAdd(gr + 3, ga.w);
                }
            }
            flush(gsrc, ch_off);                                                  // end of this (tile, view, slice)
        }
    }
}

// grad_ref only (fully written, no atomics): the per-item kernel with its grad_src scatter switched off
hipError_t launch_cvf_bwd_ref_only(const CvParams& p, const float* gout, float* grad_ref, hipStream_t stream, bool* handled) {
    *handled = false;
    if (p.feat_bf16) return hipSuccess;
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 24)) return hipSuccess;
    if ((size_t)(p.h + 2) * (p.w + 2) * p.F * 4 >= ((size_t)1 << 32)) return hipSuccess;
    const size_t lds = (size_t)4 * (p.V * 512 + 1088 + 272);
    if (lds > 64 * 1024) return hipSuccess;
    const int nchunk = p.F * 4 / 16;
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    *handled = true;
    if (nchunk <= 8)       hipLaunchKernelGGL((cvf_bwd_kernel<1>), grid, block, lds, stream, p, gout, grad_ref, (float*)nullptr);
    else if (nchunk <= 16) hipLaunchKernelGGL((cvf_bwd_kernel<2>), grid, block, lds, stream, p, gout, grad_ref, (float*)nullptr);
    else if (nchunk <= 32) hipLaunchKernelGGL((cvf_bwd_kernel<4>), grid, block, lds, stream, p, gout, grad_ref, (float*)nullptr);
    else { *handled = false; return hipSuccess; }
    return hipGetLastError();
}

hipError_t launch_cvf_bwd(const CvParams& p, const float* gout, float* grad_ref, float* grad_src, hipStream_t stream,
                          bool* handled) {
    *handled = false;
    if (p.feat_bf16) return hipSuccess;
    if ((size_t)(p.h + 2) * (p.w + 2) >= ((size_t)1 << 24)) return hipSuccess;
    if ((size_t)(p.h + 2) * (p.w + 2) * p.F * 4 >= ((size_t)1 << 32)) return hipSuccess;
    const size_t lds = (size_t)4 * (p.V * 512 + 1088 + 272);
    if (lds > 64 * 1024) return hipSuccess;
    const int nchunk = p.F * 4 / 16;
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    *handled = true;
    const size_t lds_tile = (size_t)HT_SLOTS * 4 + (size_t)HT_SLOTS * 128 + 16 + (size_t)4 * (p.V * 512 + 1088 + 272 + 1088);
    if ((p.F % 32) == 0 && lds_tile <= 80 * 1024 && !(CV_DEV(p) & 32)) {        // dev bit 32: the per-item atomic kernel
        hipError_t e0 = hipMemsetAsync(grad_ref, 0, (size_t)p.B * p.h * p.w * p.F * sizeof(float), stream);   // accumulated into
        if (e0 != hipSuccess) return e0;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cvf_bwd_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(cvf_bwd_tile_kernel, grid, block, lds_tile, stream, p, gout, grad_ref, grad_src);
        return hipGetLastError();
    }
    if (nchunk <= 8)       hipLaunchKernelGGL((cvf_bwd_kernel<1>), grid, block, lds, stream, p, gout, grad_ref, grad_src);
    else if (nchunk <= 16) hipLaunchKernelGGL((cvf_bwd_kernel<2>), grid, block, lds, stream, p, gout, grad_ref, grad_src);
    else if (nchunk <= 32) hipLaunchKernelGGL((cvf_bwd_kernel<4>), grid, block, lds, stream, p, gout, grad_ref, grad_src);
    else { *handled = false; return hipSuccess; }
    return hipGetLastError();
}

}  // namespace magnet
