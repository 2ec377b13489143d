// cost_volume_f_gather.hip — backward of est_costvolume_F w.r.t. the SOURCE features as a GATHER: no atomics, deterministic.
//
// Reference: est_costvolume_F / _compute_cost_F (models/submodules/homography.py:10-75), differentiated by autograd while the
// F-Net is trained (train_FNet.py:95-96):   grad_src_v[f, texel] = sum_{(j,p,tap) -> texel}  g[j,p]/V * w_tap * ref[f,p].
//
// The scatter kernels (cost_volume_f_bwd.hip) are bound by atomics: ~1.5e9 global fp32 atomics per 16-frame launch even after an
// LDS hash table merges them (the L2 retires ~1e11 lane-atomics/s) plus ~8e9 ds_add_f32 at one lane per clock.  est_costvolume_F's
// depth bins are the same for every pixel, so for a fixed (view, bin) the map reference pixel -> source position is ONE
// homography: a 16x4 tile of reference pixels lands in a compact source window.  That makes ownership by SOURCE tile cheap:
//   1. cvf_prep_kernel: per (frame, view, reference tile): the depth-linear projection terms of its 64 pixels (so that nobody
//      recomputes K R ray) and, per bin, the bounding box of the tile's sample quads in the source map — exact forward
//      arithmetic (warp_math.hpp), 8 bytes per (frame, view, bin, tile);
//   2. cvf_gather_src_kernel: one WAVE per (frame, view, source row segment of 32 texels).  For every bin the wave scans the
//      bounding boxes, re-projects the pixels of every reference tile whose box touches the segment (lane = pixel, the forward's
//      arithmetic bit for bit, so forward and backward agree on every quad), compacts the (pixel, tap) pairs that land inside
//      the segment into a list, and applies them with lane = CHANNEL: acc[texel][lane] += coef * ref[pixel][lane] as plain LDS
//      read-modify-write (a wave's LDS operations execute in order; the accumulators are wave-private).  Every texel is
//      stored ONCE with a plain 256-byte store: fixed summation order, bit-identical run to run.
//   grad_ref comes from cvf_bwd_kernel with its grad_src atomics switched off (cost_volume_f_bwd.hip).
// Workspace (caller-provided): 32 B per (frame, view, pixel) + 8 B per (frame, view, bin, reference tile).
#include "cv_common.hpp"

namespace magnet {

namespace {

__device__ __forceinline__ void gat_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct __attribute__((aligned(8))) TileBox { short xmin, xmax, ymin, ymax; };      // quad origins, padded-map texel coordinates

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m));
    return v;
}

}  // namespace

// grid = B * V * (reference tiles); 64 threads: lane = pixel of the 16x4 reference tile
__global__ __launch_bounds__(64) void cvf_prep_kernel(const CvParams p, float4* __restrict__ pvt, TileBox* __restrict__ boxes) {
    const int lane = threadIdx.x;
    const int ntiles = p.tiles_x * p.tiles_y;
    const int tile = blockIdx.x % ntiles, bv = blockIdx.x / ntiles;
    const int v = bv % p.V, b = bv / p.V;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x = tx * TILE_W + (lane & 15), y = ty * TILE_H + (lane >> 4);
    const bool pin = x < p.w && y < p.h;
    const size_t hw = (size_t)p.h * p.w;
    const int xc = min(x, p.w - 1), yc = min(y, p.h - 1);
    float r0, r1, r2;
    load_ray(p, b, hw, xc, yc, r0, r1, r2);
    const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9, p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
    if (pin) {
        const size_t e = ((size_t)bv * hw + (size_t)y * p.w + x) * 2;
        pvt[e + 0] = make_float4(pv.rpx, pv.rpy, pv.rpz, pv.rcz);
        pvt[e + 1] = make_float4(pv.kt0, pv.kt1, pv.kt2, pv.tz);
    }
    const bool valid = p.is_valid[b * p.V + v] == 1;
    const GridConst gc = grid_const(p);
    const float fw = (float)p.w, fh = (float)p.h;
    for (int j = 0; j < p.D; ++j) {
        float ix, iy, zw;
        project(pv, gc, p.k[j], ix, iy, zw);
        int x0, y0; bool inwin;
        (void)make_taps(ix, iy, fw, fh, x0, y0, inwin);
        const bool ok = pin && inwin && valid;
        const int qx = x0 + 1, qy = y0 + 1;
        const int xmn = wave_min(ok ? qx : 32767), xmx = wave_max(ok ? qx : -1);
        const int ymn = wave_min(ok ? qy : 32767), ymx = wave_max(ok ? qy : -1);
        if (lane == 0) {
            TileBox tb; tb.xmin = (short)xmn; tb.xmax = (short)xmx; tb.ymin = (short)ymn; tb.ymax = (short)ymx;
            boxes[((size_t)bv * p.D + j) * ntiles + tile] = tb;
        }
    }
}

constexpr int GS_LIST = 64;                              // pixels of one reference tile whose quad touches a segment: <= 64

// One WAVE per (frame, view, source row segment of SW texels): fully independent waves (no workgroup barrier, no shared state), lane =
// pixel while re-projecting, lane = channel while accumulating.  LDS per wave: (SW + 2) x 256 B of accumulators + the 1 KB pixel list,
// so 28 waves (SW = 16) fit a CU.  Measured history on 16 ScanNet-shape frames (fwd + bwd): LDS hash scatter with atomics 53.5 ms;
// workgroup per 16x4 source tile with one 64-texel accumulator copy per wave (16 KB each: 8 waves per CU, bound by the latency of
// the chain list entry -> reference row load -> LDS read-modify-write) 27.7 ms; the same tile shared and updated with ds_add_f32
// 105 ms (LDS float atomics retire about one lane per clock); wave per row segment 21.6 ms; one list entry per PIXEL (its two
// taps on this row share the reference row load) 17.2 ms.
// grid = (ceil(B * V * h * segments / 4), channel blocks of 64); 256 threads = 4 independent waves
template <int SW>
__global__ __launch_bounds__(256) void cvf_gather_src_kernel(const CvParams p, const float* __restrict__ gout, float* __restrict__ grad_src,
                                                             const float4* __restrict__ pvt, const TileBox* __restrict__ boxes, int segs_x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WAVE_LDS = (SW + 2) * 256 + GS_LIST * 16;
    float* acc = reinterpret_cast<float*>(smem + wv * WAVE_LDS);                              // [guard, SW texels, guard][64 channels]
    uint4* list = reinterpret_cast<uint4*>(smem + wv * WAVE_LDS + (SW + 2) * 256);            // {left slot | pixel << 8, coefficients}
    const long long unit = (long long)blockIdx.x * 4 + wv;                                    // (frame, view, row, segment)
    const long long nunits = (long long)p.B * p.V * p.h * segs_x;
    if (unit >= nunits) return;                           // wave-uniform; nothing below synchronises across waves
    const int seg = (int)(unit % segs_x);
    const int row = (int)((unit / segs_x) % p.h);
    const int bv = (int)(unit / ((long long)segs_x * p.h));
    const int v = bv % p.V, b = bv / p.V;
    const int cb = blockIdx.y * 64;                       // first channel of this block; lane = channel cb + lane
    const bool chan = cb + lane < p.F;
    const int SX = 1 + seg * SW, SY = 1 + row;            // padded-map coordinates of the segment
    const int Wp = p.w + 2, Hp = p.h + 2;
    const size_t hw = (size_t)p.h * p.w;
    const int ntiles = p.tiles_x * p.tiles_y;
    for (int e = lane; e < (SW + 2) * 64; e += 64) acc[e] = 0.f;
    gat_lds_fence();
    const bool valid = p.is_valid[b * p.V + v] == 1;      // homography.py:26 (wave-uniform)
    if (valid) {
        const GridConst gc = grid_const(p);
        const float fw = (float)p.w, fh = (float)p.h, fV = (float)p.V;
        const float* __restrict__ refc = reinterpret_cast<const float*>(p.ref_feat) + (size_t)b * hw * p.F + cb + lane;
        for (int j = 0; j < p.D; ++j) {
            const float d = p.k[j];
            const TileBox* __restrict__ bb = boxes + ((size_t)bv * p.D + j) * ntiles;
            const float* __restrict__ gj = gout + ((size_t)b * p.D + j) * hw;
            for (int t0 = 0; t0 < ntiles; t0 += 64) {
                const int t = t0 + lane;
                bool hit = false;
                if (t < ntiles) {
                    const TileBox e = bb[t];
                    hit = (e.xmin <= SX + SW - 1) && (e.xmax + 1 >= SX) && (e.ymin <= SY) && (e.ymax + 1 >= SY);
                }
                unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                while (mask) {                                                                // reference tiles touching this segment
                    const int T = t0 + (int)__builtin_ctzll(mask);
                    mask &= mask - 1;
                    // ---- re-project the tile's 64 pixels (forward arithmetic, bit for bit) ----
                    const int x = (T % p.tiles_x) * TILE_W + (lane & 15), y = (T / p.tiles_x) * TILE_H + (lane >> 4);
                    const bool pin = x < p.w && y < p.h;
                    const uint32_t pix = (uint32_t)(min(y, p.h - 1) * p.w + min(x, p.w - 1));
                    const float4 pa = pvt[((size_t)bv * hw + pix) * 2 + 0], pb = pvt[((size_t)bv * hw + pix) * 2 + 1];
                    PixelView pv;
                    pv.rpx = pa.x; pv.rpy = pa.y; pv.rpz = pa.z; pv.rcz = pa.w;
                    pv.kt0 = pb.x; pv.kt1 = pb.y; pv.kt2 = pb.z; pv.tz = pb.w;
                    float ix, iy, zw;
                    project(pv, gc, d, ix, iy, zw);
                    int x0, y0; bool inwin;
                    const Taps tw = make_taps(ix, iy, fw, fh, x0, y0, inwin);
                    const float G = (pin && inwin) ? gj[pix] / fV : 0.f;                      // d(cost)/d(view sum), homography.py:46
                    // a pixel's quad touches this row with its top taps (nw, ne) or its bottom taps (sw, se), never both: one list
                    // entry per pixel = {left slot | pixel << 8, left coefficient, right coefficient}; slots are shifted by one
                    // guard texel on either side so that a quad straddling the segment's end needs no test
                    const int lx0 = x0 + 1 - SX, ly0 = y0 + 1 - SY;
                    const bool top = ly0 == 0;
                    const bool ok = (G != 0.f) && (top || ly0 == -1) && ((unsigned)(lx0 + 1) <= (unsigned)SW);
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(ok);
                    const int n = __popcll(bal);
                    if (ok) {
                        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        list[pos] = make_uint4((uint32_t)(lx0 + 1) | (pix << 8), __float_as_uint(G * (top ? tw.nw : tw.sw)),
                                               __float_as_uint(G * (top ? tw.ne : tw.se)), 0u);
                    }
                    gat_lds_fence();
                    // ---- apply: lane = channel, plain read-modify-write (in-order LDS, wave-private accumulators) ----
                    constexpr int NB = 8;
                    for (int k0 = 0; k0 < n; k0 += NB) {
                        uint4 e[NB]; float rv[NB];
#pragma unroll
                        for (int i = 0; i < NB; ++i) e[i] = list[min(k0 + i, n - 1)];         // wave-uniform address: broadcast read
#pragma unroll
                        for (int i = 0; i < NB; ++i) {
                            const uint32_t pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[i].x);
                            rv[i] = chan ? refc[(size_t)(pk >> 8) * p.F] : 0.f;               // 256 contiguous bytes per pixel
                            e[i].x = pk;
                        }
#pragma unroll
                        for (int i = 0; i < NB; ++i) {
                            if (k0 + i < n) {                                                 // wave-uniform
                                float* cell = acc + (e[i].x & 255u) * 64 + lane;
                                cell[0] = __builtin_fmaf(__uint_as_float(e[i].y), rv[i], cell[0]);
                                cell[64] = __builtin_fmaf(__uint_as_float(e[i].z), rv[i], cell[64]);
                            }
                        }
                    }
                    gat_lds_fence();                                                          // the list is rewritten by the next tile
                }
            }
        }
    }
    // ---- one plain 256-byte store per texel ----
    float* __restrict__ gdst = grad_src + (size_t)((size_t)v * p.B + b) * Hp * Wp * p.F + ((size_t)SY * Wp) * p.F;
    if (chan)
        for (int tx = 0; tx < SW; ++tx)
            if (SX + tx <= p.w) gdst[(size_t)(SX + tx) * p.F + cb + lane] = acc[(tx + 1) * 64 + lane];
}

size_t cvf_gather_workspace_bytes(const CvParams& p) {
    const size_t ntiles = (size_t)p.tiles_x * p.tiles_y;
    return (size_t)p.B * p.V * p.h * p.w * 32 + (size_t)p.B * p.V * p.D * ntiles * sizeof(TileBox) + 256;
}

// grad_src (interior texels of every valid AND invalid view are written; the one-texel border is left untouched)
hipError_t launch_cvf_gather_src(const CvParams& p, const float* gout, float* grad_src, void* workspace, size_t ws_bytes, hipStream_t stream,
                                 bool* handled) {
    *handled = false;
    if (p.feat_bf16 || !workspace || ws_bytes < cvf_gather_workspace_bytes(p)) return hipSuccess;
    if ((size_t)p.h * p.w >= ((size_t)1 << 24) || p.w + 2 > 32000 || p.h + 2 > 32000) return hipSuccess;   // 24-bit pixel index, 16-bit boxes
    const size_t ntiles = (size_t)p.tiles_x * p.tiles_y;
    float4* pvt = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(workspace) + 15) & ~(uintptr_t)15);
    TileBox* boxes = reinterpret_cast<TileBox*>(pvt + (size_t)p.B * p.V * p.h * p.w * 2);
    *handled = true;
    hipLaunchKernelGGL(cvf_prep_kernel, dim3((unsigned)((size_t)p.B * p.V * ntiles)), dim3(64), 0, stream, p, pvt, boxes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // 16-texel segments: 5.5 KB of LDS per wave, 28 waves per CU.  dev (path bit 14): 32-texel segments (fewer duplicated tile
    // re-projections, 16 waves per CU): 12 % slower on both training shapes; 8-texel segments were 3 - 13 % slower.
    const int SWv = (CV_DEV(p) & 0x40) ? 32 : 16;
    const int segs_x = (p.w + SWv - 1) / SWv;
    const long long nunits = (long long)p.B * p.V * p.h * segs_x;
    const dim3 grid((unsigned)((nunits + 3) / 4), (unsigned)((p.F + 63) / 64));
    const size_t lds = (size_t)4 * ((SWv + 2) * 256 + GS_LIST * 16);
    if (SWv == 32)      hipLaunchKernelGGL(cvf_gather_src_kernel<32>, grid, dim3(256), lds, stream, p, gout, grad_src, pvt, boxes, segs_x);
    else                hipLaunchKernelGGL(cvf_gather_src_kernel<16>, grid, dim3(256), lds, stream, p, gout, grad_src, pvt, boxes, segs_x);
    return hipGetLastError();
}

}  // namespace magnet
