// cost_volume.hip — fused candidate sampling + plane-sweep warp + consistency-weighted matching
// score for gfx950 (replaces models/MAGNET.py:153-156 + models/submodules/homography.py:79-161).
//
// Work decomposition (both kernels): one workgroup = one 16x4 tile of reference pixels of one
// reference frame; 256 threads = 64 pixels x 4 candidate slices (wave w owns candidates
// [w*DPT, (w+1)*DPT)), so every wave's 64 lanes are 64 neighbouring pixels: coalesced (mu,sigma),
// ray and cost-volume accesses, and wave-uniform control flow over (view, candidate).
//
//   cv_generic_kernel : for each (pixel, candidate, view) gathers the 4 bilinear taps x F channels
//                       straight from the channel-last feature maps (L1/L2 served).  Arithmetic is
//                       the oracle's to the bit (per-channel fused bilerp, ATen cascade sum).  It is
//                       the bit-exact reference path and the fallback for shapes the worklist kernel does not take.
//   cv_worklist_kernel: the fast path, see cost_volume_worklist.hip
#include "cv_common.hpp"

namespace magnet {

template <typename FeatT> struct FeatChunk;            // one 16-byte load of consecutive channels
template <> struct FeatChunk<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    __device__ __forceinline__ void zero() { v[0] = v[1] = v[2] = v[3] = 0.f; }
};
template <> struct FeatChunk<uint16_t> {
    static constexpr int N = 8;
    float v[8];
    __device__ __forceinline__ void load(const uint16_t* p) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i]     = __uint_as_float(u[i] << 16);
            v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
};

template <typename FeatT>
__global__ __launch_bounds__(256) void cv_generic_kernel(const CvParams p) {
    const int DPT = (p.D + 3) >> 2;                              // candidates per wave (slice)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int slice = tid >> 6;
    int tile, b;
    tile_of_block(p, tile, b);
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x = tx * TILE_W + (lane & (TILE_W - 1));
    const int y = ty * TILE_H + (lane / TILE_W);
    const bool inb = (x < p.w) && (y < p.h);
    const int xc = inb ? x : 0, yc = inb ? y : 0;          // clamp so every lane reads valid memory
    const size_t hw = (size_t)p.h * p.w;
    const size_t pix = (size_t)yc * p.w + xc;

    float r0, r1, r2;

    load_ray(p, b, hw, xc, yc, r0, r1, r2);
    float mu = 0.f, sg = 0.f;
    if (!p.d_volume && !p.mode_f) {
        mu = p.ref_gmm[((size_t)b * 2 + 0) * hw + pix];
        sg = p.ref_gmm[((size_t)b * 2 + 1) * hw + pix];
    }
    const GridConst gc = grid_const(p);
    const int Wp = p.w + 2, Hp = p.h + 2;
    const FeatT* __restrict__ ref = reinterpret_cast<const FeatT*>(p.ref_feat) + ((size_t)b * hw + pix) * p.F;

    const float fV = (float)p.V;
    // candidates outer, views inner: the fp64 view accumulator (homography.py:116,159) is a scalar
#pragma unroll 1
    for (int i = 0; i < DPT; ++i) {
        const int j = slice * DPT + i;
        if (j >= p.D) break;                                       // wave-uniform
        float d;
        if (p.d_volume) d = p.d_volume[((size_t)b * p.D + j) * hw + pix];
        else if (p.mode_f) d = p.k[j];                             // est_costvolume_F: fixed depth bins
        else { const float sk = sg * p.k[j]; d = mu + sk; }        // MAGNET.py:155 (mul, then add)
        double acc = 0.0;
        float accf = 0.f;                                          // mode 1: fp32 view sum (homography.py:42)
#pragma unroll 1
        for (int v = 0; v < p.V; ++v) {
            if (p.is_valid[b * p.V + v] != 1) continue;            // homography.py:97 (wave-uniform)
            const PixelView pv = make_pixel_view(p.intM + (size_t)b * 9,
                                                 p.poses + ((size_t)b * p.V + v) * 16, r0, r1, r2);
            const size_t sidx = (size_t)v * p.B + b;               // view-major, homography.py:105
            const FeatT* __restrict__ src = reinterpret_cast<const FeatT*>(p.src_feat) + sidx * (size_t)Hp * Wp * p.F;
            const float* __restrict__ sgm = p.src_gmm + sidx * (size_t)Hp * Wp * 2;
            float ix, iy, zw;
            project(pv, gc, d, ix, iy, zw);
            int qx0, qy0; bool inwin;
            const Taps t = make_taps(ix, iy, (float)p.w, (float)p.h, qx0, qy0, inwin);
            // quads with at least one tap inside the image live entirely inside the zero-bordered
            // (h+2)x(w+2) source maps, so no per-tap bounds checks are needed (zeros padding,
            // homography.py:150-152); everything else contributes exactly 0 with a closed gate.
            float c = 0.f, mu_w = 0.f, sg_w = 0.f;
            if (inwin) {
                const size_t o_nw = (size_t)(qy0 + 1) * Wp + (qx0 + 1);
                const size_t o_ne = o_nw + 1, o_sw = o_nw + Wp, o_se = o_nw + Wp + 1;
                float lvl0 = 0.f, lvl1 = 0.f, lvl2 = 0.f;          // ATen cascade sum (homography.py:155)
                for (int f0 = 0; f0 < p.F; f0 += FeatChunk<FeatT>::N) {
                    FeatChunk<FeatT> cr, ca, cb, cc, cd;
                    cr.load(ref + f0);
                    ca.load(src + o_nw * p.F + f0);
                    cb.load(src + o_ne * p.F + f0);
                    cc.load(src + o_sw * p.F + f0);
                    cd.load(src + o_se * p.F + f0);
#pragma unroll
                    for (int q = 0; q < FeatChunk<FeatT>::N; ++q) {
                        const float wv = bilerp(ca.v[q], cb.v[q], cc.v[q], cd.v[q], t);
                        const float pr = cr.v[q] * wv;
                        lvl0 = lvl0 + pr;
                        const int f = f0 + q;
                        if ((f & 15) == 15) {
                            lvl1 = lvl1 + lvl0; lvl0 = 0.f;
                            if ((f & 255) == 255) { lvl2 = lvl2 + lvl1; lvl1 = 0.f; }
                        }
                    }
                }
                c = (lvl0 + lvl1) + lvl2;
                if (!p.mode_f) {
                    mu_w = bilerp(sgm[o_nw * 2], sgm[o_ne * 2], sgm[o_sw * 2], sgm[o_se * 2], t);
                    sg_w = bilerp(sgm[o_nw * 2 + 1], sgm[o_ne * 2 + 1], sgm[o_sw * 2 + 1], sgm[o_se * 2 + 1], t);
                }
            }
            const bool gate = __builtin_fabsf(zw - mu_w) < sg_w * p.kappa;   // homography.py:157-158
            if (p.mode_f) accf = accf + c;                                   // no gate, fp32 sum (homography.py:42)
            else acc += (double)c * (gate ? 1.0 : 0.0);                      // fp64 view sum, :159,116
        }
        if (inb) p.cost[(size_t)b * p.cost_bstride + (size_t)j * hw + pix] = (p.mode_f ? accf : (float)acc) / fV;   // :46 / :118,120
    }
    if (p.stats && tid == 0) atomicAdd(p.stats + 1, 1u);
}

template <typename FeatT>
static hipError_t launch_generic_t(const CvParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)((size_t)p.tiles_x * p.tiles_y * p.B)), block(256);
    hipLaunchKernelGGL((cv_generic_kernel<FeatT>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_cv_generic(const CvParams& p, hipStream_t stream) {
    return p.feat_bf16 ? launch_generic_t<uint16_t>(p, stream) : launch_generic_t<float>(p, stream);
}

}  // namespace magnet
