// cv_common.hpp — launch parameters and block->tile mapping shared by the cost-volume kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/magnet_hip.h"
#include "warp_math.hpp"

namespace magnet {

constexpr int TILE_W = 16;      // reference-pixel tile of one workgroup: 16 x 4 = 64 pixels = 1 wave
constexpr int TILE_H = 4;
constexpr int NUM_XCD = 8;      // MI355X: 8 XCDs, block i is dispatched to XCD i % 8

struct CvParams {
    int B, V, F, D, h, w;
    int tiles_x, tiles_y;
    int feat_bf16;
    int mode_f;                       // 1 = est_costvolume_F semantics (fixed bins, no gate, fp32 view sum)
    int ablate;                       // MagnetCostVolumeArgs.dev_flags, dev library only — read through CV_DEV(p) below, never directly
    float kappa;
    const void*    ref_feat;
    const void*    src_feat;          // (V*B, h+2, w+2, F) channel-last, one-texel zero border
    const float*   src_gmm;           // (V*B, h+2, w+2, 2) interleaved [mu,sigma], one-texel zero border
    const float*   src_gmq;           // optional (V*B, h+2, w+2, 8): the same map per QUAD origin in quad form (magnet_pack_gmm_quad)
    const float*   ref_gmm;
    const float*   d_volume;
    const float*   poses;
    const int32_t* is_valid;
    const float*   intM;
    const float*   rays;
    float*         cost;
    uint32_t*      stats;
    long long      cost_bstride;      // elements between frames of `cost`
    uint16_t*      cost_hi;           // optional split-bf16 channel-last output (see include/magnet_hip.h)
    uint16_t*      cost_lo;
    long long      cost_ld;
    uint8_t*       gate_bits;         // optional debug output (B,V,D,h,w)
    int            npx;               // cost_volume_fast64.hip: pixels per wave (set by its launcher)
    int            strip_tx;          // cost_volume_fast64.hip / cost_volume_v3.hip: > 0 = blocks walk the frame in vertical strips of this many tiles instead of raster order (v3: 0 or 1)
    uint32_t       magic_tiles, magic_tiles_x, magic_tiles_y;   // cost_volume_v3.hip: ceil(2^32 / (tiles_x * tiles_y)), ceil(2^32 / tiles_x), ceil(2^32 / tiles_y) (set by its launcher)
    const double*  ray_params;        // optional (B,8) fx, fy, cx, cy, sx, sy, left, top: rays generated in the kernel
    float k[MAGNET_MAX_CANDIDATES];   // (float)k_j, read with wave-uniform indices (scalar loads)
};

// Development switches (timing ablations, kernel variants for same-box A/B: tools/README.md).  The product library is compiled without
// MAGNET_DEV: every CV_DEV(p) is the constant 0 there and the branches it guards are removed by the compiler.
#ifdef MAGNET_DEV
#define CV_DEV(p) ((p).ablate)
#else
#define CV_DEV(p) 0
#endif

__device__ __forceinline__ GridConst grid_const(const CvParams& p) {
    GridConst gc;
    gc.cw = (float)((double)p.w / 2.0);
    gc.ch = (float)((double)p.h / 2.0);
    gc.rcw = 1.0f / gc.cw;
    gc.rch = 1.0f / gc.ch;
    gc.sw = (float)p.w / 2.0f;
    gc.sh = (float)p.h / 2.0f;
    return gc;
}

// unit ray of grid pixel (x, y) of frame b: the loader's table entry (dataloader_scannet.py:139-147, dataloader_kitti.py:113-118),
// or — when the caller passed the 8 scalars instead of the 12*h*w-byte table — the same float64 expression evaluated here:
// ((x + 0.5) * sx - cx + left) / fx, cast to fp32 once.  IEEE double mul / add / div (no contraction: -ffp-contract=off), so
// the result equals the table bit for bit (tests/test_gpu_rays_poses.py).
__device__ __forceinline__ void load_ray(const CvParams& p, int b, size_t hw, int x, int y, float& r0, float& r1, float& r2) {
    if (p.ray_params) {
        const double* q = p.ray_params + (size_t)b * 8;
        r0 = (float)(((((double)x + 0.5) * q[4]) - q[2] + q[6]) / q[0]);
        r1 = (float)(((((double)y + 0.5) * q[5]) - q[3] + q[7]) / q[1]);
        r2 = 1.0f;
    } else {
        const size_t pix = (size_t)y * p.w + x;
        r0 = p.rays[((size_t)b * 3 + 0) * hw + pix];
        r1 = p.rays[((size_t)b * 3 + 1) * hw + pix];
        r2 = p.rays[((size_t)b * 3 + 2) * hw + pix];
    }
}

// XCD-aware block -> (tile, frame) map.  Hardware round-robins consecutive block ids over the 8
// XCDs; remap so that each XCD (= one private 4 MiB L2) works through one contiguous run of tiles
// (row-major within a frame), because neighbouring tiles read overlapping source-view footprints.
// Bijective for any grid size (cdna_hip_programming.md T1).  Speed only — never correctness.
__device__ __forceinline__ void tile_of_block(const CvParams& p, int& tile, int& b) {
    const unsigned n = gridDim.x, bid = blockIdx.x;
    const unsigned q = n / NUM_XCD, r = n % NUM_XCD;
    const unsigned xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
    const unsigned start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned logical = start + idx;
    const unsigned tiles = (unsigned)(p.tiles_x * p.tiles_y);
    b = (int)(logical / tiles);
    tile = (int)(logical % tiles);
}

hipError_t launch_cv_generic(const CvParams& p, hipStream_t stream);
hipError_t launch_cv_worklist(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cv_cand(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cv_fast(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cv_fast64(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cv_v3(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cv_v4(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cv_v5(const CvParams& p, hipStream_t stream, bool* handled);
hipError_t launch_cvf_bwd(const CvParams& p, const float* gout, float* grad_ref, float* grad_src, hipStream_t stream,
                          bool* handled);
hipError_t launch_cvf_bwd_ref_only(const CvParams& p, const float* gout, float* grad_ref, hipStream_t stream, bool* handled);
size_t cvf_gather_workspace_bytes(const CvParams& p);
hipError_t launch_cvf_gather_src(const CvParams& p, const float* gout, float* grad_src, void* workspace, size_t ws_bytes, hipStream_t stream,
                                 bool* handled);

}  // namespace magnet
