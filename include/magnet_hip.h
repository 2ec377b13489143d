/*
 * magnet_hip.h — C ABI of libmagnet_hip.so, the MI355X (gfx950) implementation of MaGNet's
 * multi-view matching hot path.  Plain pointers and sizes only; no torch types.  Every entry
 * point replaces a specific piece of the reference (file:line into baegwangbin/MaGNet):
 *
 *   magnet_pack_features     -> the layout hand-off from F-Net's NCHW fp32 output
 *                               (models/MAGNET.py:142-144) to the kernel's channel-last storage
 *   magnet_pack_gmm          -> the same for D-Net's source-view (mu,sigma) maps (models/MAGNET.py:139)
 *   magnet_cost_volume_cw    -> homography.est_costvolume_CW  (models/submodules/homography.py:79-121)
 *                               + _compute_cost_CW            (homography.py:124-161)
 *                               + the candidate sampling in front of it (models/MAGNET.py:153-156)
 *   magnet_gaussian_update   -> the element-wise tail of GNET.forward (models/MAGNET.py:60-69)
 *   magnet_upsample_depth    -> upsample_depth_via_mask       (models/MAGNET.py:15-27)
 *   magnet_depth_metrics     -> utils.compute_depth_errors + validate()'s masking (utils/utils.py:106-144,
 *                               test_MaGNet.py:43,58-79), so full depth maps never leave the device
 *   magnet_conv_mfma (+ magnet_pack_split, magnet_gaussian_update_cl, magnet_upsample_depth_cl[_n])
 *                            -> the g_net / mask_head nn.Conv2d stacks (models/MAGNET.py:51-56,111-116)
 *
 * Conventions
 *   - All data pointers are DEVICE pointers unless the comment says HOST.  Buffers are caller-owned;
 *     the library never synchronises the device and keeps no device memory between calls.  ONE exception to
 *     "never allocates": magnet_depth_metrics / magnet_depth_metrics_crop take their B x 64 x 13 doubles of
 *     partial sums from the stream-ordered pool (hipMallocAsync + hipFreeAsync on `stream`, inside the call);
 *     every other entry point works in caller-provided buffers only.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     asynchronous on that stream and re-entrant across streams.
 *   - Return value: 0 on success; >0 = MAGNET_E_* argument error; <0 = -(hipError_t).  Nothing is
 *     thrown.  magnet_last_error() returns a thread-local message for the last failing call.
 *   - Layouts follow the reference: NCHW fp32 for (mu,sigma) maps and the cost volume, source-view
 *     tensors VIEW-MAJOR (index v*B + b, homography.py:105), rays (B,3,h*w), intM (B,3,3),
 *     poses (B,V,4,4) row-major [R|t], is_valid (B,V) int32 (1 = use the view, homography.py:97).
 */
#ifndef MAGNET_HIP_H
#define MAGNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAGNET_API __attribute__((visibility("default")))

#define MAGNET_HIP_VERSION 400            /* major*10000 + minor*100 + patch.
                                           * BINARY COMPATIBILITY: the argument structs below carry no size field and have grown at their END between versions
                                           * (MagnetConvArgs: v300 up_*, v301 gu_*, v302 in_sc / w_sc / sc_rows).  A caller must be compiled against the header whose
                                           * MAGNET_HIP_VERSION equals magnet_version() of the library it loads, and must check that at load time (INTEGRATION.md;
                                           * magnet_amd/lib.py mirrors the structs field for field and tests/test_abi.py compares the layouts with a gcc probe).
                                           * From v400 on a struct change bumps the MINOR number, never only the patch. */

enum {                                     /* storage dtype of channel-last feature maps */
    MAGNET_FEAT_F32  = 0,
    MAGNET_FEAT_BF16 = 1
};

enum {                                     /* argument errors (positive return values) */
    MAGNET_E_NULL     = 1,                 /* a required pointer is NULL */
    MAGNET_E_DIM      = 2,                 /* a dimension is <= 0 or exceeds a kernel limit */
    MAGNET_E_DTYPE    = 3,                 /* unknown dtype enum */
    MAGNET_E_ALIGN    = 4,                 /* a pointer is not 16-byte aligned */
    MAGNET_E_NODEVICE = 5,                 /* no gfx950 device / kernel image not loadable */
    MAGNET_E_SHAPE    = 6                  /* valid arguments, but the kernel selected by `path` (or the requested output form)
                                              does not take this shape / option set: choose another path.  Never a HIP failure. */
};

#define MAGNET_MAX_CANDIDATES 256          /* D limit of magnet_cost_volume_cw */

/* Arguments of magnet_cost_volume_cw.  Zero-initialise, then fill. */
typedef struct MagnetCostVolumeArgs {
    int32_t B, V, F, D, h, w;              /* ref frames, source views, channels (F % 8 == 0), candidates, grid */
    float   kappa;                         /* consistency threshold: int(weighting.split('CW')[1]), MAGNET.py:159 */
    int32_t feat_dtype;                    /* MAGNET_FEAT_* of ref_feat_cl / src_feat_cl */
    const void    *ref_feat_cl;            /* (B, h, w, F) channel-last            (magnet_pack_features, pad = 0) */
    const void    *src_feat_pad;           /* (V*B, h+2, w+2, F) channel-last with a one-texel ZERO border, view-major
                                              (magnet_pack_features, pad = 1): grid_sample's zeros padding
                                              (homography.py:150) becomes plain loads */
    const float   *src_gmm_pad;            /* (V*B, h+2, w+2, 2) interleaved source [mu, sigma], zero border
                                              (magnet_pack_gmm).  May be NULL when src_gmm_quad is given: a call that ends up in a
                                              kernel that reads this layout then returns MAGNET_E_SHAPE */
    const float   *ref_gmm;                /* (B, 2, h, w) reference [mu, sigma]; used when d_volume == NULL */
    const double  *k_list;                 /* HOST, D float64 quantile offsets (MAGNET.depth_sampling, MAGNET.py:120-128);
                                              used when d_volume == NULL: d_j = mu + sigma*(float)k_j */
    const float   *d_volume;               /* optional (B,D,h,w) explicit candidate depths (the reference's first
                                              argument); NULL = fused sampling from ref_gmm + k_list */
    const float   *poses;                  /* (B,V,4,4) relative poses ref->source (utils.data_preprocess) */
    const int32_t *is_valid;               /* (B,V) */
    const float   *intM;                   /* (B,3,3) intrinsics at grid resolution */
    const float   *rays;                   /* (B,3,h*w) unit_ray_array_2D; may be NULL when ray_params is given (path 0/1/2/4) */
    float         *cost;                   /* OUT (B,D,h,w) fp32; frame b starts at cost + b*cost_batch_stride */
    int32_t        path;                   /* kernel selection:
                                              0 = auto: the PRODUCTION matcher whenever the candidates are sampled in the kernel
                                                  (d_volume == NULL, mode 0, stats == NULL); otherwise the exact candidate-lane
                                                  kernel; the generic kernel for shapes neither takes.
                                                  Production contract (TOLERANCE parity with homography.py:124-161): at most a 1e-5 x
                                                  max(1, (max(h, w) + 1) / 161) fraction of the consistency gates differs from the
                                                  reference's (1e-5 up to 160-wide grids; measured 1.2e-5 at the 640-wide C2L grid: both
                                                  sides' positions carry ~ulp(max(h, w)) texels of fp32 rounding); every entry none
                                                  of whose gates differs is within 2e-5 + 2e-5|c| + eps*S of the reference, where
                                                  S = sum over open views of (|dc/dx| + |dc/dy|) is the score's slope in the sample
                                                  position and eps = 4 ulp(max(h, w) + 1) texels (6 ulp on grids wider than 512): the kernel's sample position
                                                  differs from the reference's by its normalise / unnormalise rounding (<= 1.5e-5
                                                  texel), so on features that vary strongly from texel to texel (white noise) the
                                                  plain 2e-5 + 2e-5|c| bound alone does NOT hold; on smooth features it does
                                                  (tests/test_gpu_fast_matcher.py).  Downstream of the volume, north_star's bar — depth abs_rel
                                                  < 1e-4 against the reference loop — holds with a measured worst case of 2.2e-5 (C3 shape,
                                                  bf16 feature storage, I = 3, production matcher + matrix-core convolutions against the
                                                  oracle loop; 8e-8 .. 2.8e-7 against the reference's own forward on its fixtures G6 / G13);
                                              1 = generic gather kernel (bit-exact reference arithmetic, slow);
                                              2 = exact candidate-lane kernel (the reference's fp32 geometry to the bit) or MAGNET_E_SHAPE;
                                              3 = exact pixel-lane worklist kernel or MAGNET_E_SHAPE;
                                              4 = production matcher or MAGNET_E_SHAPE.
                                              Any other value: MAGNET_E_DIM. */
    uint32_t      *stats;                  /* optional device uint32[4]: {tiles run by a fast kernel, tiles run by the
                                              generic kernel, items (distinct open quads) correlated, 0}, accumulated with atomics; NULL = off */
    int64_t        cost_batch_stride;      /* elements between consecutive frames of `cost`; 0 = D*h*w (dense).
                                              Lets the kernel write the first D channels of G-Net's
                                              (B, D+256, h, w) input directly (models/MAGNET.py:167). */
    void          *cost_hi, *cost_lo;      /* optional second output form (candidate-lane kernel only): the cost volume as
                                              split-bf16 planes (hi = bf16(c), lo = bf16(c - hi)) in the conv kernel's
                                              zero-bordered channel-last layout, element [((b*(h+2)+y+1)*(w+2)+x+1)*cost_ld + j].
                                              When set, `cost` may be NULL (nothing is written there). */
    int64_t        cost_ld;                /* row pitch (elements) of cost_hi / cost_lo */
    int32_t        mode;                   /* 0 = consistency-weighted matcher (est_costvolume_CW);
                                              1 = plain feature-matching volume of est_costvolume_F / _compute_cost_F
                                                  (homography.py:10-75) BEFORE its softmax: k_list holds the D fixed depth
                                                  bins d_center (same for all pixels), there is no gate (src_gmm_pad,
                                                  ref_gmm unused, may be NULL) and views are summed in fp32 (:42).
                                                  Candidate-lane and generic kernels only. */
    uint8_t       *gate_bits;              /* optional debug OUT (B,V,D,h,w) uint8: the consistency gate of every
                                              (view, candidate, pixel) sample, homography.py:157-158 (1 = open).  Written by the
                                              production matcher (path 0/4) and by the exact candidate-lane kernel (path 2);
                                              entries of invalid views are not written (zero the buffer first).  NULL = off. */
    const double  *ray_params;             /* optional DEVICE (B,8) float64 {fx, fy, cx, cy, sx, sy, left, top} of the RAW image
                                              (sx = raw_w / w, sy = raw_h / h; left/top = crop margins, KITTI): with rays == NULL the
                                              production and candidate-lane / generic kernels evaluate the loaders' expression
                                              ((x+0.5)*sx - cx + left)/fx in float64 per reference pixel instead of reading the
                                              12*h*w-byte table (dataloader_scannet.py:139-147, dataloader_kitti.py:113-118) —
                                              bit-identical to it.  The worklist kernel and the backward need the table. */
    const float   *src_gmm_quad;           /* optional (V*B, h+2, w+2, 8): the source (mu, sigma) map per QUAD ORIGIN of the padded map in
                                              quad form {m00, m10-m00, m01-m00, m11-m10-m01+m00, s00, ...} (magnet_pack_gmm_quad), so that
                                              a bilinear sample is 3 fma per channel.  The production matcher for D > 32 needs it
                                              (without it path 0 runs the round-2 production kernel, path 4 too). */
    uint32_t       dev_flags;              /* development switches (timing ablations, kernel variants of tools/ablate.py).  Ignored
                                              unless the library was built with -DMAGNET_DEV (python -m magnet_amd.build --dev);
                                              results are not meaningful with any of them set. */
} MagnetCostVolumeArgs;

MAGNET_API int magnet_version(void);
MAGNET_API const char *magnet_last_error(void);

/* Number of gfx950 devices visible to the HIP runtime this library is bound to (0 if none). */
MAGNET_API int magnet_device_count(void);

/* NCHW fp32 (N,F,h,w) -> channel-last (N,h+2*pad,w+2*pad,F) in `out_dtype` (round-to-nearest-even for
 * bf16); pad = 1 surrounds every image with one texel of zeros. */
MAGNET_API int magnet_pack_features(const float *nchw, void *out_cl, int32_t N, int32_t F, int32_t h, int32_t w,
                         int32_t out_dtype, int32_t pad, void *stream);

/* (N,2,h,w) fp32 [mu,sigma] planes -> (N,h+2,w+2,2) interleaved with a one-texel zero border. */
MAGNET_API int magnet_pack_gmm(const float *gmm_nchw, float *out_pad, int32_t N, int32_t h, int32_t w, void *stream);

/* (N,2,h,w) fp32 [mu,sigma] planes -> (N,h+2,w+2,8): entry (y0,x0) describes the 2x2 quad of the zero-bordered map whose top-left
 * texel is padded position (y0,x0) (texels outside the padded map count as zero), as
 * {m00, m10-m00, m01-m00, (m11-m01)-(m10-m00), s00, s10-s00, s01-s00, (s11-s01)-(s10-s00)} (x is the fast index of mXY).
 * Feeds MagnetCostVolumeArgs.src_gmm_quad (grid_sample of the (mu, sigma) maps, homography.py:151-152). */
MAGNET_API int magnet_pack_gmm_quad(const float *gmm_nchw, float *out_quad, int32_t N, int32_t h, int32_t w, void *stream);

/* Consistency-weighted multi-view matching score, all (b, pixel, candidate) in one launch. */
MAGNET_API int magnet_cost_volume_cw(const MagnetCostVolumeArgs *args, void *stream);

/* Backward of mode 1 (what autograd derives for est_costvolume_F up to its softmax, homography.py:10-75,
 * used by train_scripts/train_FNET/train.py): given grad_cost (B,D,h,w) = dL/d(cost before softmax) and
 * the SAME args as the forward call (fp32 features, mode = 1; `cost` is ignored),
 *   grad_ref_cl  (B,h,w,F)         fp32 channel-last, fully written;
 *   grad_src_pad (V*B,h+2,w+2,F)   fp32 padded channel-last, ACCUMULATED into with atomics — zero it first;
 *                                  what lands in the one-texel border is the (discarded) gradient of the
 *                                  zero padding.
 * Depth bins, poses and intrinsics receive no gradient (they are data in the reference's training loop). */
MAGNET_API int magnet_cost_volume_f_backward(const MagnetCostVolumeArgs *args, const float *grad_cost, float *grad_ref_cl,
                                  float *grad_src_pad, void *stream);

/* ---- camera intrinsics and relative poses on the device (the loaders' / utils.data_preprocess' host work; row N4) ----
 * magnet_make_rays: unit_ray_array_2D (B,3,h*w) fp32 from DEVICE ray_params (B,8) float64 {fx, fy, cx, cy, sx, sy, left, top}
 *   (see MagnetCostVolumeArgs.ray_params; data/dataloader_scannet.py:113-153, dataloader_kitti.py:83-127, dataloader_7scenes.py:72-116).
 * magnet_relative_poses: utils.data_preprocess (utils/utils.py:72-98): DEVICE float64 extrinsics ext_ref (B,4,4), ext_nghbr
 *   (B,V,4,4) -> poses_out (B,V,4,4) fp32 = ext_nghbr @ inv(ext_ref) and is_valid_out (B,V) int32 (0 and a zero pose when either
 *   matrix holds a NaN or the reference is singular).  Both outputs are what magnet_cost_volume_cw consumes. */
MAGNET_API int magnet_make_rays(const double *ray_params, float *rays_out, int32_t B, int32_t h, int32_t w, void *stream);
MAGNET_API int magnet_relative_poses(const double *ext_ref, const double *ext_nghbr, float *poses_out, int32_t *is_valid_out,
                                     int32_t B, int32_t V, void *stream);

/* The same backward with a caller-provided device workspace of magnet_cost_volume_f_backward_workspace(args) bytes: grad_src is then
 * computed as a GATHER per source row segment (cost_volume_f_gather.hip) — no atomics, deterministic, every interior texel of
 * grad_src_pad stored exactly once (the one-texel border is left as the caller initialised it) — and grad_ref by the per-item
 * kernel.  Shapes the gather path does not take fall back to the scatter kernels above (which accumulate: zero grad_src_pad
 * first to be safe).  The workspace query returns -1 for invalid arguments. */
MAGNET_API int64_t magnet_cost_volume_f_backward_workspace(const MagnetCostVolumeArgs *args);
MAGNET_API int magnet_cost_volume_f_backward_ws(const MagnetCostVolumeArgs *args, const float *grad_cost, float *grad_ref_cl,
                                     float *grad_src_pad, void *workspace, int64_t workspace_bytes, void *stream);

/* gmm_out[:,0] = mu + o0*sigma ; gmm_out[:,1] = (elu(o1) + 1 + 1e-10)*sigma.   All (B,2,h*w) fp32.
 * gmm_out may alias gmm_in. */
MAGNET_API int magnet_gaussian_update(const float *gnet_out, const float *gmm_in, float *gmm_out,
                           int32_t B, int32_t hw, void *stream);

/* Learned convex upsampling: depth (B,C,h,w), mask (B,9*k*k,h,w) -> out (B,C,k*h,k*w); softmax over
 * the 9 neighbours, zero padding.  k <= 8. */
MAGNET_API int magnet_upsample_depth(const float *depth, const float *mask, float *out,
                          int32_t B, int32_t C, int32_t h, int32_t w, int32_t k, void *stream);

/* ---- G-Net / mask-head convolutions on the bf16 matrix cores, bf16x3 split operands (fp32-grade) ----
 * Replaces the nn.Conv2d stacks of models/MAGNET.py:51-56 (GNET.gnet) and :111-116 (mask_head).
 * Activations live in ZERO-BORDERED channel-last buffers (B, h+2, w+2, C) stored as two bf16 planes
 * (hi = bf16(x), lo = bf16(x - hi)); `rows` = B*(h+2)*(w+2) flattened positions; a 3x3 tap is the row
 * offset dy*(w+2)+dx.  Border rows of the output hold unspecified finite values: read interior rows only.
 * Weights: two bf16 planes of [taps][cout_pad][cin] (cin contiguous) — magnet_amd/convnet.py prepacks them
 * from nn.Conv2d's [cout][cin][kh][kw]; bias fp32 [cout_pad] (zero in the padding).
 * Limits: cin % 32 == 0; taps in {1, 4, 9}; cout_pad a multiple of 128, or 144, 64, 32, 16.
 * The same kernel runs the F-Net (PSMNet feature extractor, models/submodules/F_psmnet.py:37-124) — see the fields
 * after `addend_ld` (all zero = the behaviour described above). */
typedef struct MagnetConvArgs {
    const void  *in_hi, *in_lo;            /* bf16 (rows, cin) */
    const void  *w_hi, *w_lo;              /* bf16 (taps, cout_pad, cin) */
    const float *bias;                     /* (cout_pad) */
    void        *out_hi, *out_lo;          /* out_mode 0: bf16 (rows, cout_pad) planes */
    float       *out_f32;                  /* out_mode 1: fp32 (rows, cout_pad) */
    int64_t      rows;
    int32_t      cin, cout_pad, taps, wp;  /* wp = w + 2 (row pitch of the padded grid) */
    int32_t      relu, out_mode;
    int32_t      in_ld;                    /* elements between input rows (0 = cin): lets a layer read a channel slice
                                              of a wider buffer in place (pointer offset + in_ld) */
    const float *addend;                   /* optional fp32 (rows, addend_ld): added to the accumulator before bias / ReLU.
                                              Used to hoist the loop-invariant x_d3 part of G-Net's first layer out of
                                              the refinement loop (models/MAGNET.py:151-168): W*[cost|x_d3] = Wc*cost + Wx*x_d3 */
    int32_t      addend_ld;                /* row pitch of addend (0 = cout_pad) */
    int32_t      dil;                      /* taps = 9: dilation (0 or 1 = none); the grid's border must be >= dil wide
                                              (F_psmnet.py:47 layer4).  taps = 4: the 2x2 window (-1,-1),(-1,0),(0,-1),(0,0)
                                              of a space-to-depth tensor = a stride-2 3x3 convolution (F_psmnet.py:40,45) */
    int32_t      out_ld;                   /* elements between output rows (0 = cout_pad): write a channel slice of a wider
                                              buffer in place (the 320-channel concatenation, F_psmnet.py:122) */
    const void  *add_hi, *add_lo;          /* optional split-bf16 (rows, add_ld) residual input added before bias / ReLU
                                              (BasicBlock `out += x`, F_psmnet.py:27-33) */
    int32_t      add_ld;
    int32_t      border_hp;                /* > 0: rows are positions of (image, border_hp, wp) grids with a `border_pad`-wide
                                              border; outputs at border positions are written as ZEROS (they are the next
                                              layer's zero padding) */
    int32_t      border_pad;
    int32_t      repad;                    /* > 0 (needs border_hp): only interior positions are written, re-addressed into
                                              grids with a (repad-1)-wide border: hands the last layer's output to the
                                              matcher's layouts (border 0 = reference features, 1 = source features) */
    const void  *tail_w_hi, *tail_w_lo;    /* optional fused 1x1 tail (models/MAGNET.py:53-55, :113-115): after THIS layer (cout_pad
                                              must be 128; bias / relu / addend apply) the workgroup's 128-row tile stays in LDS and
                                              relu(1x1 128->128), relu(1x1 128->128), 1x1 128->tail_cout_pad run in the same kernel.
                                              Weights / bias laid out as for magnet_conv1x1_chain.  The only output is fp32
                                              (rows, tail_cout_pad) at out_f32 (out_mode, out_hi/lo, out_ld are ignored): the three
                                              hidden (rows,128) tensors never reach HBM. */
    const float *tail_bias;
    int32_t      tail_cout_pad;            /* 16, 128 or 144 */
    /* v301 — fused learned convex upsampling (models/MAGNET.py:15-27,172-173), for the mask head's stack (tail_cout_pad = 144):
     * the tail's last layer keeps the (rows, 144) mask logits in registers, soft-maxes the 9 neighbour weights of each of the 16
     * sub-pixels and writes the x4-upsampled maps itself: up_depth (up_npred, up_B, 2, up_h, up_w) fp32 -> up_out (up_npred, up_B,
     * 2, 4 up_h, 4 up_w) fp32.  Requires rows = up_B * (up_h + 2) * (up_w + 2), wp = up_w + 2; out_f32 is not written.
     * All zero / NULL = the plain tail above (magnet_upsample_depth_cl_n then does this step from out_f32). */
    const float *up_depth;
    float       *up_out;
    int32_t      up_npred, up_B, up_h, up_w;
    /* v301 — fused Gaussian update (models/MAGNET.py:60-69), for G-Net's stack (tail_cout_pad = 16): the head's two outputs of every
     * interior position update (mu, sigma): gu_in (up_B, 2, up_h, up_w) fp32 -> gu_out (same layout; may not alias gu_in); up_B /
     * up_h / up_w as above, out_f32 is not written.  NULL = plain tail (magnet_gaussian_update_cl then does this step). */
    const float *gu_in;
    float       *gu_out;
    /* v302 — the "2-unit" operand format of the 128-wide 3x3 layers with a fused tail (taps = 9, rows >= 65536, no addend): in_hi / w_hi
     * are FP16 planes (same shapes as the bf16 hi planes), in_lo / w_lo hold per 64-byte (row, 32-channel block) slice the 32 OCP e4m3
     * bytes of hi / 2^e_h followed by the 32 e4m3 bytes of (x - fp16(x)) / 2^e_l, and the E8M0 exponents travel separately:
     * in_sc [cin / 32][sc_rows] uint32 {e_h + 127, e_l + 127, 0, 0} per (block, row), w_sc [taps][cin / 32][cout_pad] uint32 likewise per
     * (tap, block, output channel).  magnet_pack_mx writes the activation planes; magnet_amd/convnet.py prepares the weights.  The kernel
     * runs hi*hi on the fp16 matrix instruction and lo*hi + hi*lo on the block-scaled fp8 one (x*w to ~1e-5 relative, as the bf16x3 form):
     * 2 matrix-pipe units per product term instead of 3.  NULL / 0 = the bf16x3 format described above.
     * DOMAIN of this format: |x| <= 65504 (the fp16 plane); magnet_pack_mx clamps larger magnitudes, +-Inf included, to +-65504 (the
     * bf16x3 format has no such limit).  NON-FINITE inputs: a NaN is written through to every plane (fp16 NaN, e4m3 NaN; the block
     * exponents are taken over the finite entries) and the outputs that read it are NaN in BOTH formats; +-Inf propagates as Inf in the
     * bf16x3 format and is clamped here — the one documented difference between the two formats.  `bias` is read with 16-byte loads in every format: 16-byte aligned, as all pointers of this struct. */
    const void  *in_sc, *w_sc;
    int64_t      sc_rows;
} MagnetConvArgs;                          /* out_mode 2: one bf16 plane (round-to-nearest-even) at out_hi */

MAGNET_API int magnet_conv_mfma(const MagnetConvArgs *args, void *stream);

/* fp32 NCHW (N, C, h, w) -> channels [c_off, c_off + C) of the interior of a (N, h+2, w+2, ctot) buffer in the v302 operand format of
 * magnet_conv_mfma: out_f16 (fp16 plane), out_qr (e4m3 hi | lo bytes, 64 B per (row, 32-channel block)), out_sc [ctot / 32][sc_rows]
 * uint32 E8M0 pairs.  C, c_off, ctot multiples of 32; h*w a multiple of 4; the border rows are not written (zero them once). */
MAGNET_API int magnet_pack_mx(const float *nchw, void *out_f16, void *out_qr, void *out_sc, int32_t N, int32_t C, int32_t h, int32_t w,
                              int32_t ctot, int32_t c_off, int64_t sc_rows, int64_t in_img_stride, void *stream);

/* ---- the F-Net's non-GEMM layers (PSMNet feature extractor, models/submodules/F_psmnet.py:37-124; row N3) ----
 * Activations are conv_mfma's format: zero-bordered channel-last grids as two bf16 planes (hi, lo). */

/* firstconv[0] (F_psmnet.py:40): 3x3 stride-2 pad-1 convolution 3 -> 32 channels of the NCHW fp32 image (N,3,H,W) with the
 * BatchNorm folded into wgt (32,3,3,3) / bias (32), then ReLU -> planes (N, H2+2, W2+2, 32), border 1 (interior written),
 * H2 = (H-1)/2+1, W2 = (W-1)/2+1. */
MAGNET_API int magnet_fnet_stem(const float *img, const float *wgt, const float *bias, void *out_hi, void *out_lo,
                                int32_t N, int32_t H, int32_t W, void *stream);

/* (N, H2+2, W2+2, C) border 1 -> (N, H4+2*opad, W4+2*opad, 4*C) border opad, channel (py*2+px)*C + c = in[2y+py, 2x+px, c]
 * (zero where 2y+py >= H2 or 2x+px >= W2); H4 = (H2-1)/2+1.  With it the stride-2 convolutions of layer2[0]
 * (F_psmnet.py:45,89-93) are stride-1 GEMMs: 3x3/s2 = conv_mfma taps=4 over 4*C channels, 1x1/s2 = taps=1 on phase 0.
 * C % 8 == 0; interior written only. */
MAGNET_API int magnet_space_to_depth(const void *in_hi, const void *in_lo, void *out_hi, void *out_lo, int32_t N, int32_t C,
                                     int32_t H2, int32_t W2, int32_t opad, void *stream);

/* AvgPool2d((k,k), stride (k,k)) (F_psmnet.py:50-64) of channels [0,C) at in_hi/in_lo (row pitch ld elements) of an
 * (N, h+2*pad, w+2*pad) grid -> planes (N*(h/k)*(w/k), C).  C <= 128, C % 8 == 0. */
MAGNET_API int magnet_avgpool_cl(const void *in_hi, const void *in_lo, int32_t ld, int32_t N, int32_t h, int32_t w, int32_t pad,
                                 int32_t k, int32_t C, void *out_hi, void *out_lo, void *stream);

/* F.interpolate(..., mode='bilinear', align_corners=True) (F_psmnet.py:108-119): fp32 (N*ph*pw, in_ld) -> planes, channels
 * [0,C) at out_hi/out_lo (row pitch out_ld) of an (N, h+2*pad, w+2*pad) grid, interior only.  C % 8 == 0. */
MAGNET_API int magnet_upsample_bilinear_cl(const float *in, int32_t in_ld, int32_t ph, int32_t pw, int32_t C, void *out_hi,
                                           void *out_lo, int32_t out_ld, int32_t N, int32_t h, int32_t w, int32_t pad,
                                           void *stream);

/* The 1x1 tail of a stack in one launch: relu(conv1x1 128->128), relu(conv1x1 128->128), conv1x1 128->cout_pad.
 * in: split-bf16 (rows,128) planes (the 3x3 layer's out_mode-0 output); w_hi/w_lo: the three layers' [cout][128]
 * bf16 planes concatenated; bias: 128 + 128 + cout_pad floats; out: fp32 (rows, cout_pad).  cout_pad in {16,128,144}.
 * (models/MAGNET.py:53-55 and :113-115.) */
MAGNET_API int magnet_conv1x1_chain(const void *in_hi, const void *in_lo, const void *w_hi, const void *w_lo,
                                    const float *bias, float *out, int64_t rows, int32_t cout_pad, void *stream);

/* fp32 NCHW (N, C, h, w) (image stride `in_img_stride` elements, 0 = C*h*w) -> the interior of the split-bf16
 * padded channel-last buffer (N, h+2, w+2, ctot), channels [c_off, c_off+round_up(C,8)) (the round-up lanes are
 * written as zeros).  c_off % 8 == 0.  The border must have been zeroed once by the caller (never written). */
MAGNET_API int magnet_pack_split(const float *nchw, void *out_hi, void *out_lo, int32_t N, int32_t C, int32_t h,
                                 int32_t w, int32_t ctot, int32_t c_off, int64_t in_img_stride, void *stream);

/* G-Net tail on the conv kernel's fp32 output: o = gnet_out[(b, y+1, x+1), 0..1] of a (B, h+2, w+2, ld) buffer;
 * gmm_out = [mu + o0*sigma, (elu(o1) + 1 + 1e-10)*sigma], all gmm tensors (B,2,h,w) fp32 (models/MAGNET.py:60-69). */
MAGNET_API int magnet_gaussian_update_cl(const float *gnet_out_pad, int32_t ld, const float *gmm_in, float *gmm_out,
                                         int32_t B, int32_t h, int32_t w, void *stream);

/* Learned convex upsampling with the mask in the conv kernel's padded channel-last fp32 layout
 * (B, h+2, w+2, ld), channel n*k*k + i*k + j as in models/MAGNET.py:19; depth (B,2,h,w) -> out (B,2,4h,4w); k = 4. */
MAGNET_API int magnet_upsample_depth_cl(const float *depth, const float *mask_pad, int32_t ld, float *out,
                                        int32_t B, int32_t h, int32_t w, void *stream);
/* The same for n_pred stacked predictions in one launch: depths (n_pred,B,2,h,w), outs (n_pred,B,2,4h,4w); the mask is read and
 * soft-maxed once (models/MAGNET.py:173 upsamples every iteration's prediction with the same mask). */
MAGNET_API int magnet_upsample_depth_cl_n(const float *depths, const float *mask_pad, int32_t ld, float *outs, int32_t n_pred,
                                          int32_t B, int32_t h, int32_t w, void *stream);

/* Depth-error reductions on the device (utils.compute_depth_errors, utils/utils.py:106-144, with the masking /
 * clamping of test_MaGNet.py:43,58-79): pred (B,2,H*W) [mu, sigma], gt (B,H*W).  sums: OUT double (B,16), every
 * element written by the call (no memset needed): n, sum|d|, sum|d|/gt, sum d^2/gt, sum d^2, sum(ln gt-ln p)^2, sum(ln p-ln gt), sum|log10 gt-log10 p|,
 * sum(1/gt-1/p)^2, #(t<1.25), #(t<1.25^2), #(t<1.25^3), sum nll, 0,0,0.  The 12 metrics are ratios of these
 * (magnet_amd/metrics.py). */
MAGNET_API int magnet_depth_metrics(const float *pred, const float *gt, double *sums, int32_t B, int32_t HW,
                                    float min_depth, float max_depth, void *stream);
/* The same over the evaluation window rows [y0,y1) x columns [x0,x1) only: the reference's garg / eigen crops for KITTI
 * (test_MaGNet.py:63-71).  Both entry points sum in a fixed order (64 workgroups per frame
 * writing partial sums to a stream-ordered scratch allocation, then one fixed-order final sum): deterministic, no atomics. */
MAGNET_API int magnet_depth_metrics_crop(const float *pred, const float *gt, double *sums, int32_t B, int32_t H, int32_t W,
                                         float min_depth, float max_depth, int32_t y0, int32_t y1, int32_t x0, int32_t x1,
                                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MAGNET_HIP_H */
