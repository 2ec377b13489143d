// api.hip — the extern "C" boundary declared in include/magnet_hip.h.  Argument checking, error
// reporting, kernel selection.  No device allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "cv_common.hpp"
#include "conv_common.hpp"

namespace magnet {
hipError_t launch_pack(const float*, void*, int, int, int, int, bool, int, hipStream_t);
hipError_t launch_pack_gmm(const float*, float*, int, int, int, hipStream_t);
hipError_t launch_pack_gmm_quad(const float*, float*, int, int, int, hipStream_t);
hipError_t launch_gaussian_update(const float*, const float*, float*, int, int, hipStream_t);
hipError_t launch_upsample(const float*, const float*, float*, int, int, int, int, int, hipStream_t);
hipError_t launch_pack_split(const float*, uint16_t*, uint16_t*, int, int, int, int, int, int, long long, hipStream_t);
hipError_t launch_gaussian_update_cl(const float*, int, const float*, float*, int, int, int, hipStream_t);
hipError_t launch_upsample_cl(const float*, const float*, int, float*, int, int, int, int, hipStream_t);
hipError_t launch_fnet_stem(const float*, const float*, const float*, uint16_t*, uint16_t*, int, int, int, hipStream_t);
hipError_t launch_space_to_depth(const uint16_t*, const uint16_t*, uint16_t*, uint16_t*, int, int, int, int, int, hipStream_t);
hipError_t launch_avgpool_cl(const uint16_t*, const uint16_t*, int, int, int, int, int, int, int, uint16_t*, uint16_t*, hipStream_t);
hipError_t launch_upsample_bilinear_cl(const float*, int, int, int, int, uint16_t*, uint16_t*, int, int, int, int, int, hipStream_t);
hipError_t launch_depth_metrics(const float*, const float*, double*, int, int, int, float, float, int, int, int, int, hipStream_t);
hipError_t launch_make_rays(const double*, float*, int, int, int, hipStream_t);
hipError_t launch_relative_poses(const double*, const double*, float*, int32_t*, int, int, hipStream_t);
}

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(hipError_t e, const char* what) {
    return fail(-(int)e, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" {

MAGNET_API int magnet_version(void) { return MAGNET_HIP_VERSION; }

MAGNET_API const char* magnet_last_error(void) { return g_err; }

MAGNET_API int magnet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

MAGNET_API int magnet_pack_features(const float* nchw, void* out_cl, int32_t N, int32_t F, int32_t h, int32_t w,
                         int32_t out_dtype, int32_t pad, void* stream) {
    if (!nchw || !out_cl) return fail(MAGNET_E_NULL, "magnet_pack_features: NULL pointer");
    if (N <= 0 || F <= 0 || h <= 0 || w <= 0 || (F % 8) != 0)
        return fail(MAGNET_E_DIM, "magnet_pack_features: bad dims N=%d F=%d h=%d w=%d (F must be a multiple of 8)", N, F, h, w);
    if (out_dtype != MAGNET_FEAT_F32 && out_dtype != MAGNET_FEAT_BF16)
        return fail(MAGNET_E_DTYPE, "magnet_pack_features: unknown dtype %d", out_dtype);
    if (pad != 0 && pad != 1) return fail(MAGNET_E_DIM, "magnet_pack_features: pad must be 0 or 1");
    if (!aligned16(out_cl)) return fail(MAGNET_E_ALIGN, "magnet_pack_features: out_cl not 16-byte aligned");
    hipError_t e = magnet::launch_pack(nchw, out_cl, N, F, h, w, out_dtype == MAGNET_FEAT_BF16, pad, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_pack_features launch");
}

MAGNET_API int magnet_pack_gmm(const float* gmm_nchw, float* out_pad, int32_t N, int32_t h, int32_t w, void* stream) {
    if (!gmm_nchw || !out_pad) return fail(MAGNET_E_NULL, "magnet_pack_gmm: NULL pointer");
    if (N <= 0 || h <= 0 || w <= 0) return fail(MAGNET_E_DIM, "magnet_pack_gmm: bad dims N=%d h=%d w=%d", N, h, w);
    hipError_t e = magnet::launch_pack_gmm(gmm_nchw, out_pad, N, h, w, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_pack_gmm launch");
}

MAGNET_API int magnet_pack_gmm_quad(const float* gmm_nchw, float* out_quad, int32_t N, int32_t h, int32_t w, void* stream) {
    if (!gmm_nchw || !out_quad) return fail(MAGNET_E_NULL, "magnet_pack_gmm_quad: NULL pointer");
    if (N <= 0 || h <= 0 || w <= 0) return fail(MAGNET_E_DIM, "magnet_pack_gmm_quad: bad dims N=%d h=%d w=%d", N, h, w);
    if (!aligned16(out_quad)) return fail(MAGNET_E_ALIGN, "magnet_pack_gmm_quad: out_quad not 16-byte aligned");
    hipError_t e = magnet::launch_pack_gmm_quad(gmm_nchw, out_quad, N, h, w, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_pack_gmm_quad launch");
}

// argument checks + launch parameters shared by the matcher and the backward of its mode 1
static int cv_prepare(const MagnetCostVolumeArgs* a, magnet::CvParams& p, bool backward) {
    if (!a) return fail(MAGNET_E_NULL, "magnet_cost_volume_cw: args is NULL");
    if (!a->ref_feat_cl || !a->src_feat_pad || (!a->src_gmm_pad && !a->src_gmm_quad && a->mode != 1) || !a->poses || !a->is_valid || !a->intM ||
        (!a->rays && !a->ray_params) || (!backward && !a->cost && !a->cost_hi))
        return fail(MAGNET_E_NULL, "magnet_cost_volume_cw: a required pointer is NULL");
    if (!a->rays && (backward || a->path == 3))
        return fail(MAGNET_E_NULL, "magnet_cost_volume_cw: this kernel reads the ray table (ray_params alone serves path 0/1/2/4 forward): "
                                   "build it with magnet_make_rays");
    if (a->mode != 0 && a->mode != 1) return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: unknown mode %d", a->mode);
    if (a->mode == 1 && (!a->k_list || a->d_volume))
        return fail(MAGNET_E_NULL, "magnet_cost_volume_cw: mode 1 takes its depth bins from k_list (d_volume must be NULL)");
    if (a->mode == 1 && a->path == 3)
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: mode 1 is not implemented in the worklist kernel");
    if (a->mode == 0 && !a->d_volume && (!a->ref_gmm || !a->k_list))
        return fail(MAGNET_E_NULL, "magnet_cost_volume_cw: d_volume is NULL, so ref_gmm and k_list are required");
    if (a->B <= 0 || a->V <= 0 || a->F <= 0 || a->D <= 0 || a->h <= 0 || a->w <= 0)
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: non-positive dimension");
    if (a->D > MAGNET_MAX_CANDIDATES)
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: D=%d exceeds MAGNET_MAX_CANDIDATES=%d", a->D, MAGNET_MAX_CANDIDATES);
    if ((a->F % 8) != 0) return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: F=%d must be a multiple of 8", a->F);
    if (a->feat_dtype != MAGNET_FEAT_F32 && a->feat_dtype != MAGNET_FEAT_BF16)
        return fail(MAGNET_E_DTYPE, "magnet_cost_volume_cw: unknown feat_dtype %d", a->feat_dtype);
    if (a->src_gmm_quad && !aligned16(a->src_gmm_quad)) return fail(MAGNET_E_ALIGN, "magnet_cost_volume_cw: src_gmm_quad must be 16-byte aligned");
    if (!aligned16(a->ref_feat_cl) || !aligned16(a->src_feat_pad))
        return fail(MAGNET_E_ALIGN, "magnet_cost_volume_cw: feature pointers must be 16-byte aligned");
    if (a->V > 64) return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: V=%d exceeds 64 source views", a->V);
    if ((size_t)a->h * a->w * (size_t)a->F * (size_t)a->V * (size_t)a->B >= ((size_t)1 << 40))
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: problem too large");

    memset(&p, 0, sizeof(p));
    p.B = a->B; p.V = a->V; p.F = a->F; p.D = a->D; p.h = a->h; p.w = a->w;
    p.tiles_x = (a->w + magnet::TILE_W - 1) / magnet::TILE_W;
    p.tiles_y = (a->h + magnet::TILE_H - 1) / magnet::TILE_H;
    if ((size_t)p.tiles_x * p.tiles_y * (size_t)p.B >= 0x7fffffffu)
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: grid too large");
    p.feat_bf16 = (a->feat_dtype == MAGNET_FEAT_BF16);
    p.kappa = a->kappa;
#ifdef MAGNET_DEV
    p.ablate = (int)(a->dev_flags & 0xffffffu);                    // development switches: never in the product build
    { static const int forced = getenv("MAGNET_DEV_FLAGS") ? (int)strtol(getenv("MAGNET_DEV_FLAGS"), nullptr, 0) : 0; p.ablate |= forced; }   // dev: run the test-suite against a variant
#else
    p.ablate = 0;
#endif
    p.mode_f = a->mode;
    p.ref_feat = a->ref_feat_cl; p.src_feat = a->src_feat_pad; p.src_gmm = a->src_gmm_pad; p.src_gmq = a->src_gmm_quad;
    p.ref_gmm = a->ref_gmm; p.d_volume = a->d_volume; p.poses = a->poses; p.is_valid = a->is_valid;
    p.intM = a->intM; p.rays = a->rays; p.cost = a->cost; p.stats = a->stats;
    p.cost_hi = (uint16_t*)a->cost_hi; p.cost_lo = (uint16_t*)a->cost_lo; p.cost_ld = a->cost_ld;
    p.gate_bits = a->gate_bits;
    p.ray_params = a->rays ? nullptr : a->ray_params;           // the table wins when both are given
    if (a->cost_hi && (!a->cost_lo || a->cost_ld < a->D))
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: cost_hi needs cost_lo and cost_ld >= D");
    if (a->cost_hi && a->path != 0 && a->path != 2 && a->path != 4)
        return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: the split channel-last output exists only in the candidate-lane kernels (path 0/2/4)");
    if (a->gate_bits && (a->path == 1 || a->path == 3 || a->mode != 0))
        return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: gate_bits is written by the candidate-lane kernels only (path 0/2/4, mode 0)");
    p.cost_bstride = a->cost_batch_stride ? a->cost_batch_stride : (long long)a->D * a->h * a->w;
    if (p.cost_bstride < (long long)a->D * a->h * a->w)
        return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: cost_batch_stride smaller than D*h*w");
    if (!a->d_volume)
        for (int j = 0; j < a->D; ++j) p.k[j] = (float)a->k_list[j];     // MAGNET.py:155: fp32 scalar multiply
    return 0;
}

MAGNET_API int magnet_cost_volume_cw(const MagnetCostVolumeArgs* a, void* stream) {
    magnet::CvParams p;
    if (const int rc = cv_prepare(a, p, false)) return rc;
    hipError_t e = hipSuccess;
    bool handled = false;
    const int path = a->path;
    if (path < 0 || path > 4) return fail(MAGNET_E_DIM, "magnet_cost_volume_cw: unknown path %d", path);
    if (path == 0 || path == 4) {
        e = magnet::launch_cv_fast(p, (hipStream_t)stream, &handled);
        if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_cw production-matcher launch");
        if (!handled && path == 4)
            return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: the production matcher needs fused sampling (d_volume == NULL), mode 0, "
                                      "stats == NULL and F*sizeof(feature) <= 512");
    }
    if (!handled && !a->src_gmm_pad && a->mode != 1)      // only the quad-form (mu, sigma) map was given and the kernel that reads it does not take this call
        return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: this shape / path reads src_gmm_pad (magnet_pack_gmm); src_gmm_quad alone serves the "
                                    "production matcher for D > 32 only");
    if (!handled && (path == 0 || path == 2)) {
        e = magnet::launch_cv_cand(p, (hipStream_t)stream, &handled);
        if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_cw candidate-lane launch");
        if (!handled && path == 2)
            return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: the candidate-lane kernel does not take this shape");
    } else if (handled) {
    } else if (path == 3) {
        e = magnet::launch_cv_worklist(p, (hipStream_t)stream, &handled);
        if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_cw worklist launch");
        if (!handled)
            return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: the worklist kernel does not take this shape (D > 128 or very wide F)");
    }
    if (!handled) {
        if (a->cost_hi || a->gate_bits) return fail(MAGNET_E_SHAPE, "magnet_cost_volume_cw: this shape falls back to the generic kernel, which has no split / gate-bit output");
        e = magnet::launch_cv_generic(p, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_cw generic launch");
    }
    return 0;
}

MAGNET_API int magnet_cost_volume_f_backward(const MagnetCostVolumeArgs* a, const float* grad_cost, float* grad_ref_cl,
                                             float* grad_src_pad, void* stream) {
    magnet::CvParams p;
    if (const int rc = cv_prepare(a, p, true)) return rc;
    if (a->mode != 1) return fail(MAGNET_E_DIM, "magnet_cost_volume_f_backward: only mode 1 (est_costvolume_F) is differentiable");
    if (a->feat_dtype != MAGNET_FEAT_F32) return fail(MAGNET_E_DTYPE, "magnet_cost_volume_f_backward: fp32 features required");
    if (!grad_cost || !grad_ref_cl || !grad_src_pad) return fail(MAGNET_E_NULL, "magnet_cost_volume_f_backward: NULL gradient pointer");
    if (!aligned16(grad_ref_cl) || !aligned16(grad_src_pad))
        return fail(MAGNET_E_ALIGN, "magnet_cost_volume_f_backward: gradient buffers must be 16-byte aligned");
    bool handled = false;
    hipError_t e = magnet::launch_cvf_bwd(p, grad_cost, grad_ref_cl, grad_src_pad, (hipStream_t)stream, &handled);
    if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_f_backward launch");
    if (!handled) return fail(MAGNET_E_DIM, "magnet_cost_volume_f_backward: shape not supported (F > 128, V > 31 or image too large)");
    return 0;
}

MAGNET_API int magnet_make_rays(const double* ray_params, float* rays_out, int32_t B, int32_t h, int32_t w, void* stream) {
    if (!ray_params || !rays_out) return fail(MAGNET_E_NULL, "magnet_make_rays: NULL pointer");
    if (B <= 0 || h <= 0 || w <= 0) return fail(MAGNET_E_DIM, "magnet_make_rays: bad dims B=%d h=%d w=%d", B, h, w);
    hipError_t e = magnet::launch_make_rays(ray_params, rays_out, B, h, w, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_make_rays launch");
}

MAGNET_API int magnet_relative_poses(const double* ext_ref, const double* ext_nghbr, float* poses_out, int32_t* is_valid_out,
                                     int32_t B, int32_t V, void* stream) {
    if (!ext_ref || !ext_nghbr || !poses_out || !is_valid_out) return fail(MAGNET_E_NULL, "magnet_relative_poses: NULL pointer");
    if (B <= 0 || V <= 0) return fail(MAGNET_E_DIM, "magnet_relative_poses: bad dims B=%d V=%d", B, V);
    hipError_t e = magnet::launch_relative_poses(ext_ref, ext_nghbr, poses_out, is_valid_out, B, V, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_relative_poses launch");
}

MAGNET_API int64_t magnet_cost_volume_f_backward_workspace(const MagnetCostVolumeArgs* a) {
    magnet::CvParams p;
    if (cv_prepare(a, p, true)) return -1;
    return (int64_t)magnet::cvf_gather_workspace_bytes(p);
}

MAGNET_API int magnet_cost_volume_f_backward_ws(const MagnetCostVolumeArgs* a, const float* grad_cost, float* grad_ref_cl,
                                                float* grad_src_pad, void* workspace, int64_t workspace_bytes, void* stream) {
    magnet::CvParams p;
    if (const int rc = cv_prepare(a, p, true)) return rc;
    if (a->mode != 1) return fail(MAGNET_E_DIM, "magnet_cost_volume_f_backward_ws: only mode 1 (est_costvolume_F) is differentiable");
    if (a->feat_dtype != MAGNET_FEAT_F32) return fail(MAGNET_E_DTYPE, "magnet_cost_volume_f_backward_ws: fp32 features required");
    if (!grad_cost || !grad_ref_cl || !grad_src_pad) return fail(MAGNET_E_NULL, "magnet_cost_volume_f_backward_ws: NULL gradient pointer");
    if (!aligned16(grad_ref_cl) || !aligned16(grad_src_pad))
        return fail(MAGNET_E_ALIGN, "magnet_cost_volume_f_backward_ws: gradient buffers must be 16-byte aligned");
    bool ref_ok = false, src_ok = false;
    hipError_t e = magnet::launch_cvf_bwd_ref_only(p, grad_cost, grad_ref_cl, (hipStream_t)stream, &ref_ok);
    if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_f_backward_ws grad_ref launch");
    if (ref_ok && !(CV_DEV(p) & 0x30)) {
        e = magnet::launch_cvf_gather_src(p, grad_cost, grad_src_pad, workspace, workspace_bytes < 0 ? 0 : (size_t)workspace_bytes,
                                          (hipStream_t)stream, &src_ok);
        if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_f_backward_ws grad_src launch");
    }
    if (ref_ok && src_ok) return 0;
    // shapes / workspace the gather path does not take: the scatter kernels (grad_src_pad must have been zeroed by the caller)
    bool handled = false;
    e = magnet::launch_cvf_bwd(p, grad_cost, grad_ref_cl, grad_src_pad, (hipStream_t)stream, &handled);
    if (e != hipSuccess) return hip_fail(e, "magnet_cost_volume_f_backward_ws launch");
    if (!handled) return fail(MAGNET_E_DIM, "magnet_cost_volume_f_backward_ws: shape not supported (F > 128, V > 31 or image too large)");
    return 0;
}

MAGNET_API int magnet_gaussian_update(const float* gnet_out, const float* gmm_in, float* gmm_out, int32_t B, int32_t hw,
                           void* stream) {
    if (!gnet_out || !gmm_in || !gmm_out) return fail(MAGNET_E_NULL, "magnet_gaussian_update: NULL pointer");
    if (B <= 0 || hw <= 0) return fail(MAGNET_E_DIM, "magnet_gaussian_update: bad dims B=%d hw=%d", B, hw);
    hipError_t e = magnet::launch_gaussian_update(gnet_out, gmm_in, gmm_out, B, hw, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_gaussian_update launch");
}

MAGNET_API int magnet_upsample_depth(const float* depth, const float* mask, float* out, int32_t B, int32_t C, int32_t h,
                          int32_t w, int32_t k, void* stream) {
    if (!depth || !mask || !out) return fail(MAGNET_E_NULL, "magnet_upsample_depth: NULL pointer");
    if (B <= 0 || h <= 0 || w <= 0 || (C != 1 && C != 2) || (k != 1 && k != 2 && k != 4 && k != 8))
        return fail(MAGNET_E_DIM, "magnet_upsample_depth: bad dims B=%d C=%d h=%d w=%d k=%d (C in {1,2}, k in {1,2,4,8})", B, C, h, w, k);
    if (k == 4 && !aligned16(out)) return fail(MAGNET_E_ALIGN, "magnet_upsample_depth: out not 16-byte aligned");
    hipError_t e = magnet::launch_upsample(depth, mask, out, B, C, h, w, k, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_upsample_depth launch");
}

MAGNET_API int magnet_conv_mfma(const MagnetConvArgs* a, void* stream) {
    if (!a) return fail(MAGNET_E_NULL, "magnet_conv_mfma: args is NULL");
    if (!a->in_hi || !a->in_lo || !a->w_hi || !a->w_lo || !a->bias) return fail(MAGNET_E_NULL, "magnet_conv_mfma: NULL input pointer");
    if (a->out_mode < 0 || a->out_mode > 2) return fail(MAGNET_E_DIM, "magnet_conv_mfma: out_mode must be 0, 1 or 2");
    const bool tail = a->tail_w_hi != nullptr;
    if (tail) {
        if (!a->tail_w_lo || !a->tail_bias || (!a->out_f32 && !a->up_out && !a->gu_out)) return fail(MAGNET_E_NULL, "magnet_conv_mfma: fused tail needs tail_w_lo, tail_bias and out_f32 (or up_out / gu_out)");
        if (a->cout_pad != 128 || (a->tail_cout_pad != 16 && a->tail_cout_pad != 128 && a->tail_cout_pad != 144))
            return fail(MAGNET_E_DIM, "magnet_conv_mfma: fused tail needs cout_pad == 128 and tail_cout_pad in {16,128,144}");
        if (!aligned16(a->tail_w_hi) || !aligned16(a->tail_w_lo) || !aligned16(a->tail_bias) || !aligned16(a->out_f32))
            return fail(MAGNET_E_ALIGN, "magnet_conv_mfma: fused tail pointers must be 16-byte aligned");
        if (a->border_hp || a->repad || a->add_hi) return fail(MAGNET_E_DIM, "magnet_conv_mfma: fused tail excludes border / repad / residual input");
    }
    if (!tail && (a->out_mode == 0 ? (!a->out_hi || !a->out_lo) : (a->out_mode == 1 ? !a->out_f32 : !a->out_hi)))
        return fail(MAGNET_E_NULL, "magnet_conv_mfma: NULL output pointer");
    if (a->rows <= 0 || a->cin <= 0 || (a->cin % 32) != 0) return fail(MAGNET_E_DIM, "magnet_conv_mfma: rows > 0 and cin %% 32 == 0 required (cin=%d)", a->cin);
    if (a->taps != 1 && a->taps != 4 && a->taps != 9) return fail(MAGNET_E_DIM, "magnet_conv_mfma: taps must be 1, 4 or 9");
    if (a->taps != 1 && a->wp < 3) return fail(MAGNET_E_DIM, "magnet_conv_mfma: wp (row pitch of the bordered grid) missing");
    if (!((a->cout_pad > 0 && a->cout_pad % 128 == 0) || a->cout_pad == 144 || a->cout_pad == 16 || a->cout_pad == 32 || a->cout_pad == 64))
        return fail(MAGNET_E_DIM, "magnet_conv_mfma: cout_pad=%d unsupported (multiple of 128, or 144, 64, 32, 16)", a->cout_pad);
    if (!aligned16(a->in_hi) || !aligned16(a->in_lo) || !aligned16(a->w_hi) || !aligned16(a->w_lo) || !aligned16(a->bias) ||
        (!tail && (a->out_mode == 0 ? (!aligned16(a->out_hi) || !aligned16(a->out_lo)) : (a->out_mode == 1 ? !aligned16(a->out_f32) : !aligned16(a->out_hi)))))
        return fail(MAGNET_E_ALIGN, "magnet_conv_mfma: pointers must be 16-byte aligned");
    if (a->dil < 0 || a->dil > 8) return fail(MAGNET_E_DIM, "magnet_conv_mfma: dil=%d out of range", a->dil);
    magnet::ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in_hi = (const uint16_t*)a->in_hi; p.in_lo = (const uint16_t*)a->in_lo;
    p.w_hi = (const uint16_t*)a->w_hi; p.w_lo = (const uint16_t*)a->w_lo; p.bias = a->bias;
    p.out_hi = (uint16_t*)a->out_hi; p.out_lo = (uint16_t*)a->out_lo; p.out_f32 = a->out_f32;
    p.rows = a->rows; p.cin = a->cin; p.cout_pad = a->cout_pad; p.taps = a->taps; p.wp = a->wp;
    p.relu = a->relu; p.out_mode = a->out_mode;
    p.in_ld = a->in_ld ? a->in_ld : a->cin;
    p.addend = a->addend; p.addend_ld = a->addend_ld ? a->addend_ld : a->cout_pad;
    if (a->addend && (!aligned16(a->addend) || (p.addend_ld % 4) != 0 || p.addend_ld < a->cout_pad))
        return fail(MAGNET_E_ALIGN, "magnet_conv_mfma: addend must be 16-byte aligned with addend_ld >= cout_pad, a multiple of 4");
    if (p.in_ld < a->cin || (p.in_ld % 8) != 0) return fail(MAGNET_E_DIM, "magnet_conv_mfma: in_ld=%d must be >= cin and a multiple of 8", p.in_ld);
    const int dil = a->dil > 1 ? a->dil : 1;
    if (a->taps == 9) for (int t = 0; t < 9; ++t) p.tap_off[t] = ((t / 3 - 1) * a->wp + (t % 3 - 1)) * dil;
    if (a->taps == 4) { p.tap_off[0] = -a->wp - 1; p.tap_off[1] = -a->wp; p.tap_off[2] = -1; p.tap_off[3] = 0; }
    p.tap_n = a->taps == 9 ? 3 : (a->taps == 4 ? 2 : 1);
    p.tap_o0 = a->taps == 1 ? 0 : -1;
    p.tap_sy = a->taps == 9 ? a->wp * dil : (a->taps == 4 ? a->wp : 0);
    p.tap_sx = a->taps == 9 ? dil : (a->taps == 4 ? 1 : 0);
    for (int t = 0; t < a->taps; ++t) { p.min_off = p.tap_off[t] < p.min_off ? p.tap_off[t] : p.min_off; p.max_off = p.tap_off[t] > p.max_off ? p.tap_off[t] : p.max_off; }
    if (((long long)128 + p.max_off - p.min_off + 256) * (long long)p.in_ld * 2 >= ((long long)1 << 31) || (long long)a->taps * a->cout_pad * a->cin * 2 >= ((long long)1 << 31))
        return fail(MAGNET_E_DIM, "magnet_conv_mfma: row window or weight tensor exceeds 2 GiB");
    p.out_ld = a->out_ld ? a->out_ld : a->cout_pad;
    if (p.out_ld < a->cout_pad || (p.out_ld % 8) != 0) return fail(MAGNET_E_DIM, "magnet_conv_mfma: out_ld=%d must be >= cout_pad and a multiple of 8", p.out_ld);
    p.add_hi = (const uint16_t*)a->add_hi; p.add_lo = (const uint16_t*)a->add_lo; p.add_ld = a->add_ld ? a->add_ld : a->cout_pad;
    if ((a->add_hi != nullptr) != (a->add_lo != nullptr)) return fail(MAGNET_E_NULL, "magnet_conv_mfma: add_hi and add_lo come together");
    if (a->add_hi && (!aligned16(a->add_hi) || !aligned16(a->add_lo) || (p.add_ld % 8) != 0 || p.add_ld < a->cout_pad))
        return fail(MAGNET_E_ALIGN, "magnet_conv_mfma: add_hi/add_lo must be 16-byte aligned with add_ld >= cout_pad, a multiple of 8");
    if (a->border_hp) {
        if (a->wp < 1 || a->border_pad < 0 || a->border_hp <= 2 * a->border_pad || a->wp <= 2 * a->border_pad ||
            (a->rows % ((long long)a->border_hp * a->wp)) != 0)
            return fail(MAGNET_E_DIM, "magnet_conv_mfma: border_hp=%d wp=%d border_pad=%d do not tile rows=%lld", a->border_hp, a->wp, a->border_pad, (long long)a->rows);
        if ((long long)a->border_hp * a->wp >= (1 << 24)) return fail(MAGNET_E_DIM, "magnet_conv_mfma: border_hp * wp must be < 2^24");
        p.img_rows = a->border_hp * a->wp; p.hp = a->border_hp; p.pad = a->border_pad;
    }
    if (a->repad < 0 || (a->repad && !a->border_hp)) return fail(MAGNET_E_DIM, "magnet_conv_mfma: repad needs border_hp");
    p.repad = a->repad;
    p.tail_w_hi = (const uint16_t*)a->tail_w_hi; p.tail_w_lo = (const uint16_t*)a->tail_w_lo; p.tail_bias = a->tail_bias;
    p.tail_cout = a->tail_cout_pad;
    if (a->in_sc || a->w_sc) {                                       // v302: fp16 + block-scaled e4m3 operand format
        if (!a->in_sc || !a->w_sc) return fail(MAGNET_E_NULL, "magnet_conv_mfma: in_sc and w_sc come together");
        if (!tail || a->taps != 9 || a->dil > 1 || a->addend || a->rows < 256ll * 256 || a->sc_rows < a->rows || ((uintptr_t)a->in_sc & 3) || ((uintptr_t)a->w_sc & 3) ||
            ((int64_t)(a->cin / 32) * a->sc_rows * 4 >= ((int64_t)1 << 31)))
            return fail(MAGNET_E_SHAPE, "magnet_conv_mfma: the fp16 + e4m3 operand format serves the 3x3 layers with a fused tail, rows >= 65536, no addend");
        p.in_sc = (const uint32_t*)a->in_sc; p.w_sc = (const uint32_t*)a->w_sc; p.sc_rows = a->sc_rows;
    }
    if (a->up_out || a->up_depth) {                                  // fused convex upsampling in the tail's last layer
        if (!a->up_out || !a->up_depth) return fail(MAGNET_E_NULL, "magnet_conv_mfma: up_depth and up_out come together");
        if (!tail || a->tail_cout_pad != 144) return fail(MAGNET_E_DIM, "magnet_conv_mfma: the fused upsampling needs the 144-channel fused tail");
        if (a->up_npred <= 0 || a->up_B <= 0 || a->up_h <= 0 || a->up_w <= 0 || a->wp != a->up_w + 2 ||
            a->rows != (long long)a->up_B * (a->up_h + 2) * (a->up_w + 2) || a->rows >= ((long long)1 << 31))
            return fail(MAGNET_E_DIM, "magnet_conv_mfma: up_B=%d up_h=%d up_w=%d do not describe rows=%lld, wp=%d", a->up_B, a->up_h, a->up_w,
                        (long long)a->rows, a->wp);
        if (!aligned16(a->up_out) || ((uintptr_t)a->up_depth & 3)) return fail(MAGNET_E_ALIGN, "magnet_conv_mfma: up_out must be 16-byte aligned");
        p.up_depth = a->up_depth; p.up_out = a->up_out; p.up_npred = a->up_npred; p.up_B = a->up_B; p.up_h = a->up_h; p.up_w = a->up_w;
    }
    if (a->gu_out || a->gu_in) {                                     // fused Gaussian update behind G-Net's head
        if (!a->gu_out || !a->gu_in || a->gu_in == a->gu_out) return fail(MAGNET_E_NULL, "magnet_conv_mfma: gu_in and gu_out come together and must not alias");
        if (!tail || a->tail_cout_pad != 16 || a->up_out) return fail(MAGNET_E_DIM, "magnet_conv_mfma: the fused Gaussian update needs the 16-channel fused tail");
        if (a->up_B <= 0 || a->up_h <= 0 || a->up_w <= 0 || a->wp != a->up_w + 2 || a->rows != (long long)a->up_B * (a->up_h + 2) * (a->up_w + 2) ||
            a->rows >= ((long long)1 << 31))
            return fail(MAGNET_E_DIM, "magnet_conv_mfma: up_B=%d up_h=%d up_w=%d do not describe rows=%lld, wp=%d", a->up_B, a->up_h, a->up_w,
                        (long long)a->rows, a->wp);
        p.gu_in = a->gu_in; p.gu_out = a->gu_out; p.up_B = a->up_B; p.up_h = a->up_h; p.up_w = a->up_w;
    }
#ifdef MAGNET_DEV
    { static const int dev_variant = getenv("MAGNET_CONV_VARIANT") ? atoi(getenv("MAGNET_CONV_VARIANT")) : 0; p.variant = dev_variant; }   // dev A/B switch (dev build only)
#endif
    hipError_t e = magnet::launch_conv_mfma(p, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_conv_mfma launch");
}

MAGNET_API int magnet_fnet_stem(const float* img, const float* wgt, const float* bias, void* out_hi, void* out_lo, int32_t N,
                                int32_t H, int32_t W, void* stream) {
    if (!img || !wgt || !bias || !out_hi || !out_lo) return fail(MAGNET_E_NULL, "magnet_fnet_stem: NULL pointer");
    if (N <= 0 || H < 2 || W < 2) return fail(MAGNET_E_DIM, "magnet_fnet_stem: bad dims N=%d H=%d W=%d", N, H, W);
    if (!aligned16(out_hi) || !aligned16(out_lo)) return fail(MAGNET_E_ALIGN, "magnet_fnet_stem: outputs must be 16-byte aligned");
    hipError_t e = magnet::launch_fnet_stem(img, wgt, bias, (uint16_t*)out_hi, (uint16_t*)out_lo, N, H, W, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_fnet_stem launch");
}

MAGNET_API int magnet_space_to_depth(const void* in_hi, const void* in_lo, void* out_hi, void* out_lo, int32_t N, int32_t C,
                                     int32_t H2, int32_t W2, int32_t opad, void* stream) {
    if (!in_hi || !in_lo || !out_hi || !out_lo) return fail(MAGNET_E_NULL, "magnet_space_to_depth: NULL pointer");
    if (N <= 0 || C <= 0 || (C % 8) != 0 || H2 <= 0 || W2 <= 0 || opad < 0)
        return fail(MAGNET_E_DIM, "magnet_space_to_depth: bad dims N=%d C=%d H2=%d W2=%d opad=%d", N, C, H2, W2, opad);
    if (!aligned16(in_hi) || !aligned16(in_lo) || !aligned16(out_hi) || !aligned16(out_lo))
        return fail(MAGNET_E_ALIGN, "magnet_space_to_depth: pointers must be 16-byte aligned");
    hipError_t e = magnet::launch_space_to_depth((const uint16_t*)in_hi, (const uint16_t*)in_lo, (uint16_t*)out_hi, (uint16_t*)out_lo,
                                                 N, C, H2, W2, opad, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_space_to_depth launch");
}

MAGNET_API int magnet_avgpool_cl(const void* in_hi, const void* in_lo, int32_t ld, int32_t N, int32_t h, int32_t w, int32_t pad,
                                 int32_t k, int32_t C, void* out_hi, void* out_lo, void* stream) {
    if (!in_hi || !in_lo || !out_hi || !out_lo) return fail(MAGNET_E_NULL, "magnet_avgpool_cl: NULL pointer");
    if (N <= 0 || C <= 0 || C > 128 || (C % 8) != 0 || ld < C || (ld % 8) != 0 || k <= 0 || h < k || w < k || pad < 0)
        return fail(MAGNET_E_DIM, "magnet_avgpool_cl: bad dims N=%d C=%d ld=%d h=%d w=%d k=%d pad=%d", N, C, ld, h, w, k, pad);
    if (!aligned16(in_hi) || !aligned16(in_lo) || !aligned16(out_hi) || !aligned16(out_lo))
        return fail(MAGNET_E_ALIGN, "magnet_avgpool_cl: pointers must be 16-byte aligned");
    hipError_t e = magnet::launch_avgpool_cl((const uint16_t*)in_hi, (const uint16_t*)in_lo, ld, N, h, w, pad, k, C, (uint16_t*)out_hi,
                                             (uint16_t*)out_lo, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_avgpool_cl launch");
}

MAGNET_API int magnet_upsample_bilinear_cl(const float* in, int32_t in_ld, int32_t ph, int32_t pw, int32_t C, void* out_hi,
                                           void* out_lo, int32_t out_ld, int32_t N, int32_t h, int32_t w, int32_t pad, void* stream) {
    if (!in || !out_hi || !out_lo) return fail(MAGNET_E_NULL, "magnet_upsample_bilinear_cl: NULL pointer");
    if (N <= 0 || C <= 0 || (C % 8) != 0 || in_ld < C || out_ld < C || (out_ld % 8) != 0 || ph <= 0 || pw <= 0 || h <= 0 || w <= 0 || pad < 0)
        return fail(MAGNET_E_DIM, "magnet_upsample_bilinear_cl: bad dims");
    if (!aligned16(out_hi) || !aligned16(out_lo)) return fail(MAGNET_E_ALIGN, "magnet_upsample_bilinear_cl: outputs must be 16-byte aligned");
    hipError_t e = magnet::launch_upsample_bilinear_cl(in, in_ld, ph, pw, C, (uint16_t*)out_hi, (uint16_t*)out_lo, out_ld, N, h, w, pad,
                                                       (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_upsample_bilinear_cl launch");
}

MAGNET_API int magnet_conv1x1_chain(const void* in_hi, const void* in_lo, const void* w_hi, const void* w_lo, const float* bias,
                                    float* out, int64_t rows, int32_t cout_pad, void* stream) {
    if (!in_hi || !in_lo || !w_hi || !w_lo || !bias || !out) return fail(MAGNET_E_NULL, "magnet_conv1x1_chain: NULL pointer");
    if (rows <= 0 || (cout_pad != 16 && cout_pad != 144 && cout_pad != 128))
        return fail(MAGNET_E_DIM, "magnet_conv1x1_chain: rows > 0 and cout_pad in {16, 128, 144} required (got %d)", cout_pad);
    if (!aligned16(in_hi) || !aligned16(in_lo) || !aligned16(w_hi) || !aligned16(w_lo) || !aligned16(out) || !aligned16(bias))
        return fail(MAGNET_E_ALIGN, "magnet_conv1x1_chain: pointers must be 16-byte aligned");
    magnet::ChainParams p;
    p.in_hi = (const uint16_t*)in_hi; p.in_lo = (const uint16_t*)in_lo; p.w_hi = (const uint16_t*)w_hi; p.w_lo = (const uint16_t*)w_lo;
    p.bias = bias; p.out = out; p.rows = rows; p.cout_pad = cout_pad;
    hipError_t e = magnet::launch_conv1x1_chain(p, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_conv1x1_chain launch");
}

MAGNET_API int magnet_pack_split(const float* nchw, void* out_hi, void* out_lo, int32_t N, int32_t C, int32_t h, int32_t w,
                                 int32_t ctot, int32_t c_off, int64_t in_img_stride, void* stream) {
    if (!nchw || !out_hi || !out_lo) return fail(MAGNET_E_NULL, "magnet_pack_split: NULL pointer");
    if (N <= 0 || C <= 0 || h <= 0 || w <= 0 || (c_off % 8) || c_off < 0 || c_off + ((C + 7) / 8) * 8 > ctot || (ctot % 8))
        return fail(MAGNET_E_DIM, "magnet_pack_split: bad dims N=%d C=%d h=%d w=%d ctot=%d c_off=%d", N, C, h, w, ctot, c_off);
    if (!aligned16(out_hi) || !aligned16(out_lo)) return fail(MAGNET_E_ALIGN, "magnet_pack_split: outputs not 16-byte aligned");
    hipError_t e = magnet::launch_pack_split(nchw, (uint16_t*)out_hi, (uint16_t*)out_lo, N, C, h, w, ctot, c_off,
                                             in_img_stride ? in_img_stride : (long long)C * h * w, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_pack_split launch");
}

MAGNET_API int magnet_pack_mx(const float* nchw, void* out_f16, void* out_qr, void* out_sc, int32_t N, int32_t C, int32_t h, int32_t w,
                              int32_t ctot, int32_t c_off, int64_t sc_rows, int64_t in_img_stride, void* stream) {
    if (!nchw || !out_f16 || !out_qr || !out_sc) return fail(MAGNET_E_NULL, "magnet_pack_mx: NULL pointer");
    if (N <= 0 || C <= 0 || h <= 0 || w <= 0 || (C % 32) || (c_off % 32) || c_off < 0 || c_off + C > ctot || (ctot % 32) || ((h * w) % 4))
        return fail(MAGNET_E_DIM, "magnet_pack_mx: bad dims N=%d C=%d h=%d w=%d ctot=%d c_off=%d (C, c_off, ctot multiples of 32; h*w of 4)", N, C, h, w, ctot, c_off);
    if (sc_rows < (int64_t)N * (h + 2) * (w + 2)) return fail(MAGNET_E_DIM, "magnet_pack_mx: sc_rows smaller than N*(h+2)*(w+2)");
    const int64_t stride = in_img_stride ? in_img_stride : (int64_t)C * h * w;
    if (!aligned16(out_f16) || !aligned16(out_qr) || !aligned16(nchw) || (stride % 4)) return fail(MAGNET_E_ALIGN, "magnet_pack_mx: pointers not 16-byte aligned");
    hipError_t e = magnet::launch_pack_mx(nchw, (uint16_t*)out_f16, (uint8_t*)out_qr, (uint32_t*)out_sc, N, C, h, w, ctot, c_off, sc_rows, stride,
                                          (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_pack_mx launch");
}

MAGNET_API int magnet_gaussian_update_cl(const float* gnet_out_pad, int32_t ld, const float* gmm_in, float* gmm_out,
                                         int32_t B, int32_t h, int32_t w, void* stream) {
    if (!gnet_out_pad || !gmm_in || !gmm_out) return fail(MAGNET_E_NULL, "magnet_gaussian_update_cl: NULL pointer");
    if (B <= 0 || h <= 0 || w <= 0 || ld < 2) return fail(MAGNET_E_DIM, "magnet_gaussian_update_cl: bad dims");
    hipError_t e = magnet::launch_gaussian_update_cl(gnet_out_pad, ld, gmm_in, gmm_out, B, h, w, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_gaussian_update_cl launch");
}

MAGNET_API int magnet_upsample_depth_cl(const float* depth, const float* mask_pad, int32_t ld, float* out, int32_t B,
                                        int32_t h, int32_t w, void* stream) {
    if (!depth || !mask_pad || !out) return fail(MAGNET_E_NULL, "magnet_upsample_depth_cl: NULL pointer");
    if (B <= 0 || h <= 0 || w <= 0 || ld < 144 || (ld % 4)) return fail(MAGNET_E_DIM, "magnet_upsample_depth_cl: bad dims (ld >= 144, ld %% 4 == 0)");
    if (!aligned16(out) || !aligned16(mask_pad)) return fail(MAGNET_E_ALIGN, "magnet_upsample_depth_cl: pointers not 16-byte aligned");
    hipError_t e = magnet::launch_upsample_cl(depth, mask_pad, ld, out, B, h, w, 1, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_upsample_depth_cl launch");
}

MAGNET_API int magnet_upsample_depth_cl_n(const float* depths, const float* mask_pad, int32_t ld, float* outs, int32_t n_pred, int32_t B,
                                          int32_t h, int32_t w, void* stream) {
    if (!depths || !mask_pad || !outs) return fail(MAGNET_E_NULL, "magnet_upsample_depth_cl_n: NULL pointer");
    if (n_pred <= 0 || n_pred > 64 || B <= 0 || h <= 0 || w <= 0 || ld < 144 || (ld % 4))
        return fail(MAGNET_E_DIM, "magnet_upsample_depth_cl_n: bad dims (1 <= n_pred <= 64, ld >= 144, ld %% 4 == 0)");
    if (!aligned16(outs) || !aligned16(mask_pad)) return fail(MAGNET_E_ALIGN, "magnet_upsample_depth_cl_n: pointers not 16-byte aligned");
    hipError_t e = magnet::launch_upsample_cl(depths, mask_pad, ld, outs, B, h, w, n_pred, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_upsample_depth_cl_n launch");
}

MAGNET_API int magnet_depth_metrics(const float* pred, const float* gt, double* sums, int32_t B, int32_t HW, float min_depth,
                                    float max_depth, void* stream) {
    if (!pred || !gt || !sums) return fail(MAGNET_E_NULL, "magnet_depth_metrics: NULL pointer");
    if (B <= 0 || HW <= 0 || !(max_depth > min_depth)) return fail(MAGNET_E_DIM, "magnet_depth_metrics: bad arguments");
    hipError_t e = magnet::launch_depth_metrics(pred, gt, sums, B, HW, HW, min_depth, max_depth, 0, -1, 0, 0, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_depth_metrics launch");
}

MAGNET_API int magnet_depth_metrics_crop(const float* pred, const float* gt, double* sums, int32_t B, int32_t H, int32_t W,
                                         float min_depth, float max_depth, int32_t y0, int32_t y1, int32_t x0, int32_t x1, void* stream) {
    if (!pred || !gt || !sums) return fail(MAGNET_E_NULL, "magnet_depth_metrics_crop: NULL pointer");
    if (B <= 0 || H <= 0 || W <= 0 || !(max_depth > min_depth) || y0 < 0 || x0 < 0 || y1 > H || x1 > W || y1 < y0 || x1 < x0)
        return fail(MAGNET_E_DIM, "magnet_depth_metrics_crop: bad arguments");
    hipError_t e = magnet::launch_depth_metrics(pred, gt, sums, B, H * W, W, min_depth, max_depth, y0, y1, x0, x1, (hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "magnet_depth_metrics_crop launch");
}

}  // extern "C"
