/*
 * oracle/cost_volume_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, per-pixel closed-form restatement of the reference's consistency-weighted
 * multi-view matching (reference: models/submodules/homography.py:79-161 `est_costvolume_CW`
 * + `_compute_cost_CW`, candidate sampling models/MAGNET.py:153-156).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (magnet_amd/csrc) never calls it.
 *
 * Parity pin: validated in the build container against the imported reference
 * (tests/golden/make_golden.py) and on every run against the committed golden vectors
 * (the .npz files under tests/golden/, checked by tests/test_oracle_golden.py).
 *
 * Arithmetic contract (each step cites the reference line it restates); everything is IEEE
 * fp32 with separate multiply and add (build with -ffp-contract=off), because the reference
 * runs one ATen op per arithmetic operation — except inside sgemm and grid_sampler, whose
 * CPU builds contract to FMA; those spots use explicit fmaf below:
 *
 *   d        = mu + sigma*(float)k_j                    MAGNET.py:155   (mul, then add)
 *   KR       = K*R                                      homography.py:102 (sgemm: a0b0, then 2 fma)
 *   Kt       = K*t                                      homography.py:101 (sgemv: plain l-to-r, unfused)
 *   r_pix    = KR*ray ; r_cam_z = (R*ray)_z             homography.py:100,102
 *   P        = t_pix + r_pix*d                          homography.py:132
 *   P       /= (P_z + 1e-10)                            homography.py:133   (no behind-camera test)
 *   z_warp   = t_z + r_cam_z*d                          homography.py:137-138
 *   g        = (P_xy - c)/c, clamp to [-10,10]          homography.py:141-148  (c = w/2, h/2)
 *   i        = fma(g + 1, size/2, -0.5)                 ATen grid_sampler unnormalize, align_corners=False
 *   bilinear, zeros padding: weights (x1-ix)(y1-iy) ...  homography.py:150-152 (nw*v, then fma ne,sw,se)
 *   c        = sum_f ref[f]*warp[f]                     homography.py:155 (ATen cascade sum, 16-chunks)
 *   gate     = |z_warp - mu_w| < sigma_w*kappa          homography.py:157-158  (strict <)
 *   cost     = (float)( sum_{valid v} (double)c*gate ) / (float)V     homography.py:116-120,159
 *
 * Layouts are the reference's: NCHW fp32, source tensors view-major (index v*B+b,
 * homography.py:105), rays (B,3,h*w), intM (B,3,3), poses (B,V,4,4), is_valid (B,V) int32.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

static inline float clampf(float x, float lo, float hi) {
    /* reference: src_coords[src_coords > 10] = 10; src_coords[src_coords < -10] = -10 (NaN stays NaN) */
    if (x > hi) x = hi;
    if (x < lo) x = lo;
    return x;
}

typedef struct {
    float x0f, y0f;       /* floor(ix), floor(iy) as floats */
    float nw, ne, sw, se; /* bilinear weights in ATen's order */
    int   x0, y0;         /* integer tap origin (valid only when finite) */
    int   finite;
} taps_t;

static inline void make_taps(float ix, float iy, taps_t *t) {
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    t->x0f = x0; t->y0f = y0;
    t->nw = (x1 - ix) * (y1 - iy);
    t->ne = (ix - x0) * (y1 - iy);
    t->sw = (x1 - ix) * (iy - y0);
    t->se = (ix - x0) * (iy - y0);
    /* |coordinate| is bounded by the +-10 clamp unless NaN */
    t->finite = (ix == ix) && (iy == iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
    t->x0 = t->finite ? (int)x0 : -100000;
    t->y0 = t->finite ? (int)y0 : -100000;
}

static inline float tap(const float *img, int h, int w, int y, int x) {
    return (y >= 0 && y < h && x >= 0 && x < w) ? img[(size_t)y * w + x] : 0.0f;
}

/* ATen's CPU grid_sampler is built with FMA contraction: nw_val*nw, then three fused
 * accumulations in the order ne, sw, se (verified bitwise against torch 2.10 CPU). */
static inline float bilinear(const float *img, int h, int w, const taps_t *t) {
    float v = tap(img, h, w, t->y0, t->x0) * t->nw;
    v = __builtin_fmaf(tap(img, h, w, t->y0, t->x0 + 1), t->ne, v);
    v = __builtin_fmaf(tap(img, h, w, t->y0 + 1, t->x0), t->sw, v);
    v = __builtin_fmaf(tap(img, h, w, t->y0 + 1, t->x0 + 1), t->se, v);
    return v;
}

/* 3-term dot product as the reference's BLAS sgemm rounds it: a0*b0, then two fused
 * accumulations (verified bitwise against torch 2.10 CPU matmul for K*R, K*t, (K*R)*ray, R*ray). */
static inline float dot3(const float *a, float b0, float b1, float b2) {
    return __builtin_fmaf(a[2], b2, __builtin_fmaf(a[1], b1, a[0] * b0));
}

/* Matrix-VECTOR products (IntM.matmul(t), homography.py:27,101) go through sgemv, which rounds differently from
 * sgemm: plain products summed left to right, no fusion (verified bitwise against torch 2.10 CPU on 3000 random K, t). */
static inline float dot3_gemv(const float *a, float b0, float b1, float b2) {
    return (a[0] * b0 + a[1] * b1) + a[2] * b2;
}

/*
 * d_volume: (B,D,h,w) candidate depths, or NULL to sample them in place from ref_gmm + k_list.
 * gates:    optional (B,V,D,h,w) uint8 output of the consistency gate bits (0 for invalid views).
 * featcost: optional (B,V,D,h,w) fp32 output of the un-gated per-view feature dot products.
 * n_threads <= 0 -> all cores.
 */
ORACLE_API int magnet_oracle_cost_volume_cw(
    const float *d_volume, const float *ref_gmm, const double *k_list,
    const float *ref_feat, const float *src_feat, const float *src_gmm,
    const float *poses, const int32_t *is_valid, const float *intM, const float *rays,
    int B, int V, int F, int D, int h, int w, float kappa,
    float *out, uint8_t *gates, float *featcost, int n_threads)
{
    const size_t hw = (size_t)h * w;
    const float cw = (float)((double)w / 2.0), ch = (float)((double)h / 2.0);
    const float sw_ = (float)w / 2.0f, sh_ = (float)h / 2.0f; /* ATen: scaling_factor = size/2 */
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    #pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        for (int y = 0; y < h; ++y) {
            const float *K = intM + (size_t)b * 9;
            for (int x = 0; x < w; ++x) {
                const size_t p = (size_t)y * w + x;
                const float ray0 = rays[((size_t)b * 3 + 0) * hw + p];
                const float ray1 = rays[((size_t)b * 3 + 1) * hw + p];
                const float ray2 = rays[((size_t)b * 3 + 2) * hw + p];
                const float mu = ref_gmm ? ref_gmm[((size_t)b * 2 + 0) * hw + p] : 0.f;
                const float sg = ref_gmm ? ref_gmm[((size_t)b * 2 + 1) * hw + p] : 0.f;
                for (int j = 0; j < D; ++j) {
                    float d;
                    if (d_volume) d = d_volume[((size_t)b * D + j) * hw + p];
                    else { float sk = sg * (float)k_list[j]; d = mu + sk; }
                    double acc = 0.0;
                    for (int v = 0; v < V; ++v) {
                        const size_t bvi = (((size_t)b * V + v) * D + j) * hw + p;
                        if (is_valid[b * V + v] != 1) {
                            if (gates) gates[bvi] = 0;
                            if (featcost) featcost[bvi] = 0.f;
                            continue;
                        }
                        const float *T = poses + ((size_t)b * V + v) * 16;
                        float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
                        float t[3] = {T[3], T[7], T[11]};
                        float KR[9], Kt[3];
                        for (int i = 0; i < 3; ++i) {
                            for (int c = 0; c < 3; ++c)
                                KR[i * 3 + c] = dot3(K + i * 3, R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]);
                            Kt[i] = dot3_gemv(K + i * 3, t[0], t[1], t[2]);
                        }
                        const float rpx = dot3(KR + 0, ray0, ray1, ray2);
                        const float rpy = dot3(KR + 3, ray0, ray1, ray2);
                        const float rpz = dot3(KR + 6, ray0, ray1, ray2);
                        const float rcz = dot3(R + 6, ray0, ray1, ray2);
                        float Px = Kt[0] + rpx * d;
                        float Py = Kt[1] + rpy * d;
                        float Pz = Kt[2] + rpz * d;
                        const float zz = Pz + 1e-10f;
                        Px = Px / zz; Py = Py / zz;
                        const float zw = t[2] + rcz * d;
                        float gx = (Px - cw) / cw, gy = (Py - ch) / ch;
                        gx = clampf(gx, -10.f, 10.f); gy = clampf(gy, -10.f, 10.f);
                        const float ix = __builtin_fmaf(gx + 1.0f, sw_, -0.5f);
                        const float iy = __builtin_fmaf(gy + 1.0f, sh_, -0.5f);
                        taps_t tp; make_taps(ix, iy, &tp);
                        const size_t sidx = (size_t)v * B + b;     /* view-major */
                        const float *sf = src_feat + sidx * F * hw;
                        const float *rf = ref_feat + (size_t)b * F * hw + p;
                        float c = 0.f;
                        if (tp.finite && tp.x0 >= -1 && tp.x0 < w && tp.y0 >= -1 && tp.y0 < h) {
                            /* torch.sum(ref*warp, axis=1): fp32 products, ATen cascade sum —
                             * sequential inside 16-element chunks, chunk totals added in order */
                            float lvl0 = 0.f, lvl1 = 0.f, lvl2 = 0.f;
                            for (int f = 0; f < F; ++f) {
                                float wv = bilinear(sf + (size_t)f * hw, h, w, &tp);
                                float pr = rf[(size_t)f * hw] * wv;
                                lvl0 = lvl0 + pr;
                                if ((f & 15) == 15) { lvl1 = lvl1 + lvl0; lvl0 = 0.f;
                                    if ((f & 255) == 255) { lvl2 = lvl2 + lvl1; lvl1 = 0.f; } }
                            }
                            c = (lvl0 + lvl1) + lvl2;   /* ATen: acc[0] += acc[1] += ... at the end */
                        }
                        const float mu_w = bilinear(src_gmm + (sidx * 2 + 0) * hw, h, w, &tp);
                        const float sg_w = bilinear(src_gmm + (sidx * 2 + 1) * hw, h, w, &tp);
                        const int gate = fabsf(zw - mu_w) < sg_w * kappa;
                        if (gates) gates[bvi] = (uint8_t)gate;
                        if (featcost) featcost[bvi] = c;
                        acc += (double)c * (gate ? 1.0 : 0.0);
                    }
                    out[((size_t)b * D + j) * hw + p] = (float)acc / (float)V;
                }
            }
        }
    }
    return 0;
}

/*
 * est_costvolume_F / _compute_cost_F (models/submodules/homography.py:10-75), BEFORE the final softmax:
 * fixed depth bins d_center[j] shared by all pixels, no consistency gate, feature cost summed over valid views
 * in FP32 (homography.py:42: ref_mv_cost = ref_mv_cost + cost, both fp32) and divided by ALL views (:46).
 * Also returns, when the grad_* pointers are given, the gradients of  L = sum(gout * out)  with respect to the
 * reference and source features (what autograd gives through grid_sample / mul / sum; double accumulation here,
 * compared with a tolerance) — the checker of the HIP backward kernel.
 */
ORACLE_API int magnet_oracle_cost_volume_f(
    const float *d_center, const float *ref_feat, const float *src_feat,
    const float *poses, const int32_t *is_valid, const float *intM, const float *rays,
    int B, int V, int F, int D, int h, int w,
    float *out, const float *gout, double *grad_ref, double *grad_src, int n_threads)
{
    const size_t hw = (size_t)h * w;
    const float cw = (float)((double)w / 2.0), ch = (float)((double)h / 2.0);
    const float sw_ = (float)w / 2.0f, sh_ = (float)h / 2.0f;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    /* gradients are accumulated serially over pixels when requested (small test shapes only) */
    const int want_grad = gout && grad_ref && grad_src;
    #pragma omp parallel for collapse(2) schedule(dynamic, 4) if(!want_grad)
    for (int b = 0; b < B; ++b) {
        for (int y = 0; y < h; ++y) {
            const float *K = intM + (size_t)b * 9;
            for (int x = 0; x < w; ++x) {
                const size_t p = (size_t)y * w + x;
                const float ray0 = rays[((size_t)b * 3 + 0) * hw + p];
                const float ray1 = rays[((size_t)b * 3 + 1) * hw + p];
                const float ray2 = rays[((size_t)b * 3 + 2) * hw + p];
                for (int j = 0; j < D; ++j) {
                    const float d = d_center[j];
                    float acc = 0.f;
                    for (int v = 0; v < V; ++v) {
                        if (is_valid[b * V + v] != 1) continue;
                        const float *T = poses + ((size_t)b * V + v) * 16;
                        float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
                        float t[3] = {T[3], T[7], T[11]};
                        float KR[9], Kt[3];
                        for (int i = 0; i < 3; ++i) {
                            for (int c = 0; c < 3; ++c)
                                KR[i * 3 + c] = dot3(K + i * 3, R[0 * 3 + c], R[1 * 3 + c], R[2 * 3 + c]);
                            Kt[i] = dot3_gemv(K + i * 3, t[0], t[1], t[2]);
                        }
                        const float rpx = dot3(KR + 0, ray0, ray1, ray2);
                        const float rpy = dot3(KR + 3, ray0, ray1, ray2);
                        const float rpz = dot3(KR + 6, ray0, ray1, ray2);
                        float Px = Kt[0] + rpx * d;
                        float Py = Kt[1] + rpy * d;
                        float Pz = Kt[2] + rpz * d;
                        const float zz = Pz + 1e-10f;
                        Px = Px / zz; Py = Py / zz;
                        float gx = (Px - cw) / cw, gy = (Py - ch) / ch;
                        gx = clampf(gx, -10.f, 10.f); gy = clampf(gy, -10.f, 10.f);
                        const float ix = __builtin_fmaf(gx + 1.0f, sw_, -0.5f);
                        const float iy = __builtin_fmaf(gy + 1.0f, sh_, -0.5f);
                        taps_t tp; make_taps(ix, iy, &tp);
                        const size_t sidx = (size_t)v * B + b;
                        const float *sf = src_feat + sidx * F * hw;
                        const float *rf = ref_feat + (size_t)b * F * hw + p;
                        float c = 0.f;
                        const int inw = tp.finite && tp.x0 >= -1 && tp.x0 < w && tp.y0 >= -1 && tp.y0 < h;
                        if (inw) {
                            float lvl0 = 0.f, lvl1 = 0.f, lvl2 = 0.f;
                            for (int f = 0; f < F; ++f) {
                                float wv = bilinear(sf + (size_t)f * hw, h, w, &tp);
                                float pr = rf[(size_t)f * hw] * wv;
                                lvl0 = lvl0 + pr;
                                if ((f & 15) == 15) { lvl1 = lvl1 + lvl0; lvl0 = 0.f;
                                    if ((f & 255) == 255) { lvl2 = lvl2 + lvl1; lvl1 = 0.f; } }
                            }
                            c = (lvl0 + lvl1) + lvl2;
                            if (want_grad) {
                                const double g = (double)gout[((size_t)b * D + j) * hw + p] / (double)V;
                                const int xs[4] = {tp.x0, tp.x0 + 1, tp.x0, tp.x0 + 1};
                                const int ys[4] = {tp.y0, tp.y0, tp.y0 + 1, tp.y0 + 1};
                                const float ws[4] = {tp.nw, tp.ne, tp.sw, tp.se};
                                for (int f = 0; f < F; ++f) {
                                    double warp = 0.0;
                                    for (int k = 0; k < 4; ++k) {
                                        if (ys[k] < 0 || ys[k] >= h || xs[k] < 0 || xs[k] >= w) continue;
                                        const size_t o = (size_t)ys[k] * w + xs[k];
                                        warp += (double)ws[k] * sf[(size_t)f * hw + o];
                                        grad_src[(sidx * F + f) * hw + o] += g * ws[k] * rf[(size_t)f * hw];
                                    }
                                    grad_ref[((size_t)b * F + f) * hw + p] += g * warp;
                                }
                            }
                        }
                        acc = acc + c;
                    }
                    out[((size_t)b * D + j) * hw + p] = acc / (float)V;
                }
            }
        }
    }
    return 0;
}

/* Number of OpenMP threads a call with n_threads <= 0 will use (bench.py reports it as `cores`). */
ORACLE_API int magnet_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
