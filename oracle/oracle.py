"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's hot path, each function citing
the reference file:line it follows.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg — as the checker / timed baseline only.  `magnet_amd` never imports
this module.

Parity pin: every function here is checked against golden vectors captured from the imported
reference (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile oracle/cost_volume_oracle.c -> oracle/libmagnet_oracle.so (gcc, OpenMP)."""
    so = os.path.join(_HERE, "libmagnet_oracle.so")
    src = os.path.join(_HERE, "cost_volume_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmagnet_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        f = _LIB.magnet_oracle_cost_volume_cw
        f.restype = ctypes.c_int
        P = ctypes.c_void_p
        f.argtypes = [P] * 10 + [ctypes.c_int] * 6 + [ctypes.c_float] + [P] * 3 + [ctypes.c_int]
        _LIB.magnet_oracle_num_threads.restype = ctypes.c_int
        ff = _LIB.magnet_oracle_cost_volume_f
        ff.restype = ctypes.c_int
        ff.argtypes = [P] * 7 + [ctypes.c_int] * 6 + [P] * 4 + [ctypes.c_int]
    return _LIB


def num_threads() -> int:
    return int(_lib().magnet_oracle_num_threads())


def _f32(a):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------------------------
# A1  MAGNET.depth_sampling  (models/MAGNET.py:120-128)
# ----------------------------------------------------------------------------------------------
def depth_sampling(sampling_range: float, n_samples: int) -> list:
    """Mid-points of the N(0,1) quantile bins covering +-sampling_range.

    The reference uses scipy.special.erf / scipy.stats.norm.ppf; this restatement needs only
    math.erf and the standard library's NormalDist().inv_cdf (Wichura AS241, same algorithm class
    as scipy's ndtri; agreement to ~1e-15 is asserted against the golden vector)."""
    from statistics import NormalDist
    P_total = math.erf(sampling_range / math.sqrt(2.0))
    idx = np.arange(0, n_samples + 1)
    p = (1 - P_total) / 2 + ((idx / n_samples) * P_total)
    nd = NormalDist()
    k = np.array([nd.inv_cdf(float(x)) for x in p], dtype=np.float64)
    k = (k[1:] + k[:-1]) / 2
    return list(k)


# ----------------------------------------------------------------------------------------------
# A2  candidate sampling  (models/MAGNET.py:153-156)
# ----------------------------------------------------------------------------------------------
def depth_volume(ref_gmm, k_list) -> np.ndarray:
    g = _f32(ref_gmm)
    mu, sg = g[:, 0:1], g[:, 1:2]
    return np.concatenate([mu + sg * np.float32(k) for k in k_list], axis=1)


# ----------------------------------------------------------------------------------------------
# A4/A5  est_costvolume_CW / _compute_cost_CW  (models/submodules/homography.py:79-161)
# ----------------------------------------------------------------------------------------------
def est_costvolume_CW(d_volume, ref_feat, nghbr_feat, ref_gmms, nghbr_gmms, R, t, is_valid,
                      cam_intrins, thres, k_list=None, return_aux=False, n_threads=0):
    """Same argument list as the reference function (numpy or torch inputs, NCHW fp32).

    If `d_volume` is None the candidates are sampled in place from (`ref_gmms`, `k_list`)
    (the fused form the HIP kernel implements).  Returns (B,D,h,w) fp32; with return_aux also
    the gate bits (B,V,D,h,w) uint8 and the un-gated per-view feature costs (B,V,D,h,w) fp32."""
    ref_feat = _f32(ref_feat); nghbr_feat = _f32(nghbr_feat); nghbr_gmms = _f32(nghbr_gmms)
    B, F, h, w = ref_feat.shape
    V = nghbr_feat.shape[0] // B
    R = _f32(R).reshape(B, V, 3, 3); t = _f32(t).reshape(B, V, 3)
    poses = np.zeros((B, V, 4, 4), np.float32)
    poses[:, :, :3, :3] = R; poses[:, :, :3, 3] = t; poses[:, :, 3, 3] = 1
    return cost_volume_cw(d_volume, ref_gmms, k_list, ref_feat, nghbr_feat, nghbr_gmms, poses,
                          is_valid, cam_intrins["intM"], cam_intrins["unit_ray_array_2D"],
                          float(thres), return_aux=return_aux, n_threads=n_threads)


def cost_volume_cw(d_volume, ref_gmm, k_list, ref_feat, src_feat, src_gmm, poses, is_valid,
                   intM, rays, kappa, return_aux=False, n_threads=0):
    ref_feat = _f32(ref_feat); src_feat = _f32(src_feat); src_gmm = _f32(src_gmm)
    poses = _f32(poses); intM = _f32(intM); rays = _f32(rays)
    B, F, h, w = ref_feat.shape
    V = src_feat.shape[0] // B
    iv = np.ascontiguousarray(
        is_valid.detach().cpu().numpy() if hasattr(is_valid, "detach") else is_valid, dtype=np.int32)
    if d_volume is not None:
        dv = _f32(d_volume); D = dv.shape[1]; rg = None; kl = None
    else:
        dv = None; rg = _f32(ref_gmm); kl = np.ascontiguousarray(k_list, dtype=np.float64); D = len(kl)
    out = np.empty((B, D, h, w), np.float32)
    gates = np.empty((B, V, D, h, w), np.uint8) if return_aux else None
    fc = np.empty((B, V, D, h, w), np.float32) if return_aux else None
    rc = _lib().magnet_oracle_cost_volume_cw(
        _ptr(dv), _ptr(rg), _ptr(kl), _ptr(ref_feat), _ptr(src_feat), _ptr(src_gmm), _ptr(poses),
        _ptr(iv), _ptr(intM), _ptr(rays), B, V, F, D, h, w, ctypes.c_float(kappa),
        _ptr(out), _ptr(gates), _ptr(fc), int(n_threads))
    assert rc == 0
    return (out, gates, fc) if return_aux else out


# ----------------------------------------------------------------------------------------------
# N2  est_costvolume_F / _compute_cost_F  (models/submodules/homography.py:10-75) + gradients
# ----------------------------------------------------------------------------------------------
def cost_volume_f_raw(d_center, ref_feat, src_feat, poses, is_valid, intM, rays, gout=None, n_threads=0):
    """Feature-matching cost volume BEFORE the softmax (homography.py:46): (B,D,h,w) fp32.  With `gout` (B,D,h,w)
    also returns d(sum(gout*out))/d ref_feat and /d src_feat (float64) for the backward-kernel check."""
    ref_feat = _f32(ref_feat); src_feat = _f32(src_feat); poses = _f32(poses); intM = _f32(intM); rays = _f32(rays)
    dc = np.ascontiguousarray(np.asarray(d_center.detach().cpu().numpy() if hasattr(d_center, "detach") else d_center,
                                         dtype=np.float32).reshape(-1))
    B, F, h, w = ref_feat.shape
    V = src_feat.shape[0] // B
    D = dc.shape[0]
    iv = np.ascontiguousarray(is_valid.detach().cpu().numpy() if hasattr(is_valid, "detach") else is_valid, dtype=np.int32)
    out = np.empty((B, D, h, w), np.float32)
    if gout is not None:
        go = _f32(gout); gr = np.zeros(ref_feat.shape, np.float64); gs = np.zeros(src_feat.shape, np.float64)
    else:
        go = gr = gs = None
    rc = _lib().magnet_oracle_cost_volume_f(_ptr(dc), _ptr(ref_feat), _ptr(src_feat), _ptr(poses), _ptr(iv), _ptr(intM),
                                            _ptr(rays), B, V, F, D, h, w, _ptr(out), _ptr(go), _ptr(gr), _ptr(gs),
                                            int(n_threads))
    assert rc == 0
    return (out, gr, gs) if gout is not None else out


def softmax_dim1(x):
    x = _f32(x)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def est_costvolume_F(d_center, ref_feat, nghbr_feat, R, t, is_valid, cam_intrins, n_threads=0):
    """Reference signature (homography.py:10): softmax over D of the raw cost volume."""
    ref_feat = _f32(ref_feat)
    B = ref_feat.shape[0]; V = _f32(nghbr_feat).shape[0] // B
    R = _f32(R).reshape(B, V, 3, 3); t = _f32(t).reshape(B, V, 3)
    poses = np.zeros((B, V, 4, 4), np.float32)
    poses[:, :, :3, :3] = R; poses[:, :, :3, 3] = t; poses[:, :, 3, 3] = 1
    return softmax_dim1(cost_volume_f_raw(d_center, ref_feat, nghbr_feat, poses, is_valid, cam_intrins["intM"],
                                          cam_intrins["unit_ray_array_2D"], n_threads=n_threads))


# ----------------------------------------------------------------------------------------------
# A6 tail  GNET.forward's Gaussian update  (models/MAGNET.py:58-69)
# ----------------------------------------------------------------------------------------------
def gaussian_update(d_output, ref_gmm) -> np.ndarray:
    """mu' = mu + o0*sigma ; sigma' = (elu(o1) + 1 + 1e-10)*sigma   (fp32, reference op order)."""
    o = _f32(d_output); g = _f32(ref_gmm)
    mu0, sg0 = g[:, 0:1], g[:, 1:2]
    o0, o1 = o[:, 0:1], o[:, 1:2]
    mu_new = mu0 + (o0 * sg0)
    elu = np.where(o1 > 0, o1, np.expm1(np.minimum(o1, 0).astype(np.float32))).astype(np.float32)
    sg_new = ((elu + np.float32(1.0)) + np.float32(1e-10)) * sg0
    return np.concatenate([mu_new, sg_new], axis=1).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# A8  upsample_depth_via_mask  (models/MAGNET.py:15-27)
# ----------------------------------------------------------------------------------------------
def upsample_depth_via_mask(depth, up_mask, k: int) -> np.ndarray:
    """out[b,c,k*y+i,k*x+j] = sum_n softmax_n(mask[b, n*k*k+i*k+j, y, x]) * pad0(depth)[b,c,y+n//3-1,x+n%3-1]."""
    d = _f32(depth); m = _f32(up_mask)
    N, C, H, W = d.shape
    m = m.reshape(N, 1, 9, k, k, H, W)
    m = m - m.max(axis=2, keepdims=True)
    e = np.exp(m)
    sm = e / e.sum(axis=2, keepdims=True)
    pad = np.zeros((N, C, H + 2, W + 2), np.float32)
    pad[:, :, 1:-1, 1:-1] = d
    nb = np.stack([pad[:, :, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], axis=2)
    up = (sm * nb.reshape(N, C, 9, 1, 1, H, W)).sum(axis=2)       # N,C,k,k,H,W
    up = up.transpose(0, 1, 4, 2, 5, 3)                           # N,C,H,k,W,k
    return np.ascontiguousarray(up.reshape(N, C, k * H, k * W), dtype=np.float32)


# ----------------------------------------------------------------------------------------------
# A10  utils.data_preprocess  (utils/utils.py:72-98)
# ----------------------------------------------------------------------------------------------
def relative_poses(ref_extM, nghbr_extMs):
    """nghbr_pose = ext_nghbr @ inv(ext_ref); any NaN -> is_valid = 0 and a zero pose."""
    ref_extM = np.asarray(ref_extM)
    B = ref_extM.shape[0]; V = len(nghbr_extMs)
    poses = np.zeros((B, V, 4, 4), np.float32)
    valid = np.ones((B, V), np.int32)
    for i in range(B):
        if np.isnan(ref_extM[i].min()):
            valid[i, :] = 0
            continue
        inv = np.linalg.inv(ref_extM[i])
        for j in range(V):
            e = np.asarray(nghbr_extMs[j][i])
            if np.isnan(e.min()):
                valid[i, j] = 0
                continue
            p = e @ inv
            if np.isnan(p.min()):
                valid[i, j] = 0
            else:
                poses[i, j] = p
    return poses, valid


# ----------------------------------------------------------------------------------------------
# A11  utils.compute_depth_errors  (utils/utils.py:106-144) — abs_rel is the parity metric
# ----------------------------------------------------------------------------------------------
def compute_depth_errors(gt, pred, var=None) -> dict:
    gt = np.asarray(gt); pred = np.asarray(pred)
    thresh = np.maximum(gt / pred, pred / gt)
    out = dict(a1=(thresh < 1.25).mean(), a2=(thresh < 1.25 ** 2).mean(), a3=(thresh < 1.25 ** 3).mean(),
               abs_diff=np.mean(np.abs(gt - pred)), abs_rel=np.mean(np.abs(gt - pred) / gt),
               sq_rel=np.mean(((gt - pred) ** 2) / gt), rmse=np.sqrt(((gt - pred) ** 2).mean()),
               log_10=np.abs(np.log10(gt) - np.log10(pred)).mean(),
               irmse=np.sqrt(((1 / gt - 1 / pred) ** 2).mean()),
               rmse_log=np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean()))
    err = np.log(pred) - np.log(gt)
    out["silog"] = np.sqrt(np.mean(err ** 2) - np.mean(err) ** 2) * 100
    if var is not None:
        var = np.array(var, copy=True); var[var < 1e-6] = 1e-6
        out["nll"] = np.mean(0.5 * (np.log(var) + np.log(2 * np.pi) + np.square(gt - pred) / var))
    else:
        out["nll"] = 0.0
    return out


def abs_rel(gt, pred) -> float:
    return float(compute_depth_errors(gt, pred)["abs_rel"])
