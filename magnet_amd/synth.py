"""Seeded synthetic inputs shaped like the reference's data contract.

Nothing here reads a dataset: it produces the tensors `MAGNET.forward` /
`est_costvolume_CW` consume, with the layouts the reference's loaders define:

* ``cam_intrins`` = {'intM': (B,3,3) fp32, 'unit_ray_array_2D': (B,3,h*w) fp32}, float64 math
  then cast, row-major p = y*w + x, ray = ((x+.5)*sx-cx)/fx, ((y+.5)*sy-cy)/fy, 1
  (reference: data/dataloader_scannet.py:113-153, dataloader_kitti.py:83-127,
  dataloader_7scenes.py:72-116).
* ``nghbr_poses`` (B,V,4,4) fp32 relative poses, ``is_valid`` (B,V) int32 on CPU
  (reference: utils/utils.py:72-98).
* source-view tensors are VIEW-MAJOR: index = v*B + b (reference: homography.py:105,
  test_MaGNet.py:45-46).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

# name -> (fx, fy, cx, cy, raw_W, raw_H, mu range, sigma range, pose kind)
CAMERAS = {
    "scannet": dict(fx=1170.0, fy=1170.0, cx=648.0, cy=484.0, raw_w=1296, raw_h=968,
                    mu=(1.0, 4.0), sigma=(0.05, 0.35), t_std=0.1, forward=0.0),
    "7scenes": dict(fx=585.0, fy=585.0, cx=320.0, cy=240.0, raw_w=640, raw_h=480,
                    mu=(1.0, 4.0), sigma=(0.05, 0.35), t_std=0.1, forward=0.0),
    "kitti": dict(fx=721.5, fy=721.5, cx=609.6, cy=172.9, raw_w=1242, raw_h=375,
                  mu=(2.0, 60.0), sigma=(0.5, 3.0), t_std=0.05, forward=0.8),
}


@dataclass
class Workload:
    """One named shape from BASELINE.json / SURVEY.md §8."""
    name: str
    camera: str
    h: int
    w: int
    V: int
    D: int
    F: int = 64
    iters: int = 1
    feat_dtype: str = "fp32"   # storage dtype of F-Net features handed to the kernel

    @property
    def hw(self) -> int:
        return self.h * self.w

    def algorithmic_bytes(self) -> int:
        """s_f*hw*F*(1+V) + 4*hw*(2V+2+D) per ref-frame-iteration."""
        s_f = 2 if self.feat_dtype == "bf16" else 4
        return s_f * self.hw * self.F * (1 + self.V) + 4 * self.hw * (2 * self.V + 2 + self.D)


WORKLOADS = {
    "C1": Workload("C1", "scannet", 128, 160, V=2, D=16),
    "C2": Workload("C2", "scannet", 120, 160, V=4, D=64, iters=1, feat_dtype="bf16"),
    "C3": Workload("C3", "scannet", 120, 160, V=4, D=64, iters=3, feat_dtype="bf16"),
    "C4": Workload("C4", "kitti", 88, 304, V=4, D=128),
    "C5": Workload("C5", "7scenes", 120, 160, V=6, D=64),
    "C2L": Workload("C2L", "scannet", 480, 640, V=4, D=64, feat_dtype="bf16"),
    "C2Lf": Workload("C2Lf", "scannet", 480, 640, V=4, D=64, feat_dtype="fp32"),
    "C4L": Workload("C4L", "kitti", 352, 1216, V=4, D=128),
    "shipped": Workload("shipped", "scannet", 120, 160, V=4, D=5, iters=3),
}


def make_intrinsics(camera: str, h: int, w: int, B: int):
    """K scaled to the matching grid and the unit-ray table, as the reference loaders build them."""
    c = CAMERAS[camera]
    intM = np.zeros((3, 3))
    intM[2, 2] = 1.0
    intM[0, 0] = c["fx"] * (w / c["raw_w"])
    intM[1, 1] = c["fy"] * (h / c["raw_h"])
    intM[0, 2] = c["cx"] * (w / c["raw_w"])
    intM[1, 2] = c["cy"] * (h / c["raw_h"])
    xs = (np.arange(w) + 0.5)[None, :].repeat(h, 0)
    ys = (np.arange(h) + 0.5)[:, None].repeat(w, 1)
    ray = np.ones((h, w, 3))
    ray[:, :, 0] = (xs * (c["raw_w"] / w) - c["cx"]) / c["fx"]
    ray[:, :, 1] = (ys * (c["raw_h"] / h) - c["cy"]) / c["fy"]
    ray2d = np.reshape(np.transpose(ray, (2, 0, 1)), (3, -1)).astype(np.float32)
    return {
        "intM": torch.from_numpy(intM.astype(np.float32))[None].repeat(B, 1, 1).contiguous(),
        "unit_ray_array_2D": torch.from_numpy(ray2d)[None].repeat(B, 1, 1).contiguous(),
    }


def _rodrigues(omega: torch.Tensor) -> torch.Tensor:
    """exp([omega]x) for (...,3) float64 vectors."""
    theta = omega.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = omega / theta
    K = torch.zeros(omega.shape[:-1] + (3, 3), dtype=omega.dtype)
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    s = torch.sin(theta)[..., None]
    c = torch.cos(theta)[..., None]
    eye = torch.eye(3, dtype=omega.dtype).expand_as(K)
    return eye + s * K + (1 - c) * (K @ K)


def make_poses(camera: str, B: int, V: int, gen: torch.Generator) -> torch.Tensor:
    c = CAMERAS[camera]
    omega = torch.randn(B, V, 3, generator=gen, dtype=torch.float64) * 0.05
    t = torch.randn(B, V, 3, generator=gen, dtype=torch.float64) * c["t_std"]
    if c["forward"] > 0:   # KITTI: forward motion along z, +-0.8 m per frame gap
        gaps = torch.tensor([(-1) ** i * (1 + i // 2) for i in range(V)], dtype=torch.float64)
        t[..., 2] += c["forward"] * gaps[None, :]
    poses = torch.zeros(B, V, 4, 4, dtype=torch.float64)
    poses[..., :3, :3] = _rodrigues(omega)
    poses[..., :3, 3] = t
    poses[..., 3, 3] = 1.0
    return poses.to(torch.float32)


def _box(x: torch.Tensor, r: int, passes: int = 2) -> torch.Tensor:
    """(2r+1)^2 box filter applied `passes` times to every (h, w) plane of x (zero padding): a cheap low-pass."""
    k = torch.ones(1, 1, 2 * r + 1, 2 * r + 1) / float((2 * r + 1) ** 2)
    y = x.reshape(-1, 1, x.shape[-2], x.shape[-1])
    for _ in range(passes):
        y = torch.nn.functional.conv2d(y, k, padding=r)
    return y.reshape(x.shape)


def make_inputs(wl: Workload, B: int, seed: int = 0, smooth_feats: bool = False,
                invalid=(), round_bf16: bool | None = None, smooth: int = 0):
    """Inputs of `est_costvolume_CW` plus the D-Net side tensors the loop needs.

    Returns a dict of CPU tensors:
      ref_feat (B,F,h,w), nghbr_feat (V*B,F,h,w) view-major, ref_gmms (B,2,h,w),
      nghbr_gmms (V*B,2,h,w), nghbr_poses (B,V,4,4), is_valid (B,V) int32,
      cam_intrins {...}, x_d3 is NOT generated here (bench builds it on device).
    `invalid` is a list of (b, v) pairs to mark is_valid=0.
    `round_bf16` (default: wl.feat_dtype == 'bf16') rounds features to bf16 and back, which is
    the parity definition for bf16 storage (oracle sees the rounded values).
    `smooth` = r > 0: the SMOOTH variant (SURVEY.md §8d "optionally smooth ... so bilinear taps are correlated"): features
    low-passed by two passes of a (2r+1)^2 box and rescaled to unit variance, (mu, sigma) maps low-passed the same way around
    their range centre — what real F-Net / D-Net outputs look like (neighbouring texels correlated), against the default
    white-noise maps in which a 1e-5 texel shift of the sample position already moves a 64-channel score by 1e-4.
    """
    gen = torch.Generator().manual_seed(seed)
    c = CAMERAS[wl.camera]
    h, w, V, F = wl.h, wl.w, wl.V, wl.F
    ref_feat = torch.randn(B, F, h, w, generator=gen)
    nghbr_feat = torch.randn(V * B, F, h, w, generator=gen)
    if smooth_feats:
        k = torch.ones(1, 1, 3, 3) / 9.0
        sm = lambda x: torch.nn.functional.conv2d(
            x.reshape(-1, 1, h, w), k, padding=1).reshape(x.shape)
        ref_feat, nghbr_feat = sm(ref_feat), sm(nghbr_feat)
    if smooth > 0:
        def lp(x):
            y = _box(x, smooth)
            return y / y.std().clamp_min(1e-6)
        ref_feat, nghbr_feat = lp(ref_feat), lp(nghbr_feat)
    if round_bf16 is None:
        round_bf16 = wl.feat_dtype == "bf16"
    if round_bf16:
        ref_feat = ref_feat.to(torch.bfloat16).to(torch.float32)
        nghbr_feat = nghbr_feat.to(torch.bfloat16).to(torch.float32)

    def gmm(n):
        mu = torch.rand(n, 1, h, w, generator=gen) * (c["mu"][1] - c["mu"][0]) + c["mu"][0]
        sg = torch.rand(n, 1, h, w, generator=gen) * (c["sigma"][1] - c["sigma"][0]) + c["sigma"][0]
        if smooth > 0:       # smooth depth / uncertainty maps with the same range: blur the deviation from the range centre, restore its scale
            def lpr(x, lo, hi):
                mid, dev = 0.5 * (lo + hi), x - 0.5 * (lo + hi)
                y = _box(dev, smooth)
                y = y * (0.5 * (hi - lo) / y.abs().max().clamp_min(1e-6))
                return mid + y
            mu, sg = lpr(mu, *c["mu"]), lpr(sg, *c["sigma"])
        return torch.cat([mu, sg], dim=1)

    ref_gmms = gmm(B)
    nghbr_gmms = gmm(V * B)
    poses = make_poses(wl.camera, B, V, gen)
    is_valid = torch.ones(B, V, dtype=torch.int32)
    for (b, v) in invalid:
        is_valid[b, v] = 0
    return dict(ref_feat=ref_feat, nghbr_feat=nghbr_feat, ref_gmms=ref_gmms,
                nghbr_gmms=nghbr_gmms, nghbr_poses=poses, is_valid=is_valid,
                cam_intrins=make_intrinsics(wl.camera, h, w, B))


def depth_volume_from_gmm(ref_gmms: torch.Tensor, k_list) -> torch.Tensor:
    """d[b,j] = mu + sigma*k_j, the reference's op order (MAGNET.py:153-156): k is a python
    float (float64) multiplied into an fp32 tensor, then added."""
    mu, sg = torch.split(ref_gmms, 1, dim=1)
    return torch.cat([mu + sg * float(k) for k in k_list], dim=1)
