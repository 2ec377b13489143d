"""Tiny deterministic stand-ins for the out-of-scope backbones (D-Net needs torch.hub / network,
SURVEY.md §8c), shared by the golden generator (driving the REFERENCE MAGNET.forward) and the
tests (driving magnet_amd.MAGNET) so both see identical (mu, sigma, x_d3, features)."""
from types import SimpleNamespace

import torch
import torch.nn as nn


from magnet_amd.standin import StubDNet, StubFNet, make_args, seeded_magnet_weights  # noqa: F401  (shared with the drivers)


def seeded_fnet_state(module, seed=0):
    """Deterministic, platform-independent parameters/buffers for a PSMNet-structured module (ours or the reference's):
    every tensor is drawn from numpy's MT19937 seeded by crc32(key) + seed, so the golden generator (reference class)
    and the tests (magnet_amd.fnet.PSMNet) rebuild identical weights without a 13 MB fixture."""
    import zlib
    import numpy as np
    sd = module.state_dict()
    out = {}
    for key, t in sd.items():
        rs = np.random.RandomState((zlib.crc32(key.encode()) + seed) & 0x7fffffff)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.tensor(100, dtype=t.dtype)
        elif t.dim() == 4:                                      # conv weight: N(0, sqrt(2/(k*k*cout))) like the reference's init
            cout, _, kh, kw = t.shape
            out[key] = torch.from_numpy(rs.standard_normal(tuple(t.shape)) * np.sqrt(2.0 / (kh * kw * cout))).float()
        elif key.endswith("running_var"):
            out[key] = torch.from_numpy(rs.uniform(0.5, 1.5, tuple(t.shape))).float()
        elif key.endswith("running_mean"):
            out[key] = torch.from_numpy(rs.standard_normal(tuple(t.shape)) * 0.1).float()
        elif key.endswith("weight"):                            # BN gamma; small on the residual branch (25 un-normalised sums)
            lo, hi = (0.1, 0.3) if ".conv2.1." in key else (0.5, 1.0)
            out[key] = torch.from_numpy(rs.uniform(lo, hi, tuple(t.shape))).float()
        else:                                                   # BN beta
            out[key] = torch.from_numpy(rs.standard_normal(tuple(t.shape)) * 0.1).float()
    module.load_state_dict(out)
    return module


def procedural_images(N, H, W):
    """Deterministic (N,3,H,W) fp32 test images from closed-form fp64 expressions (no RNG dependence)."""
    import numpy as np
    y, x = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    img = np.empty((N, 3, H, W), np.float64)
    for n in range(N):
        for c in range(3):
            img[n, c] = np.sin(0.05 * x * (c + 1) + 0.3 * n) * np.cos(0.07 * y + 0.5 * c) + 0.1 * ((x * y + 3 * n) % 7) - 0.3
    return torch.from_numpy(img.astype(np.float32))


def train_case():
    """Inputs of the training-step golden vector G11 (tests/golden/make_golden_r2.py): the G6 stub configuration plus a
    seeded ground-truth depth map and validity mask at full resolution."""
    from magnet_amd import synth
    args = make_args(D=5, iters=3, dpv_h=12, dpv_w=16)
    gen = torch.Generator().manual_seed(41)
    B, V = 2, 3
    ref_img = torch.rand(B, 3, 48, 64, generator=gen)
    nghbr_imgs = torch.rand(V * B, 3, 48, 64, generator=gen)
    poses = synth.make_poses("scannet", B, V, gen)
    valid = torch.ones(B, V, dtype=torch.int32); valid[1, 2] = 0
    intr = synth.make_intrinsics("scannet", 12, 16, B)
    gt = torch.rand(B, 1, 48, 64, generator=gen) * 3.0 + 1.0
    gt_mask = torch.rand(B, 1, 48, 64, generator=gen) > 0.2
    return args, ref_img, nghbr_imgs, poses, valid, intr, gt, gt_mask


def g13_case():
    """Inputs of the D = 64 full-forward golden vector G13 (tests/golden/make_golden_r3.py): 240 x 320 images -> 60 x 80 grid,
    B = 2 reference frames, V = 4 source views (one invalid), D = 64, I = 3, F = 64 — regenerated from seeds by the test."""
    from magnet_amd import synth
    args = make_args(D=64, iters=3, dpv_h=60, dpv_w=80)
    gen = torch.Generator().manual_seed(131)
    B, V = 2, 4
    ref_img = torch.rand(B, 3, 240, 320, generator=gen)
    nghbr_imgs = torch.rand(V * B, 3, 240, 320, generator=gen)
    poses = synth.make_poses("scannet", B, V, gen)
    valid = torch.ones(B, V, dtype=torch.int32); valid[1, 2] = 0
    intr = synth.make_intrinsics("scannet", 60, 80, B)
    # gain 0.3 on the seeded G-Net weights: sigma stays O(0.1) and mu moves by ~1 % per iteration (with gain 1 sigma collapses to 1e-3 after the
    # first update and the later iterations' gates are almost all closed — a vacuous check of the matcher)
    return args, ref_img, nghbr_imgs, poses, valid, intr, dict(d=121, f=122, w=123, gain=0.3)


def magnet_nll_loss(pred_list, gt_depth, gt_mask, gamma=0.8):
    """The reference's training loss for MaGNet (utils/losses.py:28-52, 'gaussian'): gamma-weighted mean NLL of every
    iteration's (mu, sigma) against the ground truth at the valid pixels.  Test-side restatement (losses are outside the
    product's scope); pinned by G11_loss."""
    gt = gt_depth[gt_mask]
    n = len(pred_list)
    loss = 0.0
    for i, pred in enumerate(pred_list):
        mu, sigma = pred[:, 0:1][gt_mask], pred[:, 1:2][gt_mask]
        var = torch.clamp(sigma * sigma, min=1e-10)
        loss = loss + gamma ** (n - i - 1) * torch.mean((mu - gt) ** 2 / (2 * var) + 0.5 * torch.log(var))
    return loss


def c5_case():
    """Inputs of the C5 end-to-end golden vector G15 (tests/golden/make_golden_r4.py): 480 x 640 images, V = 6, D = 64, I = 1, F = 64,
    7-Scenes intrinsics with the loader's ray table (the reference reads the table), seeded poses; everything regenerated by the test."""
    from magnet_amd import data, synth
    V, D = 6, 64
    args = make_args(D=D, iters=1, dpv_h=120, dpv_w=160, fdim=64, V=V)
    args.FNET_architecture, args.FNET_feature_dim = "PSM-Net", 64
    gen = torch.Generator().manual_seed(3)
    poses = synth.make_poses("7scenes", 1, V, gen)
    valid = torch.ones(1, V, dtype=torch.int32)
    cam = {kk: vv[None] for kk, vv in data.cam_intrinsics_7scenes(120, 160).items() if kk != "ray_params"}
    ref_img = procedural_images(1, 480, 640); nb = procedural_images(V, 480, 640).flip(0)
    return args, ref_img, nb, poses, valid, cam, dict(d=0, f=5, w=4)
