"""Tiny deterministic stand-ins for the out-of-scope backbones (D-Net needs torch.hub / network,
SURVEY.md §8c), shared by the golden generator (driving the REFERENCE MAGNET.forward) and the
tests (driving magnet_amd.MAGNET) so both see identical (mu, sigma, x_d3, features)."""
from types import SimpleNamespace

import torch
import torch.nn as nn


class StubDNet(nn.Module):
    """img (N,3,H,W) -> ((N,2,H/4,W/4) [mu, sigma>0], (N,256,H/4,W/4)) like DNET(dnet=False)
    (reference: models/DNET.py:62-67, submodules/D_dense_depth.py:187-195)."""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.head = nn.Conv2d(3, 2, 4, stride=4)
        self.feat = nn.Conv2d(3, 256, 4, stride=4)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)

    def forward(self, img):
        o = self.head(img)
        mu = 1.0 + 3.0 * torch.sigmoid(o[:, 0:1])
        sigma = 0.05 + 0.3 * torch.sigmoid(o[:, 1:2])
        return torch.cat([mu, sigma], dim=1), self.feat(img)


class StubFNet(nn.Module):
    """img (N,3,H,W) -> (N,fdim,H/4,W/4) linear signed features like FNET (models/FNET.py:19-20)."""

    def __init__(self, seed=0, fdim=64):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.conv = nn.Conv2d(3, fdim, 4, stride=4)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.7)

    def forward(self, img):
        return self.conv(img)


def make_args(D=5, iters=3, dpv_h=120, dpv_w=160, beta=3, weighting="CW5", fdim=64, V=4):
    """The argparse fields MAGNET.__init__ reads (reference: models/MAGNET.py:95-104,
    test_MaGNet.py:89-147)."""
    return SimpleNamespace(
        MAGNET_sampling_range=beta, MAGNET_num_samples=D, MAGNET_mvs_weighting=weighting,
        MAGNET_num_train_iter=iters, MAGNET_num_test_iter=iters, MAGNET_num_source_views=V,
        dpv_height=dpv_h, dpv_width=dpv_w, downsample_ratio=4, FNET_feature_dim=fdim,
        DNET_ckpt=None, FNET_ckpt=None, MAGNET_ckpt=None)


def seeded_magnet_weights(model, seed=0):
    """Deterministic g_net / mask_head weights (same draw order for the reference module and ours:
    both expose `g_net.gnet.{0,2,4,6}` and `mask_head.{0,2,4,6}`)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in (model.g_net, model.mask_head):
            for name, p in sorted(mod.state_dict().items()):
                scale = 0.05 if p.dim() > 1 else 0.01
                p.copy_(torch.randn(p.shape, generator=g) * scale)


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
