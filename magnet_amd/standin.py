"""Deterministic stand-ins for the backbones this build does not contain (the D-Net needs torch.hub and
its checkpoint; the F-Net has a matrix-core implementation in magnet_amd/fnet.py but no weights offline): tiny seeded
modules with the backbones' output contracts, the argparse fields `MAGNET.__init__` reads, and seeded g_net / mask_head
weights.  Used by the synthetic evaluation driver (eval_synthetic.py), `__graft_entry__.smoke()` and the tests — the golden
generator drives the REFERENCE `MAGNET.forward` with the same stand-ins, so both sides see identical (mu, sigma, x_d3,
features)."""
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


def seeded_magnet_weights(model, seed=0, gain=1.0):
    """Deterministic g_net / mask_head weights (same draw order for the reference module and ours:
    both expose `g_net.gnet.{0,2,4,6}` and `mask_head.{0,2,4,6}`); `gain` scales every tensor (the training-step vector
    uses 0.25 so that sigma stays away from the loss's variance clamp)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in (model.g_net, model.mask_head):
            for name, p in sorted(mod.state_dict().items()):
                scale = 0.05 if p.dim() > 1 else 0.01
                p.copy_(torch.randn(p.shape, generator=g) * (scale * gain))
