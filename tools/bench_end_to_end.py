"""End-to-end reference frames/s WITH the F-Net in the timed step (C2 shapes: 480x640 images, V = 4, D = 64, I = 1):
images -> F-Net on the matrix cores (features land in the matcher's layouts) -> matcher -> G-Net -> update -> mask head ->
upsampling.  The D-Net cannot be built offline (torch.hub), so its outputs ((mu,sigma) maps and x_d3) are resident
synthetic tensors, exactly as in bench.py.  One JSON line; `--torch-fnet` runs the F-Net as the torch module (MIOpen)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import fnet, synth  # noqa: E402
from magnet_amd.magnet import MAGNET  # noqa: E402


class ResidentDNet(nn.Module):
    def __init__(self, gmms, x_d3):
        super().__init__()
        self.gmms, self.x_d3 = gmms, x_d3

    def forward(self, img):
        return self.gmms, self.x_d3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--torch-fnet", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl = synth.WORKLOADS["C2"]
    B, V, h, w = a.frames, wl.V, wl.h, wl.w

    class Args:
        MAGNET_sampling_range = 3; MAGNET_num_samples = wl.D; MAGNET_mvs_weighting = "CW5"
        MAGNET_num_train_iter = 1; MAGNET_num_test_iter = 1; dpv_height = h; dpv_width = w; downsample_ratio = 4
        FNET_architecture = "PSM-Net"; FNET_feature_dim = 64
    torch.manual_seed(0)
    g = torch.Generator(device=dev).manual_seed(1)
    cam = synth.CAMERAS[wl.camera]
    uni = lambda lo, hi, *s: torch.rand(*s, generator=g, device=dev) * (hi - lo) + lo
    N = (1 + V) * B
    gmms = torch.cat([uni(*cam["mu"], N, 1, h, w), uni(*cam["sigma"], N, 1, h, w)], dim=1)
    x_d3 = torch.randn(N, 256, h, w, generator=g, device=dev) * 0.5
    model = MAGNET(Args(), d_net=ResidentDNet(gmms, x_d3), f_net=fnet.FNET(Args()), feat_dtype="bf16").to(dev).eval()
    model.fnet_mfma = not a.torch_fnet
    ref_img = torch.randn(B, 3, 4 * h, 4 * w, generator=g, device=dev)
    nb_img = torch.randn(V * B, 3, 4 * h, 4 * w, generator=g, device=dev)
    poses = synth.make_poses(wl.camera, B, V, torch.Generator().manual_seed(2)).to(dev)
    valid = torch.ones(B, V, dtype=torch.int32)
    intr = synth.make_intrinsics(wl.camera, h, w, B)

    def step():
        with torch.no_grad():
            return model(ref_img, nb_img, poses, valid, intr, mode="test")
    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": "C2 + F-Net: 480x640 images, V=4, D=64, I=1, bf16 features; D-Net outputs resident",
                      "frames_per_step": B, "ms_per_step": dt * 1e3, "ref_frames_per_s": B / dt,
                      "fnet": "torch (MIOpen fp32) + pack" if a.torch_fnet else "matrix-core path (magnet_amd/fnet.py)"}))


if __name__ == "__main__":
    main()
