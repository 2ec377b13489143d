"""Time est_costvolume_F (row N2: the F-Net training volume) forward + backward on one GPU at the reference's
training shapes (train_scripts/fnet/*.txt: D = 80 SID bins, F = 64, ScanNet 120x160 V=4, KITTI 88x304 V=2).
Prints one JSON line per shape.  `--cpu` also times the CPU oracle (forward + gradients) on one batch."""
import argparse
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import homography, synth  # noqa: E402

SHAPES = {"scannet": dict(cam="scannet", h=120, w=160, V=4, dmax=10.0), "kitti": dict(cam="kitti", h=88, w=304, V=2, dmax=80.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--path", type=lambda x: int(x, 0), default=0, help="backward kernel: 0 gather (default), 0x4000 gather with private "
                    "accumulator copies, 0x1000 LDS hash-table scatter, 0x2000 per-item atomics")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, s in SHAPES.items():
        wl = synth.Workload(name, s["cam"], s["h"], s["w"], V=s["V"], D=80, F=64)
        inp = synth.make_inputs(wl, B=a.batch, seed=1, smooth_feats=True)
        D = 80
        b = np.exp(np.log(s["dmax"] + 1 - 1e-3) * np.arange(D + 1) / D) - (1 - 1e-3)
        dc = torch.tensor(((b[:-1] + b[1:]) / 2).astype(np.float32)).view(1, D, 1, 1)
        rf = inp["ref_feat"].to(dev).requires_grad_(True); sf = inp["nghbr_feat"].to(dev).requires_grad_(True)
        R = inp["nghbr_poses"][:, :, :3, :3].to(dev); t = inp["nghbr_poses"][:, :, :3, 3].to(dev)
        dcd = dc.to(dev)

        def step():
            cv = homography.est_costvolume_F(dc, rf, sf, R, t, inp["is_valid"], inp["cam_intrins"], bwd_path=a.path)
            loss = (cv * dcd).sum(dim=1).abs().mean()                 # train_FNet.py:96 expectation + an L1-like loss
            rf.grad = None; sf.grad = None
            loss.backward()

        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
        rec = {"shape": name, "B": a.batch, "V": s["V"], "D": D, "F": 64, "h": s["h"], "w": s["w"],
               "ms_fwd_bwd": dt * 1e3, "frames_per_s": a.batch / dt, "bwd_path": hex(a.path)}
        if a.cpu:
            from oracle import oracle
            nb = min(2, a.batch)
            args = (dc.numpy(), inp["ref_feat"][:nb].numpy(),
                    inp["nghbr_feat"].view(s["V"], a.batch, 64, s["h"], s["w"])[:, :nb].reshape(-1, 64, s["h"], s["w"]).contiguous().numpy(),
                    inp["nghbr_poses"][:nb].numpy(), inp["is_valid"][:nb].numpy(), inp["cam_intrins"]["intM"][:nb].numpy(),
                    inp["cam_intrins"]["unit_ray_array_2D"][:nb].numpy())
            g = np.ones((nb, D, s["h"], s["w"]), np.float32)
            t0 = time.perf_counter(); oracle.cost_volume_f_raw(*args, gout=g); dtc = time.perf_counter() - t0
            rec["cpu_oracle_frames_per_s"] = nb / dtc; rec["cpu_threads"] = oracle.num_threads()
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
