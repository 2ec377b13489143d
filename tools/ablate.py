#!/usr/bin/env python3
"""Dev tool (GPU box): time the fused cost-volume kernel alone under ablations / paths.
usage: python tools/ablate.py [workload] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import synth, lib
from magnet_amd.homography import CostVolumeCW
from magnet_amd.magnet import depth_sampling
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import device_inputs

wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
inp = device_inputs(wl, B, 1000, dev)
k = depth_sampling(3, wl.D)
out = torch.empty(B, wl.D, wl.h, wl.w, device=dev)
for fdt in ("bf16",):
    for name, path in (("cand", 2), ("cand noP2", 0x102), ("cand nogmm", 0x202), ("cand noP2 nogmm", 0x302), ("cand geom only", 0x802), ("cand geom only nogmm", 0xA02)):
        cv = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"],
                          inp["cam_intrins"], 5, feat_dtype=fdt, path=path)
        stats = torch.zeros(4, dtype=torch.int32, device=dev)
        cv(ref_gmm=inp["ref_gmms"], k_list=k, out=out, stats=stats)
        torch.cuda.synchronize()
        n = 3 if path == 1 else 10
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            cv(ref_gmm=inp["ref_gmms"], k_list=k, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        st = stats.cpu().tolist()
        gbs = wl.algorithmic_bytes() * B / (ms * 1e-3) / 1e9 if fdt == wl.feat_dtype else float("nan")
        print(f"{wl.name} B={B} {fdt:5s} {name:12s}: {ms:8.3f} ms/launch  {1e3*ms/B:8.2f} us/frame  alg {gbs:7.1f} GB/s  "
              f"tiles wl/gen {st[0]}/{st[1]} items {st[2]} ({st[2]/max(1,B*wl.hw*wl.V):.2f} per pixel-view)")
