#!/usr/bin/env python3
"""Dev tool (GPU box): how long after matrix-core work does the matcher get its own clock back?

Round-5 record: the production matcher takes 0.80 ms alone and warm (2.39 GHz) and 0.866 ms inside the C2 step (2.04 GHz), although
0.9 ms of HBM-bound pack launches already sit between the previous step's mask head and the matcher.  This tool runs the contract's
step back to back and inserts an idle gap (a one-block spin kernel: the chip is busy, its power is not) directly in front of every
matcher launch; the matcher's HIP-event time per gap shows the recovery time of the chip's power management.  It answers whether any
ORDER of the step's launches could give the matcher the alone-warm time: only if the recovery is shorter than what can be put in between.

usage: python tools/clock_recovery.py [frames]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import synth, lib
from magnet_amd.homography import CostVolumeCW
from magnet_amd.magnet import MAGNET
from bench import device_inputs, make_args, _NoBackbone

wl = synth.WORKLOADS["C2"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
lib.load()
torch.manual_seed(1234)
model = MAGNET(make_args(wl, 1), d_net=_NoBackbone(), f_net=_NoBackbone(), feat_dtype="bf16").to(dev).eval()
inp = device_inputs(wl, B, 1000, dev)


def step():
    with torch.no_grad():
        model.match_and_refine(inp["ref_gmms"], inp["x_d3"], inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                               inp["is_valid"], inp["cam_intrins"], mode="test")


# calibrate torch.cuda._sleep (spin cycles -> ms)
for _ in range(3):
    torch.cuda._sleep(1_000_000)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 20_000_000 / e0.elapsed_time(e1)

orig_call = CostVolumeCW.__call__
gap_cycles = 0


def patched(self, *a, **k):
    if gap_cycles:
        torch.cuda._sleep(int(gap_cycles))
    return orig_call(self, *a, **k)


CostVolumeCW.__call__ = patched
for _ in range(60):                                   # warm: the clock settles over the first ~0.3 s of load
    step()
torch.cuda.synchronize()
rows = []
for gap_ms in (0.0, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0, 0.0):
    gap_cycles = gap_ms * cyc_per_ms
    ev = []
    for _ in range(5):
        step()
    CostVolumeCW.event_sink = ev
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0.record()
    n = 30
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    CostVolumeCW.event_sink = None
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    rows.append({"gap_ms_before_matcher": gap_ms, "matcher_ms_median": round(ms[len(ms) // 2], 4), "matcher_ms_min": round(ms[0], 4),
                 "step_ms_minus_gap": round(t0.elapsed_time(t1) / n - gap_ms, 4)})
    print(json.dumps(rows[-1]), flush=True)

# the matcher alone, warm (the reference point: no matrix-core work anywhere near it)
CostVolumeCW.__call__ = orig_call
from magnet_amd.magnet import depth_sampling
cv = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], 5, feat_dtype="bf16")
ld = (wl.D + 7) // 8 * 8 + 256
hi = torch.zeros(B * (wl.h + 2) * (wl.w + 2), ld, dtype=torch.bfloat16, device=dev); lo = torch.zeros_like(hi)
k = depth_sampling(3, wl.D)
for _ in range(300):
    cv(ref_gmm=inp["ref_gmms"], k_list=k, out_split=(hi, lo, ld))
torch.cuda.synchronize()
e0.record()
for _ in range(100):
    cv(ref_gmm=inp["ref_gmms"], k_list=k, out_split=(hi, lo, ld))
e1.record(); torch.cuda.synchronize()
print(json.dumps({"matcher_alone_warm_ms": round(e0.elapsed_time(e1) / 100, 4), "sleep_cycles_per_ms": round(cyc_per_ms)}))
