#!/usr/bin/env python3
"""Dev tool (GPU box): socket power and shader clock (rocm-smi, sampled from a side thread) while the chip runs, for ~4 s each,
(a) the matcher alone, (b) the two convolution launches alone, (c) the layout packs alone, (d) the whole C2 step.
Evidence for DESIGN.md section 4.3 / 8: which parts of the step run at the chip's power cap.
usage: python tools/power_probe.py [frames]"""
import json
import os
import re
import subprocess
import sys
import threading
import time
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


def full_step():
    with torch.no_grad():
        model.match_and_refine(inp["ref_gmms"], inp["x_d3"], inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                               inp["is_valid"], inp["cam_intrins"], mode="test")


for _ in range(3):
    full_step()
torch.cuda.synchronize()
D, h, w = wl.D, wl.h, wl.w
gin_hi, gin_lo, ctot, Dp = model.gnet_input_buffer(B, h, w, dev)
g_stack, m_stack = model._stacks
rows, wp = B * (h + 2) * (w + 2), w + 2
work = model._work[(str(dev), B, h, w, ctot)]
pred0 = inp["ref_gmms"].float().contiguous(); pred1 = torch.empty_like(pred0)
outs = torch.empty((1, B, 2, 4 * h, 4 * w), dtype=torch.float32, device=dev)
cv = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], 5, feat_dtype="bf16")


def matcher():
    cv(ref_gmm=pred0, k_list=model.k_list, out_split=(gin_hi, gin_lo, ctot))


def convs():
    g_stack.run(gin_hi, gin_lo, ctot, rows, wp, work, n_var=D, inv_off=Dp, gauss=(pred0, pred1))
    m_stack.run(gin_hi[:, Dp:], gin_lo[:, Dp:], ctot, rows, wp, work.setdefault("mask", {}), upsample=(pred1.unsqueeze(0), outs))


def packs():
    lib.pack_features(inp["ref_feat"], lib.feat_enum("bf16"), pad=0)
    lib.pack_features(inp["nghbr_feat"], lib.feat_enum("bf16"), pad=1)
    lib.pack_split(inp["x_d3"], gin_hi, gin_lo, ctot, Dp)


def sample(stop, rows_):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            c = d[next(iter(d))]
            p = next((float(v) for k, v in c.items() if "power" in k.lower() and re.match(r"^[0-9.]+$", str(v))), None)
            sclk = next((v for k, v in c.items() if "sclk" in k.lower()), None)
            m = re.search(r"([0-9.]+)\s*Mhz", str(sclk), re.I)
            rows_.append((p, float(m.group(1)) if m else None))
        except Exception:
            pass
        time.sleep(0.1)


def run(name, fn, seconds=4.0):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    stop, got = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, got)); th.start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pw = [p for p, _ in got[2:] if p]; ck = [c for _, c in got[2:] if c]
    print(json.dumps({"load": name, "ms_per_call": round(1e3 * dt / n, 3), "samples": len(pw),
                      "socket_power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "socket_power_w_max": max(pw) if pw else None,
                      "sclk_mhz_mean": round(sum(ck) / len(ck)) if ck else None}), flush=True)


try:
    cap = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=5).stdout
    print(json.dumps({"rocm_smi_showmaxpower": json.loads(cap)}), flush=True)
except Exception as e:
    print(json.dumps({"rocm_smi_showmaxpower": f"unavailable ({type(e).__name__})"}))
run("idle (host sleeps)", lambda: time.sleep(0.05), 2.0)
run("matcher alone", matcher)
run("layout packs alone", packs)
run("both convolution launches alone", convs)
run("whole C2 step", full_step)
