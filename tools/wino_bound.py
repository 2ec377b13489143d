#!/usr/bin/env python3
"""Dev tool (GPU box): an empirical UPPER bound on what a Winograd form of the 3x3 layers could save, measured with the existing
kernels on GEMMs of exactly the Winograd forms' matrix work (random operands; results are meaningless, launch times are not).

  direct     9 taps x 256 channels over R rows                  (the x_d3 part of either stack's first layer: what runs today)
  F(2,3) 1-D 12 products per output PAIR  = 4 taps x 768 channels over R/2 rows   (3 kernel rows x 4 Winograd positions x 256 channels)
  F(2x2,3x3) 16 products per 2x2 TILE      = 4 taps x 1024 channels over R/4 rows
The stand-ins keep ONE accumulator set (a real Winograd kernel needs 4 or 16 live sets per tile, i.e. smaller tiles and more weight
traffic per matrix instruction — DESIGN.md section 8.1), read their A operand through the 2x2-tap window loop, and do not pay the input
transform, its re-split into bf16 hi/lo, or the output transform: a real kernel can only be slower.
usage: python tools/wino_bound.py [frames]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import lib

lib.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = B * 122 * 162
g = torch.Generator(device=dev).manual_seed(1)


def planes(rows, c):
    x = torch.randn(rows, c, generator=g, device=dev) * 0.5
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


def run(name, rows, cin, taps, wp, n=30):
    a_hi, a_lo = planes(rows, cin)
    w_hi, w_lo = planes(taps * 128, cin)
    w_hi, w_lo = w_hi.view(taps, 128, cin), w_lo.view(taps, 128, cin)
    bias = torch.zeros(128, device=dev)
    out = torch.empty(rows, 128, device=dev)
    f = lambda: lib.conv_mfma(a_hi, a_lo, cin, cin, w_hi, w_lo, bias, taps, wp, False, rows, out_f32=out)
    for _ in range(15):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    macs = rows * 128 * cin * taps
    print(json.dumps({"form": name, "rows": rows, "cin": cin, "taps": taps, "ms": round(ms, 4), "bf16_mfma_tflops": round(3 * 2 * macs / ms / 1e9, 1),
                      "products_per_output_px": round(taps * cin * rows / R / 256, 3)}), flush=True)
    return ms


d = run("direct 3x3 over 256 channels (today)", R, 256, 9, 162)
w1 = run("F(2,3) along x: 12 products per output pair", R // 2, 768, 4, 81)
w2 = run("F(2x2,3x3): 16 products per 2x2 tile", R // 4, 1024, 4, 81)
d2 = run("direct, again", R, 256, 9, 162)
print(json.dumps({"upper_bound_saving_ms_per_stack": {"F(2,3) 1-D": round(min(d, d2) - w1, 3), "F(2x2,3x3)": round(min(d, d2) - w2, 3)},
                  "note": "before the input / output transforms, the bf16 re-split of the transformed input (VALU), 4 / 16 live accumulator sets, and "
                          "(1-D form, transform in the pack kernel) +0.22 ms for writing 2x the x_d3 bytes"}))
