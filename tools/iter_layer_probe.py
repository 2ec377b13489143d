"""Probe (GPU box): the per-iteration G-Net launch of the shipped configuration (D = 5, 47 frames, hoisted x_d3 part as addend) as it is —
3x3 over a 32-channel slice (5 real channels) — against an im2col'd 1x1 form with K = 96 (9 taps x 8 channels), same fused tail, same
addend.  Timing only (random operands): is the tail-dominated launch worth re-forming?"""
import sys, torch
sys.path.insert(0, ".")
from magnet_amd import lib

dev = torch.device("cuda:0")
B, h, w = 47, 120, 160
wp = w + 2
rows = B * (h + 2) * wp
g = torch.Generator().manual_seed(0)
def planes(r, c):
    x = torch.randn((r, c), generator=g).to(dev)
    hi = x.to(torch.bfloat16); return hi, (x - hi.float()).to(torch.bfloat16)
def wts(t, co, ci):
    x = (torch.randn((t, co, ci), generator=g) * 0.05).to(dev)
    hi = x.to(torch.bfloat16); return hi, (x - hi.float()).to(torch.bfloat16)
addend = torch.randn((rows, 128), generator=g).to(dev)
tw = wts(1, 128 + 128 + 16, 128); tail = (tw[0].reshape(-1).contiguous(), tw[1].reshape(-1).contiguous(), torch.zeros(272, device=dev), 16)
bias = torch.zeros(128, device=dev)
out = torch.empty((rows, 16), dtype=torch.float32, device=dev)
gi = torch.rand((B, 2, h, w), device=dev) + 0.5; go = torch.empty_like(gi)
res = {}
for name, taps, cin in (("3x3, K = 9 x 32 (today)", 9, 32), ("1x1, K = 96 (im2col)", 1, 96), ("1x1, K = 64", 1, 64)):
    xh, xl = planes(rows + 2 * wp + 8, cin); wh, wl = wts(taps, 128, cin)
    off = wp + 1
    def run():
        lib.conv_mfma(xh[off:], xl[off:], cin, cin, wh, wl, bias, taps, wp, True, rows, addend=addend, tail=tail, gauss=(gi, go))
    for _ in range(20): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    print(f"{name:28s}: {sorted(ts)[2]:.4f} ms", flush=True)
