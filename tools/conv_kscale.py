"""The 128-wide 3x3 matrix-core convolution at several input widths on the C2 grid (64 frames, 122 x 162 padded rows): time per launch is
fixed cost per tile (prologue DMA latency, epilogue, wave quantisation) + slope x K.  Prints the fit; the fixed part is what a persistent
workgroup with a cross-tile prologue could hide."""
import sys, torch
sys.path.insert(0, ".")
import os
from magnet_amd import lib
if os.environ.get("CONV_LIB"):
    lib.LIB_PATH = os.path.abspath(os.environ["CONV_LIB"])      # an ablation build of tools/build_conv_abl.sh
elif os.environ.get("CONV_DEV_LIB"):
    lib.use_dev_build()      # MAGNET_CONV_VARIANT=4096: the persistent-workgroup form (dev)

def main():
    dev = torch.device("cuda:0")
    B, hp, wp = 64, 122, 162
    rows = B * hp * wp
    g = torch.Generator(device="cpu").manual_seed(1)
    res = []
    for cin in (64, 128, 256, 512):
        x = torch.randn((rows + 2 * wp + 8, cin), generator=g).to(dev)
        xh = x.to(torch.bfloat16); xl = (x - xh.float()).to(torch.bfloat16)
        w = (torch.randn((9, 128, cin), generator=g) * 0.05).to(dev)
        wh = w.to(torch.bfloat16); wl = (w - wh.float()).to(torch.bfloat16)
        bias = torch.zeros(128, device=dev)
        out = torch.empty((rows, 128), dtype=torch.float32, device=dev)
        off = wp + 1                                              # row 0 of the launch = an interior offset of the buffer
        def run():
            lib.conv_mfma(xh[off:], xl[off:], cin, cin, wh, wl, bias, 9, wp, False, rows - 2 * off, out_f32=out)
        for _ in range(30): run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        t = sorted(ts)[2]
        tf = 2.0 * (rows - 2 * off) * 128 * 9 * cin / t / 1e9
        print(f"cin {cin:4d}: {t:.4f} ms  {tf:.1f} TFLOP/s fp32-equiv", flush=True)
        res.append((cin, t))
    (c0, t0), (c1, t1) = res[1], res[3]
    slope = (t1 - t0) / (c1 - c0)
    fixed = t0 - slope * c0
    print(f"fit on cin 128 / 512: {slope * 32 * 1e3:.2f} us per 32-channel group of 9 taps, fixed {fixed:.4f} ms "
          f"(= {100 * fixed / res[2][1]:.1f} % of the cin = 256 launch)")

if __name__ == "__main__":
    main()
