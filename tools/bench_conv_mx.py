#!/usr/bin/env python3
"""Dev tool (GPU box): the 128-wide 3x3 layer + fused tail at the bench shape, bf16x3 operand format vs the fp16 + e4m3 one (ABI v302)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from magnet_amd import lib
from magnet_amd.convnet import ConvStackMFMA
if os.environ.get("CONV_LIB"):
    lib.LIB_PATH = os.path.abspath(os.environ["CONV_LIB"])      # a build of tools/build_conv_abl.sh
elif os.environ.get("CONV_DEV_LIB"):
    lib.use_dev_build()      # MAGNET_CONV_VARIANT then selects K-loop variants / timing ablations (4096: no correction MFMAs, 8192: no correction operand reads)
dev = torch.device("cuda:0")
B, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 120, 160
rows = B * (h + 2) * (w + 2)
for cin, cout in ((320, 2), (256, 144)):
    torch.manual_seed(cin)
    seq = nn.Sequential(nn.Conv2d(cin, 128, 3, padding=1), nn.ReLU(), nn.Conv2d(128, 128, 1), nn.ReLU(), nn.Conv2d(128, 128, 1), nn.ReLU(), nn.Conv2d(128, cout, 1)).to(dev).eval()
    st = ConvStackMFMA(seq)
    x = torch.randn(B, cin, h, w, device=dev) * 0.5
    hi = torch.zeros((rows, cin), dtype=torch.bfloat16, device=dev); lo = torch.zeros_like(hi)
    lib.pack_split(x, hi, lo, cin, 0)
    f16 = torch.zeros((rows, cin), dtype=torch.float16, device=dev); qr = torch.zeros((rows, cin), dtype=torch.int16, device=dev)
    sc = torch.zeros((cin // 32, rows), dtype=torch.int32, device=dev)
    lib.pack_mx(x, f16, qr, sc, cin, 0, rows)
    work = {}
    flops = 2.0 * rows * (128 * cin * 9 + 128 * (256 + (16 if cout == 2 else 144)))
    def timeit(fn, n=int(os.environ.get("CONV_N", 30))):
        for _ in range(int(os.environ.get("CONV_WARM", 40))): fn()
        torch.cuda.synchronize()
        s = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); s.append(e0.elapsed_time(e1) / n)
        return sorted(s)[2]
    t3 = timeit(lambda: st.run(hi, lo, cin, rows, w + 2, work))
    tm = timeit(lambda: st.run(f16, qr, cin, rows, w + 2, work, mx=(sc, rows)))
    t3b = timeit(lambda: st.run(hi, lo, cin, rows, w + 2, work))
    tp3 = timeit(lambda: lib.pack_split(x, hi, lo, cin, 0)); tpm = timeit(lambda: lib.pack_mx(x, f16, qr, sc, cin, 0, rows))
    print(f"conv {cin}->128->128->128->{cout}, {B} frames: bf16x3 {t3:.3f} ms ({flops / t3 / 1e9:.0f} TF)  fp16+e4m3 {tm:.3f} ms ({flops / tm / 1e9:.0f} TF)  bf16x3 again {t3b:.3f} ms;  pack_split {tp3:.3f} ms  pack_mx {tpm:.3f} ms")
