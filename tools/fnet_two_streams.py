"""Dev experiment (GPU box): does the F-Net run faster as TWO half-batches on two streams (two independent FNetMFMA instances, 20 images
each) than as one 40-image batch?  The convolution workgroups of one launch run in lockstep (equal tile times): every CU reaches its
epilogue at once and the stores arrive as one burst (conv_mfma.hip, epilogue comment).  Two launches of different layers co-resident on
the CUs would be out of phase with each other."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import fnet, lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
psm = fnet.PSMNet(feature_dim=64).eval().to(dev)
N, H, W = 40, 480, 640
img = torch.randn(N, 3, H, W, device=dev)
one = fnet.FNetMFMA(psm)
halves = [fnet.FNetMFMA(psm), fnet.FNetMFMA(psm)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
imgs = [img[:20].contiguous(), img[20:].contiguous()]

def run_one():
    one.run(img, n_ref=8, feat_dtype="bf16")

def run_two(skew_ms=0.0):
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            if k == 1 and skew_ms > 0: torch.cuda._sleep(int(skew_ms * 2.1e6))
            halves[k].run(imgs[k], n_ref=4, feat_dtype="bf16")
    for s in streams: main.wait_stream(s)

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print(f"one 40-image batch          : {timed(run_one):.2f} ms")
print(f"two 20-image batches, 2 streams: {timed(run_two):.2f} ms")
print(f"  ... second stream 0.1 ms late: {timed(lambda: run_two(0.1)):.2f} ms")
print(f"one 40-image batch again    : {timed(run_one):.2f} ms")
halves[0].run(imgs[0], n_ref=4, feat_dtype="bf16"); torch.cuda.synchronize()
print(f"one 20-image batch alone    : {timed(lambda: halves[0].run(imgs[0], n_ref=4, feat_dtype='bf16')):.2f} ms (x2 = sequential halves)")
