"""Time the F-Net (row N3) on one GPU at the reference's inference shape (480x640 ScanNet images, 5 images per
reference frame): matrix-core path (FNetMFMA, output in the matcher's layouts) vs the same torch module on MIOpen
(fp32, NCHW) + the pack pass it needs.  One JSON line.  --profile-layers prints per-layer time / TFLOP/s."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import fnet, lib  # noqa: E402


def conv_flops(psm, H, W):
    """Multiply-add flops (x2) of every convolution of one image."""
    tot = 0.0
    x = torch.zeros(1, 3, H, W)
    hooks = []

    def hook(m, i, o):
        nonlocal tot
        tot += 2.0 * o.numel() * m.in_channels * m.kernel_size[0] * m.kernel_size[1]
    for m in psm.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        psm(x)
    for h in hooks:
        h.remove()
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8, help="reference frames per step (x5 images)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--feat-dtype", default="bf16")
    ap.add_argument("--skip-torch", action="store_true")
    ap.add_argument("--profile-layers", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="one stream for the whole batch (default: two half-batches on two streams)")
    ap.add_argument("--parts", type=int, default=0, help="dev: number of part-batches / streams (0 = the runner's default)")
    ap.add_argument("--dev-lib", action="store_true", help="bind to libmagnet_hip_dev.so (MAGNET_CONV_VARIANT A/B)")
    a = ap.parse_args()
    if a.dev_lib:
        lib.use_dev_build()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    psm = fnet.PSMNet(feature_dim=64).eval()
    flops_img = conv_flops(psm, a.height, a.width)
    psm = psm.to(dev)
    N = 5 * a.frames
    img = torch.randn(N, 3, a.height, a.width, device=dev)
    run = fnet.FNetMFMA(psm)
    if a.parts:
        run.split_parts = a.parts
    if a.no_split or a.profile_layers:                     # (per-layer events of two overlapping chains would not add up to the wall time)
        run.split_min_images = 10 ** 9

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps

    dt = timed(lambda: run.run(img, n_ref=a.frames, feat_dtype=a.feat_dtype))
    rec = {"workload": f"F-Net PSMNet {a.height}x{a.width}, {N} images ({a.frames} ref frames x 5)", "ms_mfma": dt * 1e3,
           "images_per_s_mfma": N / dt, "gflop_per_image": flops_img / 1e9, "tflops_fp32_equiv_mfma": flops_img * N / dt / 1e12,
           "out": f"matcher layouts, {a.feat_dtype}", "streams": 1 if run.split_min_images > N else run.split_parts}
    if not a.skip_torch:
        fe = lib.feat_enum(a.feat_dtype)

        def torch_path():
            with torch.no_grad():
                f = psm(img)
            lib.pack_features(f[:a.frames].contiguous(), fe, pad=0); lib.pack_features(f[a.frames:].contiguous(), fe, pad=1)
        dtt = timed(torch_path)
        rec.update({"ms_torch_miopen_fp32": dtt * 1e3, "images_per_s_torch": N / dtt, "speedup": dtt / dt})
    print(json.dumps(rec))
    if a.profile_layers:
        sink = fnet.FNetMFMA.event_sink = []
        run.run(img, n_ref=a.frames, feat_dtype=a.feat_dtype)
        torch.cuda.synchronize()
        fnet.FNetMFMA.event_sink = None
        names = [k for k in run._packed if k != "stem"]
        tot = 0.0
        for i, (e0, e1, fl) in enumerate(sink):
            ms = e0.elapsed_time(e1); tot += ms
            print(f"  conv {i:3d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s fp32-equiv")
        print(f"  conv launches total {tot:.2f} ms of {dt * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
