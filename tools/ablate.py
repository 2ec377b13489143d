#!/usr/bin/env python3
"""Dev tool (GPU box): time the fused cost-volume kernel alone under kernel selection / dev switches (`path`).
usage: python tools/ablate.py [workload] [frames] [split]      (split: time the split-bf16 channel-last output form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magnet_amd import synth, lib
lib.use_dev_build()
from magnet_amd.homography import CostVolumeCW
from magnet_amd.magnet import depth_sampling
from bench import device_inputs

wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
split = len(sys.argv) > 3 and sys.argv[3] == "split"
dev = torch.device("cuda:0")
inp = device_inputs(wl, B, 1000, dev)
if os.environ.get("ABLATE_SMOOTH"):                      # smooth synthetic variant (synth.make_inputs smooth=): low-passed features and (mu, sigma) maps
    sm = synth.make_inputs(wl, B=B, seed=1000, smooth=int(os.environ["ABLATE_SMOOTH"]), round_bf16=(wl.feat_dtype == "bf16"))
    for k_ in ("ref_feat", "nghbr_feat", "ref_gmms", "nghbr_gmms"):
        inp[k_] = sm[k_].to(dev)
k = depth_sampling(3, wl.D)
out = torch.empty(B, wl.D, wl.h, wl.w, device=dev)
ld = (wl.D + 7) // 8 * 8 + 256
hi = torch.zeros(B * (wl.h + 2) * (wl.w + 2), ld, dtype=torch.bfloat16, device=dev); lo = torch.zeros_like(hi)
fdt = wl.feat_dtype
# needs a dev build of the library (python -m magnet_amd.build --dev): bits 8.. of `path` travel as MagnetCostVolumeArgs.dev_flags
R2 = 0x100 << 8                                       # dev flag 0x100: the round-2 production kernels although the quad map is given
M4, M8, NP2, NP3, NP4 = 0x2000 << 8, 0x1000 << 8, 0x4000 << 8, 0x40000 << 8, 0x80000 << 8
VG4, VG1, M7 = 0x400000 << 8, 0x800000 << 8, 0x10000 << 8
V4 = 0x8 << 8                                         # dev flag 0x8: round 4's LDS-staged / matrix-pipe experiment (cost_volume_v4.hip)
V3 = 0                                                # production = cost_volume_v3.hip (round 4: scalar frame bases, buffer-addressed quad loads)
V5 = 0x20 << 8                                        # dev flag 0x20: round 4's quad-prefetch experiment (cost_volume_v5.hip)
VARIANTS = [("production (auto) = cost_volume_v3.hip", 0), ("v5 experiment (quads prefetched by LDS-DMA)", V5), ("production without dot products (timing only)", 0x200 << 8),
            ("round-2 kernel (fast64)", 4 | R2), ("exact cand", 2), ("production (auto), again", 0), ("v5 experiment, again", V5), ("v4 experiment (LDS staging + MFMA correlation)", V4)]
if os.environ.get("ABLATE_V4"):
    VARIANTS = [("v4 experiment", V4), ("v4, 16 slots per round (6 workgroups / CU)", V4 | (0x1000 << 8)), ("v4 capped at 4 workgroups / CU", V4 | (0x100000 << 8)),
                ("v4 capped at 3 workgroups / CU", V4 | (0x200000 << 8)), ("production (cost_volume_v3.hip)", 0)]
if os.environ.get("ABLATE_V3"):
    VARIANTS = [("v3 (2 views in flight)", V3), ("v3, 4 views in flight", V3 | (0x400000 << 8)), ("v3, 4 views, compiled for 8 waves", V3 | (0x401000 << 8)),
                ("v3, 2 views, compiled for 8 waves", V3 | (0x1000 << 8)), ("v3, 3 correlation passes in flight", V3 | (0x40000 << 8)),
                ("v3, 4 correlation passes in flight", V3 | (0x80000 << 8)), ("v3 without dot products", V3 | (0x200 << 8)), ("v3 again", V3)]
if os.environ.get("ABLATE_TX"):                        # round 5: texel-pair items against the quad items of rounds 2 - 4 (dev flag 0x400), same box
    QI = 0x400 << 8
    VARIANTS = [("production (auto)", 0), ("same kernel, quad items (rounds 2 - 4)", QI), ("production (auto), again", 0), ("quad items, again", QI)]
    if wl.D > 32 and wl.w <= 512:
        VARIANTS += [("batched-view kernel (fast64), pair items", R2), ("batched-view kernel (fast64), quad items", R2 | QI)]
if os.environ.get("ABLATE_PX2"):                       # round 5: two pixels per correlation batch in cost_volume_v3.hip (product; dev flag 0x10 = the one-pixel loop; split output only)
    PX1 = 0x10 << 8
    VARIANTS = [("production (two pixels per batch, 8 waves)", 0), ("one pixel per batch (rounds 3 - 4)", PX1), ("production, again", 0), ("one pixel per batch, again", PX1)]
    if split:                                          # bit-identity of the two forms on this workload
        outs = []
        for path in (0, PX1):
            cvx = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], 5, feat_dtype=fdt, path=path)
            hi.zero_(); lo.zero_()
            cvx(ref_gmm=inp["ref_gmms"], k_list=k, out_split=(hi, lo, ld))
            torch.cuda.synchronize()
            outs.append((hi.clone(), lo.clone()))
        print(f"{wl.name}: two-pixel batches bit-identical to the one-pixel loop: {torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])}")
HALFQ = 0x80 << 8
if os.environ.get("ABLATE_HALFQ"):                     # round 6: what the quad-form (mu, sigma) map's bytes cost cost_volume_v3.hip's product instance: the same map as 8 fp16 per
    VARIANTS = [("production (fp32 quad form, 32 B)", 0), ("fp16 quad form, 16 B per candidate", HALFQ),   # entry (16-byte stride), one load per candidate; gates differ at the 1e-3 level
                ("production, again", 0), ("fp16 quad form, again", HALFQ)]
if os.environ.get("ABLATE_SHORT"):
    VARIANTS = VARIANTS[:2] if os.environ.get("ABLATE_TX") else [VARIANTS[0], VARIANTS[1], VARIANTS[5], VARIANTS[6]]
_prod = None
for name, path in VARIANTS:
    if split and (path & 0xff) == 3:
        continue
    cv = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"],
                      inp["cam_intrins"], 5, feat_dtype=fdt, path=path)
    kw = dict(out_split=(hi, lo, ld)) if split else dict(out=out)
    try:
        cv(ref_gmm=inp["ref_gmms"], k_list=k, **kw)
    except lib.MagnetError as e:
        print(f"{wl.name} {name}: {e}"); continue
    if path & HALFQ and os.environ.get("ABLATE_HALFQ"):   # rewrite the packed map in place: entry e = 8 fp16 at byte 16 e (the front half of the buffer)
        q16 = cv._gmm_quad.reshape(-1).to(torch.float16)
        cv._gmm_quad.view(torch.float16).reshape(-1)[:q16.numel()] = q16
        cv(ref_gmm=inp["ref_gmms"], k_list=k, **kw)
        torch.cuda.synchronize()
        if split and _prod is not None:                  # same amount of work: the cost channels differ where a gate flipped, nowhere else by more than rounding
            a, b = hi[:, :wl.D].float(), _prod
            print(f"   fp16 map vs fp32 map: {float((a != b).float().mean()):.4f} of the cost entries differ, {float(((a - b).abs() > 0.05 * b.abs().clamp_min(1.0)).float().mean()):.5f} by more than 5 %; "
                  f"non-zero entries {float((a != 0).float().mean()):.4f} vs {float((b != 0).float().mean()):.4f}")
    elif split and os.environ.get("ABLATE_HALFQ"):
        torch.cuda.synchronize(); _prod = hi[:, :wl.D].float().clone()
    # the chip's clock / power state drifts for the first second of load (20-launch samples differed by 8 % between the first and
    # the last variant of one process): ~0.3 s of the same kernel first, then the median of 5 samples of 40 launches
    for _ in range(300):
        cv(ref_gmm=inp["ref_gmms"], k_list=k, **kw)
    torch.cuda.synchronize()
    n, samples = 40, []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            cv(ref_gmm=inp["ref_gmms"], k_list=k, **kw)
        e1.record(); torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / n)
    ms = sorted(samples)[2]
    gbs = wl.algorithmic_bytes() * B / (ms * 1e-3) / 1e9
    print(f"{wl.name} B={B} {fdt:5s} {'split' if split else 'nchw '} {name:30s}: {ms:8.3f} ms/launch (5 samples {min(samples):.3f}..{max(samples):.3f})  alg {gbs:7.1f} GB/s = {gbs / 80:5.1f} % of 8 TB/s")
