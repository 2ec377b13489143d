#!/usr/bin/env python3
"""Dev tool (GPU box): can the step's two halves run side by side on disjoint CU partitions?

  stage A = layout packs + quad-form map + matcher        (HBM- / vector-issue-bound, matrix pipe idle)
  stage B = G-Net + mask-head convolution launches          (matrix-pipe- / power-bound, 1 workgroup per CU, all registers)
The two cannot co-reside on a CU (each conv workgroup owns the CU's register file: profiles/r6/overlap_kernel_trace_tail.txt), but HIP
streams can be confined to CU subsets (hipExtStreamCreateWithCUMask; on gfx942/gfx950 mask bit i is a CU of XCC i % 8, spread over its
shader engines).  This probe times stage A and stage B alone on all CUs, alone on their partitions, and concurrently, for several splits.
usage: python tools/cu_partition_probe.py [frames]"""
import ctypes
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
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(lo, hi, ncu=256):
    words = (ncu + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for i in range(lo, hi):
        m[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(words), m)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


torch.manual_seed(1234)
model = MAGNET(make_args(wl, 1), d_net=_NoBackbone(), f_net=_NoBackbone(), feat_dtype="bf16").to(dev).eval()
inp = device_inputs(wl, B, 1000, dev)
with torch.no_grad():
    for _ in range(3):
        model.match_and_refine(inp["ref_gmms"], inp["x_d3"], inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                               inp["is_valid"], inp["cam_intrins"], mode="test")
torch.cuda.synchronize()
D, h, w = wl.D, wl.h, wl.w
gin_hi, gin_lo, ctot, Dp = model.gnet_input_buffer(B, h, w, dev)
g_stack, m_stack = model._stacks
rows, wp = B * (h + 2) * (w + 2), w + 2
work = model._work[(str(dev), B, h, w, ctot)]
pred0 = inp["ref_gmms"].float().contiguous()
pred1 = torch.empty_like(pred0)
outs = torch.empty((1, B, 2, 4 * h, 4 * w), dtype=torch.float32, device=dev)
# stage A writes its own buffer (as a double-buffered pipeline would): timing only, nothing reads it
gin2_hi, gin2_lo = torch.zeros_like(gin_hi), torch.zeros_like(gin_lo)


def stage_a():
    cv = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], 5, feat_dtype="bf16")
    lib.pack_split(inp["x_d3"], gin2_hi, gin2_lo, ctot, Dp)
    cv(ref_gmm=pred0, k_list=model.k_list, out_split=(gin2_hi, gin2_lo, ctot))


def stage_b():
    g_stack.run(gin_hi, gin_lo, ctot, rows, wp, work, n_var=D, inv_off=Dp, gauss=(pred0, pred1))
    m_stack.run(gin_hi[:, Dp:], gin_lo[:, Dp:], ctot, rows, wp, work.setdefault("mask", {}), upsample=(pred1.unsqueeze(0), outs))


def timed(fn, stream, n=20, warm=8):
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
    return e0, e1, n


def ms(t):
    e0, e1, n = t
    return e0.elapsed_time(e1) / n


# second question (PROBE_PACKS=1): only the HBM-bound layout packs of the NEXT batch on the small partition, matcher + convolutions of this
# batch on the large one
if os.environ.get("PROBE_PACKS"):
    from magnet_amd.magnet import depth_sampling
    cvm = CostVolumeCW(inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"], inp["cam_intrins"], 5, feat_dtype="bf16")

    def stage_a():                                           # the packs alone (features x 2, x_d3)
        lib.pack_features(inp["ref_feat"], lib.feat_enum("bf16"), pad=0)
        lib.pack_features(inp["nghbr_feat"], lib.feat_enum("bf16"), pad=1)
        lib.pack_split(inp["x_d3"], gin2_hi, gin2_lo, ctot, Dp)

    _conv_b = stage_b

    def stage_b():                                           # quad-form map + matcher + both convolution launches
        cvm._gmm_quad = None
        cvm(ref_gmm=pred0, k_list=model.k_list, out_split=(gin_hi, gin_lo, ctot))
        _conv_b()

full = torch.cuda.Stream(device=dev)
ta = timed(stage_a, full); torch.cuda.synchronize(); tb = timed(stage_b, full); torch.cuda.synchronize()
print(json.dumps({"partition": "none (256 CUs each, one after the other)", "stage_a_ms": round(ms(ta), 3), "stage_b_ms": round(ms(tb), 3), "sum_ms": round(ms(ta) + ms(tb), 3)}), flush=True)
# PROBE_FULL=1: both streams may use ALL 256 CUs (two separate hardware queues, no partition): what the dispatcher makes of the two kernels'
# workgroups when nothing but CU resources keeps them apart
for nb in ((256,) if os.environ.get("PROBE_FULL") else (248, 240, 232, 224, 208, 192) if os.environ.get("PROBE_PACKS") else (240, 224, 208, 192, 176, 160)):
    sb_, sa_ = masked_stream(0, nb), masked_stream(nb if nb < 256 else 0, 256)
    tb = timed(stage_b, sb_); torch.cuda.synchronize()
    ta = timed(stage_a, sa_, n=6, warm=2); torch.cuda.synchronize()
    alone_a, alone_b = ms(ta), ms(tb)
    # concurrently: both streams run until stage B has done its 30 launches; stage A loops beside it
    n_b = 30
    with torch.no_grad():
        eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_a = max(2, int(n_b * alone_b / alone_a * 1.15) + 1)          # enough stage-A work to cover stage B's whole run
        with torch.cuda.stream(sa_):
            for _ in range(2):
                stage_a()
        with torch.cuda.stream(sb_):
            for _ in range(5):
                stage_b()
        torch.cuda.synchronize()
        eb0.record(sb_); ea0.record(sa_)
        # interleave the host-side launches so that neither queue starves
        ia = ib = 0
        while ia < n_a or ib < n_b:
            if ib < n_b:
                with torch.cuda.stream(sb_):
                    stage_b()
                ib += 1
            if ia < n_a and ia * n_b <= ib * n_a:
                with torch.cuda.stream(sa_):
                    stage_a()
                ia += 1
        eb1.record(sb_); ea1.record(sa_)
        torch.cuda.synchronize()
    both_b, both_a = eb0.elapsed_time(eb1) / n_b, ea0.elapsed_time(ea1) / n_a
    print(json.dumps({"conv_cus": nb, "other_cus": 256 - nb, "stage_b_alone_ms": round(alone_b, 3), "stage_a_alone_ms": round(alone_a, 3),
                      "stage_b_concurrent_ms": round(both_b, 3), "stage_a_concurrent_ms": round(both_a, 3),
                      "pipelined_step_ms": round(max(both_a, both_b), 3)}), flush=True)
