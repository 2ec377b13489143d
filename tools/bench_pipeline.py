"""Software-pipelined batches (C2 shapes): the layout packs of batch i + 1 on a side stream beside the matcher / convolution kernels
of batch i (two buffer sets, events where the chains meet).  Every step does one full pack set and one full refinement; prints the
sequential and the pipelined ms per step.  `--dev-lib` + MAGNET_PACK_NARROW=1 selects the 64-pixel pack kernels (small footprint)."""
import argparse
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import lib, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dev-lib", action="store_true")
    a = ap.parse_args()
    if a.dev_lib:
        lib.use_dev_build()
    from bench import device_inputs, make_args, _NoBackbone
    from magnet_amd.magnet import MAGNET
    dev = torch.device("cuda:0")
    wl = synth.WORKLOADS["C2"]
    B = a.frames
    torch.manual_seed(1234)
    model = MAGNET(make_args(wl, wl.iters), d_net=_NoBackbone(), f_net=_NoBackbone(), feat_dtype="bf16").to(dev).eval()
    inp = device_inputs(wl, B, 1000, dev)
    fe = lib.feat_enum("bf16")

    def seq_step():
        with torch.no_grad():
            model.match_and_refine(inp["ref_gmms"], inp["x_d3"], inp["ref_feat"], inp["nghbr_feat"], inp["nghbr_gmms"], inp["nghbr_poses"],
                                   inp["is_valid"], inp["cam_intrins"], mode="test")
    for _ in range(5):
        seq_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        seq_step()
    torch.cuda.synchronize(); seq_ms = 1e3 * (time.perf_counter() - t0) / a.steps

    models = [model, copy.deepcopy(model)]
    sets = []
    for m in models:
        gh, gl, ctot, coff = m.gnet_input_buffer(B, wl.h, wl.w, dev)
        sets.append(dict(ref=lib.pack_features(inp["ref_feat"], fe, pad=0), src=lib.pack_features(inp["nghbr_feat"], fe, pad=1), gh=gh, gl=gl, ctot=ctot, coff=coff))
    side = torch.cuda.Stream(device=dev)
    ev_packed = [torch.cuda.Event(), torch.cuda.Event()]
    ev_done = [torch.cuda.Event(), torch.cuda.Event()]
    state = {"i": 0}

    def pack_set(k):
        s_ = sets[k]
        lib.pack_features(inp["ref_feat"], fe, pad=0, out=s_["ref"]); lib.pack_features(inp["nghbr_feat"], fe, pad=1, out=s_["src"])
        lib.pack_split(inp["x_d3"], s_["gh"], s_["gl"], s_["ctot"], s_["coff"])

    def pipe_step():
        i = state["i"]; k = i & 1
        main = torch.cuda.current_stream(dev)
        if i == 0:
            pack_set(k); side.wait_stream(main)
        else:
            main.wait_event(ev_packed[k]); side.wait_event(ev_done[1 - k])
        with torch.cuda.stream(side):
            pack_set(1 - k); ev_packed[1 - k].record(side)
        with torch.no_grad():
            models[k].match_and_refine(inp["ref_gmms"], None, None, None, inp["nghbr_gmms"], inp["nghbr_poses"], inp["is_valid"],
                                       inp["cam_intrins"], mode="test", packed_feats=(sets[k]["ref"], sets[k]["src"]), x_d3_in_place=True)
        ev_done[k].record(main)
        state["i"] = i + 1
    for _ in range(4):
        pipe_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        pipe_step()
    torch.cuda.synchronize(); pipe_ms = 1e3 * (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": "C2", "frames_per_step": B, "ms_per_step_sequential": seq_ms, "ms_per_step_pipelined_batches": pipe_ms,
                      "pack_kernels": "narrow (MAGNET_PACK_NARROW)" if os.environ.get("MAGNET_PACK_NARROW") else "wide (default)"}))


if __name__ == "__main__":
    main()
