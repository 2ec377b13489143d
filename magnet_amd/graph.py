"""HIP-graph replay of the refinement loop: the host leaves the loop.

At 64 frames per step the path is GPU-bound (5.8 ms of kernels, ~15 device operations): replaying the captured step still saves the
inter-operation gaps — 5.78 -> 5.69 ms per C2 step, same box (round 6; bench.py replays the step this way by default, --no-graph: eager).  The reference's own evaluation loop feeds ONE
frame at a time (test_MaGNet.py:166-170, batch size 1).  `GraphedRefine` captures `MAGNET.match_and_refine` for a fixed shape
into a HIP graph once (torch.cuda.CUDAGraph: our kernels are launched on torch's capture stream, so they are recorded like
torch's own) and replays it per frame: one launch from the host's point of view.  Measured on MI355X (bench.py --frames 1
[--graph]): 0.386 -> 0.356 ms per one-frame step (2 590 -> 2 810 frames/s), 4 frames 0.839 -> 0.823 ms: the small-batch step is
bound by its serial chain of ~13 kernels that each under-fill the chip (one frame = 155 convolution tiles for 256 CUs), not by
launch overhead — batching frames (64 per step: 7 500 frames/s) is what pays; the graph only removes the host from the loop.  Inputs are copied into the graph's static tensors (device-to-device,
a few MB); outputs are the graph's static output tensors (clone them to keep a result across replays)."""
from __future__ import annotations

import torch

from . import lib


class GraphedRefine:
    def __init__(self, model, ref_gmms, x_d3, ref_feat, nghbr_feat, nghbr_gmms, nghbr_poses, is_valid, cam_intrins, mode="test",
                 warmup: int = 3):
        if not ref_feat.is_cuda:
            raise lib.MagnetError("GraphedRefine: tensors must be on the GPU (no CPU fallback)")
        dev = ref_feat.device
        self.model, self.mode = model, mode
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
        self.static = [f32(t) for t in (ref_gmms, x_d3, ref_feat, nghbr_feat, nghbr_gmms, nghbr_poses)]
        # validity and intrinsics live in static device tensors too, so they can change between replays
        self.is_valid = is_valid.detach().to(device=dev, dtype=torch.int32).contiguous().clone()
        self.cam = {k: v.detach().to(device=dev, dtype=torch.float32).contiguous().clone() for k, v in cam_intrins.items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():                  # warm-up outside capture: workspaces, weight packs, attributes
            for _ in range(max(1, warmup)):
                model.match_and_refine(*self.static, self.is_valid, self.cam, mode=mode)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: another thread of the process that polls events meanwhile (torch.distributed's RCCL watchdog, once
        # a process group exists — bench.py always has one) must not invalidate this thread's capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"), torch.no_grad():
            self.out = model.match_and_refine(*self.static, self.is_valid, self.cam, mode=mode)

    def __call__(self, ref_gmms, x_d3, ref_feat, nghbr_feat, nghbr_gmms, nghbr_poses, is_valid=None, cam_intrins=None):
        for s, t in zip(self.static, (ref_gmms, x_d3, ref_feat, nghbr_feat, nghbr_gmms, nghbr_poses)):
            if t is not s:
                s.copy_(t, non_blocking=True)
        if is_valid is not None:
            self.is_valid.copy_(is_valid.to(torch.int32), non_blocking=True)
        if cam_intrins is not None:
            for k in self.cam:
                self.cam[k].copy_(cam_intrins[k], non_blocking=True)
        self.graph.replay()
        return self.out
