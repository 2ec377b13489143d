"""Two independent C2 batches on two HIP streams (two models, two buffer sets): do kernels of different batches share the chip usefully?
The matcher is a vector-ALU / address-unit kernel at 56 registers, the convolutions are matrix-pipe kernels that leave ~80 registers per
SIMD lane and 13 - 29 KB of LDS free: a matcher workgroup can co-reside with a G-Net convolution workgroup.  Prints frames/s of one
stream, of two streams, and per-stream step times.  Not the contract line: a step of one batch is no longer one isolated pass."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magnet_amd import lib, synth

def main():
    from bench import device_inputs, make_args, _NoBackbone
    from magnet_amd.magnet import MAGNET
    dev = torch.device("cuda:0")
    wl = synth.WORKLOADS["C2"]
    B, K = 64, 20
    torch.manual_seed(1234)
    model = MAGNET(make_args(wl, wl.iters), d_net=_NoBackbone(), f_net=_NoBackbone(), feat_dtype="bf16").to(dev).eval()
    models = [model, copy.deepcopy(model)]
    inps = [device_inputs(wl, B, 1000 + k, dev) for k in range(2)]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def step(k):
        i = inps[k]
        with torch.no_grad():
            models[k].match_and_refine(i["ref_gmms"], i["x_d3"], i["ref_feat"], i["nghbr_feat"], i["nghbr_gmms"], i["nghbr_poses"],
                                       i["is_valid"], i["cam_intrins"], mode="test")

    def run_one(n):
        for _ in range(n): step(0)

    def run_two(n, offset):
        main = torch.cuda.current_stream(dev)
        for s in streams: s.wait_stream(main)
        for it in range(n):
            for k in (0, 1):
                with torch.cuda.stream(streams[k]):
                    if it == 0 and k == 1 and offset: torch.cuda._sleep(int(offset * 2.1e6))
                    step(k)
        for s in streams: main.wait_stream(s)

    for fn, label, per in ((lambda: run_one(K), "one stream", 1), (lambda: run_two(K, 0.0), "two streams", 2), (lambda: run_two(K, 2.5), "two streams, second 2.5 ms late", 2),
                           (lambda: run_one(K), "one stream again", 1)):
        run_one(3) if per == 1 else run_two(3, 0.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{label:34s}: {per * K * B / dt:9.1f} frames/s  ({1e3 * dt / (per * K):.3f} ms per batch)", flush=True)

if __name__ == "__main__":
    main()
