"""bench.py --gpus N starts N ranks by itself (no torchrun): run the launcher at world size 2 on CPU (gloo, --dry-run: no
kernels) and parse the one JSON line rank 0 prints."""
import json
import os
import subprocess
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout    # the contract: ONE JSON line on stdout and nothing else (native
    return json.loads(lines[0])                                        # libraries' banners — RCCL prints one — are routed to stderr)


def test_bench_spawns_two_ranks_dry_run():
    d = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--frames", "5"])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["frames_per_step_all_ranks"] == 10                 # 5 frames per rank, every frame owned once
    assert d["counters"] == "N=1 only" and d["cpu_baseline"] == "N=1 only"   # the N > 1 line says where those fields are measured
    assert d["weight_broadcast_bytes"] == (9 * (256 + 64) * 128 + 128 + 2 * (128 * 128 + 128) + 2 * 128 + 2) * 4   # G-Net at D=64


def test_bench_reports_per_rank_rates():
    d = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--frames", "4"])
    assert len(d["per_rank_frames_per_s"]) == 2 and all(v > 0 for v in d["per_rank_frames_per_s"])


def test_dead_rank_takes_the_launch_down_quickly():
    """One rank exits with 3 before the rendezvous: the launcher must stop the surviving rank (which is waiting for the
    group) and return non-zero within seconds — not after a collective timeout."""
    import time
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["MAGNET_BENCH_FAIL_RANK"] = "1"
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2"],
                         env=e, capture_output=True, text=True, timeout=60)
    dt = time.monotonic() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert dt < 10.0 + 8.0, f"launcher took {dt:.1f} s to give up"          # the bound is 10 s; python + torch start-up of the ranks comes on top
    assert "[bench launcher] rank 1 exited with 3" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]      # no result line from a failed launch


def test_bench_single_rank_creates_a_one_rank_group_and_broadcasts():
    """A plain `python bench.py` (no RANK in the environment) still creates a ONE-rank process group and runs the weight broadcast
    and its checksum all-gather: the N = 1 line drives the same collective path as N = 8 (round-4 verdict, item 5)."""
    d = _run(["--gpus", "1", "--dry-run", "--steps", "2"])
    gnet_bytes = (9 * (256 + 64) * 128 + 128 + 2 * (128 * 128 + 128) + 2 * 128 + 2) * 4
    assert d["n_gpus"] == 1 and d["weight_broadcast_bytes"] == gnet_bytes
    assert d["rccl"] == {"world": 1, "backend": "gloo", "broadcast_bytes": gnet_bytes, "broadcast_verified": True}


def test_bench_under_torchrun_env_does_not_respawn():
    """With RANK in the environment (how the driver launches N > 1) bench.py must join the group, not spawn again."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = _run(["--gpus", "1", "--dry-run", "--steps", "2"],
             env=dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert d["n_gpus"] == 1


def test_bench_world_size_8_dry_run_with_fnet_bucket():
    """The launch the driver's scaling run makes, at its largest size: 8 ranks (gloo, no kernels), one port, 8 x (stdout + stderr)
    pumps; every frame owned once; the G-Net bucket and — with --with-fnet — the F-Net bucket (the shared weights north_star names)
    are broadcast and their sizes reported; every rank is pinned to its own CPU slice (when there are at least 8 CPUs)."""
    d = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--frames", "3", "--with-fnet"])
    assert d["n_gpus"] == 8 and d["config"]["frames_per_step_all_ranks"] == 24
    assert len(d["per_rank_frames_per_s"]) == 8 and all(v > 0 for v in d["per_rank_frames_per_s"])
    assert d["weight_broadcast_bytes"] == (9 * (256 + 64) * 128 + 128 + 2 * (128 * 128 + 128) + 2 * 128 + 2) * 4
    assert d["fnet_weight_broadcast_bytes"] == 13409632                  # PSMNet: 3.34 M parameters + BatchNorm buffers
    ncpu = len(os.sched_getaffinity(0))
    assert len(d["cpus_per_rank"]) == 8
    if ncpu >= 8:
        assert sum(d["cpus_per_rank"]) == ncpu and max(d["cpus_per_rank"]) - min(d["cpus_per_rank"]) <= 1


def test_dead_rank_at_world_size_8():
    import time
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["MAGNET_BENCH_FAIL_RANK"] = "5"
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2"],
                         env=e, capture_output=True, text=True, timeout=120)
    dt = time.monotonic() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert dt < 10.0 + 25.0, f"launcher took {dt:.1f} s to give up"      # 8 python + torch start-ups on top of the 10 s bound
    assert "[bench launcher] rank 5 exited with 3" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_counter_children_inherit_the_parents_step_flags():
    """bench.py's rocprofv3 counter children must run the configuration of the line they annotate (advisor, round 5): every flag that
    changes what a step executes is forwarded; they never create a process group and never replay a graph (counters per dispatch)."""
    import argparse
    import bench
    a = argparse.Namespace(workload="C4", frames=12, iters=2, feat_dtype="fp32", path=2, conv_backend="torch", no_fuse_tail=True,
                           no_fuse_upsample=False, graph=False, overlap=True, overlap_pack=False, packed_inputs=True, with_fnet=False,
                           dev_lib=True, kernel_only=False, nchw_out=False)
    args = bench._step_child_args(a, steps=2, warmup=1)
    for flag, val in (("--workload", "C4"), ("--frames", "12"), ("--iters", "2"), ("--feat-dtype", "fp32"), ("--path", "2"),
                      ("--conv-backend", "torch"), ("--steps", "2"), ("--warmup", "1")):
        assert args[args.index(flag) + 1] == val, (flag, args)
    for flag in ("--no-fuse-tail", "--overlap", "--packed-inputs", "--dev-lib", "--no-group", "--no-graph"):
        assert flag in args, (flag, args)
    for flag in ("--no-fuse-upsample", "--overlap-pack", "--with-fnet", "--kernel-only", "--nchw-out", "--gpus"):
        assert flag not in args, (flag, args)
