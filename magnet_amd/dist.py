"""Multi-GPU plumbing for the matching path.

Reference frames are independent (homography.py:88 has no cross-batch state), so the path shards
embarrassingly: one process per GPU, each rank owns a contiguous range of reference frames and runs
the whole loop locally.  The only collective on the inference path is a ONE-TIME broadcast of the
shared trainable weights (G-Net, mask head — and F-Net/D-Net when present) from rank 0; over xGMI
that is one flat bucket per dtype (a few MB) rather than one message per parameter.  Timing
reductions (max over ranks) and optional metric gathers use tiny all_reduce / all_gather calls.
backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_from_env(backend: str | None = None, always: bool = False):
    """Initialise torch.distributed from the torchrun environment.  Under torchrun (RANK set) the group is created even for one
    rank, rendezvous through MASTER_ADDR / MASTER_PORT as the launcher set them (never overwritten here).
    `always=True` (bench.py on a GPU) also creates a ONE-rank group for a plain `python bench.py` — only when the environment
    describes a single process (no RANK, WORLD_SIZE absent or 1): the group gets a PRIVATE store (a TCPStore this process owns, bound
    by the OS to a free port: no probe-then-rebind race between bench processes starting together, nothing read from or written to
    MASTER_*), so that the N = 1 launch exercises the same RCCL path as N = 8 (train_MaGNet.py:197-210 is the reference's pattern).
    WORLD_SIZE > 1 without RANK is a broken launcher environment and raises instead of guessing a rendezvous.
    Without `always` a plain launch stays single-process."""
    rank, world, local = env_world()
    if dist.is_initialized():
        return rank, world, local
    if "RANK" not in os.environ:
        if world > 1:
            raise RuntimeError(f"WORLD_SIZE={world} but RANK is not set: launch the ranks with torch.distributed.run "
                               "(or `python bench.py --gpus N`, which spawns them), or unset WORLD_SIZE")
        if not always:
            return rank, world, local
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if "RANK" not in os.environ:                           # our own one-rank group: never a client of somebody else's store
        kw["store"] = dist.TCPStore("127.0.0.1", 0, 1, is_master=True, wait_for_workers=False)
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def cpu_slice(local_rank: int, local_world: int, cpus=None):
    """The CPUs rank `local_rank` of `local_world` ranks on this node should run on: a contiguous, balanced slice of the CPUs the
    process may use (contiguous CPU ids share a socket / NUMA node on the hosts this targets, and GPU i hangs off the socket that
    holds slice i of 8 on the 8-GPU MI355X nodes).  With fewer CPUs than ranks every rank keeps the full set."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    lo, hi = shard_range(len(cpus), local_rank, local_world)
    return cpus[lo:hi]


def pin_to_cpu_slice(local_rank: int, local_world: int) -> list:
    """Pin this process (host-side input generation, launch thread, torch's intra-op pool) to its CPU slice, so that 8 ranks neither
    migrate across sockets nor oversubscribe each other's cores.  MAGNET_BENCH_AFFINITY=0 leaves the affinity alone.  Returns the
    CPUs in effect."""
    if os.environ.get("MAGNET_BENCH_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    mine = cpu_slice(local_rank, local_world)
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), torch.get_num_threads())))
    except OSError:
        pass
    return sorted(os.sched_getaffinity(0))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) range of reference frames owned by `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@torch.no_grad()
def broadcast_module_(module: torch.nn.Module, src: int = 0):
    """Broadcast all parameters and buffers of `module` from rank `src`, one flat bucket per dtype.  Runs whenever a process group
    exists — also a one-rank group, so that the N = 1 launch drives the same collective (and reports the same byte count) as N = 8.
    Returns the bytes broadcast (0 without a process group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    by_dtype: dict = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    nbytes = 0
    for (_, _), tensors in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        nbytes += flat.numel() * flat.element_size()
    return nbytes


@torch.no_grad()
def module_checksum(module: torch.nn.Module) -> float:
    """Order-dependent float64 checksum of all parameters and buffers (sum of element x (1 + index mod 251))."""
    acc = 0.0
    for t in list(module.parameters()) + list(module.buffers()):
        if not t.is_floating_point():
            t = t.double()
        f = t.detach().reshape(-1).double()
        w = (torch.arange(f.numel(), device=f.device, dtype=torch.float64) % 251) + 1.0
        acc += float((f * w).sum().item())
    return acc


def broadcast_verified(module: torch.nn.Module, device=None) -> bool:
    """After broadcast_module_: every rank holds bit-identical weights <=> the checksums gathered from all ranks are equal."""
    vals = gather_floats(module_checksum(module), device=device)
    return all(v == vals[0] for v in vals)


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value: float, device=None) -> list:
    """[value of rank 0, ..., value of rank world-1] on every rank (one tiny all_gather; also in a one-rank group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
