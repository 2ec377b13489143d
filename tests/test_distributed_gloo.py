"""world_size-2 gloo tests of the N>1 path (runs on CPU): frame sharding covers every frame exactly
once, the one-bucket weight broadcast makes every rank identical to rank 0, max-over-ranks timing."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from magnet_amd import dist as mdist
from magnet_amd.magnet import GNET


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = mdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # different init on every rank
    net = GNET(ch_in=256 + 5)
    before = net.gnet[0].weight.clone()
    nbytes = mdist.broadcast_module_(net, src=0)
    flat = torch.cat([p.reshape(-1) for p in net.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    changed = not torch.equal(before, net.gnet[0].weight)
    lo, hi = mdist.shard_range(13, rank, world)
    t = mdist.max_over_ranks(1.0 + rank)
    s = mdist.sum_over_ranks(hi - lo)
    mdist.barrier()
    q.put((rank, same, changed, nbytes, lo, hi, t, s))
    dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "ranks differ after broadcast"
    assert res[1][2], "rank 1 weights were not overwritten by rank 0's"
    assert res[0][3] == 334082 * 4                       # one fp32 bucket with all of G-Net (D=5)
    assert (res[0][4], res[0][5], res[1][4], res[1][5]) == (0, 7, 7, 13)
    assert res[0][6] == res[1][6] == 2.0 and res[0][7] == 13.0


@pytest.mark.parametrize("n,world", [(13, 2), (8, 8), (5, 8), (100, 3)])
def test_shard_range_partitions(n, world):
    seen = []
    for r in range(world):
        lo, hi = mdist.shard_range(n, r, world)
        seen += list(range(lo, hi))
    assert seen == list(range(n))


def _private_group(q, master_port_before):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    os.environ["MASTER_PORT"] = master_port_before        # somebody else's rendezvous: must be neither used nor overwritten
    r, w, _ = mdist.init_from_env(backend="gloo", always=True)
    ok = dist.is_initialized() and dist.get_world_size() == 1 and (r, w) == (0, 1)
    vals = mdist.gather_floats(3.5)
    q.put((ok, vals, os.environ.get("MASTER_PORT")))
    dist.destroy_process_group()


def test_one_rank_group_uses_a_private_store_and_leaves_master_env_alone():
    """`always=True` without RANK: a one-rank group on a store this process owns (OS-assigned port) — several such processes
    starting together cannot collide, and MASTER_ADDR / MASTER_PORT are not touched."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_private_group, args=(q, "1")) for _ in range(3)]     # port 1: unusable if it were used
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and vals == [3.5] and port == "1" for ok, vals, port in res), res


def test_world_size_without_rank_is_an_error(monkeypatch):
    """A launcher that exports WORLD_SIZE > 1 but no RANK: every process would otherwise pick its own rendezvous and hang."""
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="RANK is not set"):
        mdist.init_from_env(backend="gloo", always=True)
    assert not dist.is_initialized()
