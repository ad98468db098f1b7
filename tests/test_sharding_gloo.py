"""world_size-2 gloo test (CPU) of bench.py's multi-rank logic: container sharding and the
MAX-time / SUM-bytes aggregation.  The data path itself has no collective (blocks are independent)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert bench.dist_env() == (rank, world, rank)
    plan = bench.shard_plan(rank, world, 8)
    # every rank times its own containers; rank 1 is slower
    secs, tot, comp = bench.reduce_results(1.0 + rank, len(plan) * 100, len(plan) * 40 + rank, dist, device="cpu")
    q.put((rank, plan, secs, tot, comp))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    plans = [g[1] for g in got]
    assert plans[0] == list(range(0, 8)) and plans[1] == list(range(8, 16))      # disjoint, contiguous, weak scaling
    for _, _, secs, tot, comp in got:
        assert secs == 2.0                      # MAX over ranks
        assert tot == 1600 and comp == 641      # SUM over ranks


def test_single_process_passthrough():
    import bench
    assert bench.reduce_results(0.5, 10, 4, None) == (0.5, 10, 4)
    assert bench.shard_plan(0, 1, 8) == list(range(8))
