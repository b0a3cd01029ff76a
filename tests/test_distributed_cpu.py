"""N > 1 path on CPU: two gloo ranks run bench.py's cross-rank aggregation (time = MAX,
items = SUM: one independent channel per rank, no data-path collective) and the per-rank
channel/seed assignment."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # rank r "measured" dt = 1 + r seconds and consumed 1000 * (r + 1) items
    dt, total = bench.aggregate(dist, 1.0 + rank, 1000.0 * (rank + 1), torch.device("cpu"))
    # the initial channel scatter (the workload's only collective)
    n = 1000
    def make_all():
        return torch.stack([torch.full((n,), complex(r + 1, -r), dtype=torch.complex64) for r in range(world)])
    mine = bench.scatter_channels(dist, make_all, n, torch.device("cpu"), rank, world)
    out[rank] = (dt, total, complex(mine[0].item()), bool((mine == mine[0]).all()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregation_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        dt, total, first, uniform = out[r]
        assert dt == 2.0          # MAX over ranks
        assert total == 3000.0    # SUM over ranks
        assert first == complex(r + 1, -r) and uniform   # every rank received its own channel


def test_single_rank_aggregation_is_identity():
    import bench
    assert bench.aggregate(None, 1.5, 42.0, torch.device("cpu")) == (1.5, 42.0)
