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
    # configs[3]: one [channels, n] slab per rank from rank 0's host sample ring
    C = 3
    def make_bank():
        host, _ = bench.host_sample_ring(world, (C, n))
        for r in range(world):
            for c in range(C):
                host[r, c] = complex(10 * r + c, 0)
        return host
    slab = bench.scatter_channels(dist, make_bank, (C, n), torch.device("cpu"), rank, world)
    slab_ok = slab.shape == (C, n) and all(bool((slab[c] == complex(10 * rank + c, 0)).all()) for c in range(C))
    out[rank] = (dt, total, complex(mine[0].item()), bool((mine == mine[0]).all()) and slab_ok)
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


def test_bench_launcher_starts_one_process_per_gpu():
    """`python bench.py --gpus 2` without a torch.distributed launcher starts two fresh ranks itself
    (gloo rendezvous on 127.0.0.1 in --dry-run), rank 0 prints ONE line with n_gpus == --gpus"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gpus_requested"] == 2
    assert line["value"] == 3000.0 / 2.0   # SUM of items / MAX of times
    assert line["scatter_ok"] is True and line["ms_per_step_per_rank"] == [1.0, 2.0]
    job = line["job"]
    assert job["backend"] == "gloo" and job["world"] == 2 and [r["rank"] for r in job["ranks"]] == [0, 1]
    assert len({r["pci_bus_id"] for r in job["ranks"]}) == 2   # two different processes took part


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_bench_launcher_fails_when_a_rank_dies_before_the_scatter(fail_rank):
    """a rank that dies while the others are about to enter the scatter: the launcher stops the survivors (they
    would wait for ever) and returns non-zero, and no JSON line is printed"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GR4PM_BENCH_TEST_FAIL_RANK"] = str(fail_rank)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "stopping the other ranks" in r.stderr


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr


def _run_bench(argv, extra_env=None, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, env=env,
                          timeout=timeout)


@pytest.mark.parametrize("world", [4, 8])
def test_bench_launcher_at_world_size_4_and_8(world):
    """the launcher, the self-check pass in front of the job, the scatter shapes and the identities at the sizes the
    driver's scaling run uses (gloo here): N fresh ranks, one line, N distinct participants, SUM / MAX aggregation"""
    import json
    r = _run_bench(["--gpus", str(world), "--dry-run", "--steps", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # the self-check pass prints nothing
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scatter_ok"] is True
    assert line["value"] == 1000.0 * world * (world + 1) / 2 / world  # SUM of 1000 (r + 1) items / MAX of 1 + r seconds
    # round 5: every rank's own time in the line (which GPU was the slow one), and the scatter's own time and rate
    assert line["ms_per_step_per_rank"] == [1.0 + r for r in range(world)]
    sc = line["scatter"]
    assert sc["backend"] == "gloo" and sc["bytes_from_rank0"] == 8 * 64 * (world - 1) and sc["seconds"] > 0 and sc["gbs"] >= 0
    job = line["job"]
    assert job["world"] == world and [x["rank"] for x in job["ranks"]] == list(range(world))
    assert len({x["pci_bus_id"] for x in job["ranks"]}) == world


def test_bench_ranks_on_one_device_are_refused_by_every_rank():
    """the distinct-device check runs right after the rendezvous and on every rank: all of them leave with the
    message, nobody waits in a collective for a rank that quit"""
    import time
    t0 = time.time()
    r = _run_bench(["--gpus", "2", "--dry-run"], {"GR4PM_BENCH_TEST_SAME_DEVICE": "1"})
    assert r.returncode != 0 and time.time() - t0 < 120
    assert r.stderr.count("2 ranks on 1 distinct devices") >= 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_selfcheck_failure_shows_the_failing_ranks_stderr_and_stops():
    """first contact: the self-check pass fails -> its stderr is shown, the job itself is never started"""
    r = _run_bench(["--gpus", "2", "--dry-run"], {"GR4PM_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "---- stderr of rank 1 ----" in r.stderr and "forced failure (test)" in r.stderr
    assert "self-check failed" in r.stderr and "not starting the job" in r.stderr


def test_bench_rank_that_dies_inside_the_job_still_ends_it():
    """the self-check passes, a rank dies later (before the job's scatter): same outcome as before the self-check existed"""
    r = _run_bench(["--gpus", "2", "--dry-run"], {"GR4PM_BENCH_TEST_FAIL_RANK": "0", "GR4PM_BENCH_TEST_FAIL_IN_JOB": "1"})
    assert r.returncode != 0 and "self-check failed" not in r.stderr and "stopping the other ranks" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _budget_worker(rank, world, port, out, host_available):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GR4PM_BENCH_TEST_HOST_AVAILABLE=str(host_available))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    try:
        # configs[3]'s real slab: 64 channels x 2^22 samples x 8 bytes per rank, rank 0's pinned host ring holds `world` of them
        bench.scatter_budget(dist, rank, world, 64 * (1 << 22) * 8, torch.device("cpu"), host_ring=True)
        out[rank] = "passed"
    except SystemExit as e:
        out[rank] = str(e)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("host_gib,fits", [(64, True), (16, False)])
def test_scatter_budget_refuses_the_config3_ring_on_every_rank_before_anything_is_allocated(host_gib, fits):
    """configs[3] at N = 8: rank 0's pinned host sample ring is 8 x 64 x 2^22 x 8 B = 17.2 GB.  The budget check runs before
    any allocation and its verdict reaches every rank (broadcast): with 64 GiB available all eight ranks go on, with 16 GiB
    all eight leave with the same message -- nobody hangs in a scatter that rank 0 cannot feed.  (The first real 8-GPU
    run should fail on nothing but hardware.)"""
    world = 8
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_budget_worker, args=(world, _free_port(), out, host_gib << 30), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        if fits:
            assert out[r] == "passed", out[r]
        else:
            assert "cannot hold the sample ring of 8 ranks" in out[r] and "host 16.0 GiB needed / 16.0 GiB available" in out[r], out[r]

