"""CPU, world_size 2 over gloo: the N > 1 path (shard, solve, gather) gives the single-process
result.  The per-rank solver here is the oracle (the HIP library needs a GPU); on the GPU box the
same code runs with BatchSolver.solve and the nccl (RCCL) backend."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, oracle_for
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.dist import shard_range, solve_sharded
from mpc_trajectory_generator_amd.harness import synthetic_batch


def test_shard_range_covers_batch():
    for B in (0, 1, 7, 8192, 65536):
        for world in (1, 2, 3, 8):
            ends = [shard_range(B, r, world) for r in range(world)]
            assert ends[0][0] == 0 and ends[-1][1] == B
            assert all(ends[i][1] == ends[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in ends) - min(b - a for a, b in ends) <= 1


def _worker(rank, world, port, B, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = named_config("cfg1")
    o = oracle_for(cfg)
    P = synthetic_batch(cfg, 1, B, 3)
    fn = lambda p, u, y, c: o.solve_batch(p, u0=u, y0=y, c0=c, threads=2)      # noqa: E731
    U, Y, st = solve_sharded(fn, P)
    if rank == 0:
        q.put((U, Y, st["num_inner_iterations"].copy(), st["exit_status"].copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [9])
def test_two_ranks_equal_one(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    [p.start() for p in procs]
    try:
        U, Y, it, ex = q.get(timeout=120)
    finally:
        [p.join(timeout=30) for p in procs]
        [p.kill() for p in procs if p.is_alive()]
    assert all(p.exitcode == 0 for p in procs)
    cfg = named_config("cfg1")
    Uo, Yo, sto = oracle_for(cfg).solve_batch(synthetic_batch(cfg, 1, B, 3), threads=4)
    assert np.array_equal(U, Uo) and np.array_equal(Y, Yo)
    assert np.array_equal(it, sto["num_inner_iterations"]) and np.array_equal(ex, sto["exit_status"])


def _fake_solve(p, u, y, c):
    """Stand-in for the per-rank solver at sizes the oracle cannot do on a CPU: results are a fixed function of the
    parameter rows, so that every instance's slot in the gathered output can be checked."""
    from mpc_trajectory_generator_amd import _lib
    n = p.shape[0]
    U = p[:, :40] * 2.0 + 1.0
    Y = p[:, 40:80] - 3.0
    st = np.zeros(n, dtype=_lib.STATUS_DTYPE)
    st["num_inner_iterations"] = (np.abs(p[:, 0]) * 1000).astype(np.uint32)
    st["exit_status"] = (p[:, 1] > 0).astype(np.int32)
    st["cost"] = p[:, 2]
    return U, Y, st


def _worker_big(rank, world, port, B, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = np.random.default_rng(11).standard_normal((B, 90))
    U, Y, st = solve_sharded(_fake_solve, P)
    Uo, Yo, sto = _fake_solve(P, None, None, None)
    ok = bool(np.array_equal(U, Uo) and np.array_equal(Y, Yo) and st.tobytes() == sto.tobytes())
    lo, hi = shard_range(B, rank, world)
    q.put((rank, ok, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_ragged_batch_of_config3_size():
    """BASELINE config 3's sharding (65 536 instances over 8 ranks) plus three extra instances so that the shards are
    ragged: every rank ends up with the whole batch in batch order, bit for bit, through the one pack / all_gather /
    unpack path bench.py uses."""
    B, world = 65536 + 3, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_big, args=(r, world, port, B, q)) for r in range(world)]
    [p.start() for p in procs]
    try:
        got = sorted(q.get(timeout=240) for _ in range(world))
    finally:
        [p.join(timeout=60) for p in procs]
        [p.kill() for p in procs if p.is_alive()]
    assert all(p.exitcode == 0 for p in procs)
    assert [g[0] for g in got] == list(range(world)) and all(g[1] for g in got)
    sizes = [g[3] - g[2] for g in got]
    assert sum(sizes) == B and max(sizes) - min(sizes) == 1 and got[0][2] == 0 and got[-1][3] == B
