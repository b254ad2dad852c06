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
