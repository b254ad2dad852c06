"""GPU: the N > 1 code path of bench.py and dist.solve_sharded with the nccl (= RCCL) backend.

A gpurun box has one GPU, so the collective runs at world size 1 -- which still initialises RCCL, packs the
u | y | status payload on device, runs all_gather_into_tensor and checks the gathered slots -- and the launch
logic (`--gpus N` spawning N ranks, refusing more ranks than devices, refusing a WORLD_SIZE mismatch) is
checked for what it must do on this box.  World size 2 over gloo on CPU: tests/test_dist_gloo.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NMPC_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_refuses_more_ranks_than_gpus():
    """(runs on CPU too) `--gpus N` never reports a figure from fewer than N devices."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r = _run(["--gpus", str(max(have, 1) + 1), "--steps", "1", "--warmup", "0", "--no-extras"])
    assert r.returncode != 0
    assert f"only {have} GPU(s) visible" in (r.stderr + r.stdout)


def test_bench_refuses_world_size_mismatch():
    """(runs on CPU too) launched by torchrun with another world size than --gpus says: loud failure, no JSON."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
             {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_multi_gpu_path_over_rccl_world1():
    """bench.py's own N > 1 path: RCCL initialised, results packed on device, one all_gather, slots verified."""
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1024", "--no-extras"],
             {"NMPC_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(29600 + os.getpid() % 300),
              "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["gather_checked"] is True
    assert "RCCL all_gather" in out["config"]["gather"]
    assert out["value"] > 0 and out["config"]["batch_per_gpu"] == 1024
    # one line per rank (a wrong rank -> device mapping is visible) and the gather timed alone
    assert [r_["rank"] for r_ in out["ranks"]] == [0] and out["ranks"][0]["shard"] == [0, 1024] and out["ranks"][0]["device"] == 0
    assert out["ranks"][0]["kernel_ms"] > 0 and 0 < out["gather_alone"]["ms"] < out["ms_per_step"]
    assert out["gather_alone"]["bytes_per_rank"] == 1024 * (40 + 40 + 9) * 8
    # the same batch without the collective: same solver work (the gather changes nothing but the time)
    r2 = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "1024", "--no-extras"])
    assert r2.returncode == 0, r2.stderr[-2000:]
    out2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    assert out2["gather_checked"] is None and out2["mean_inner_iters"] == out["mean_inner_iters"]


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` outside torchrun launches N ranks itself.  With N = visible GPUs >= 2 that is a
    real multi-GPU run; on a one-GPU box the spawn path is covered by the refusal test above."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU visible: nothing to spawn")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1024", "--no-extras"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["gather_checked"] is True


_SHARDED = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["NMPC_ROOT"])
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.dist import solve_sharded
from mpc_trajectory_generator_amd.harness import synthetic_batch
from mpc_trajectory_generator_amd.solver import BatchSolver
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = named_config("cfg1")
P = synthetic_batch(cfg, 11, 37, 5)
s = BatchSolver(cfg, max_batch=64)
U, Y, st = solve_sharded(lambda p, u, y, c: s.solve(p, u0=u, y0=y, c0=c), P, device=torch.device("cuda", 0))
u, y, st1 = s.solve(P)
assert np.array_equal(U, u) and np.array_equal(Y, y) and st.tobytes()[:0] == b""
for f in ("exit_status", "num_inner_iterations", "cost", "penalty"):
    assert np.array_equal(st[f], st1[f]), f
dist.destroy_process_group()
print("SHARDED_OK")
"""


@pytest.mark.gpu
def test_solve_sharded_over_rccl_world1():
    """dist.solve_sharded (the function the gloo test runs at world size 2) with device tensors over RCCL."""
    env = dict(os.environ, NMPC_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SHARDED], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end_on_one_gpu():
    """The whole N = 2 path of bench.py with two real processes -- spawn, rank-dependent batches, shard bookkeeping, the gather and its check,
    max-over-ranks timing, the per-rank report -- on a box with one GPU: both ranks on device 0, the collective over gloo (RCCL refuses two
    ranks on one device).  A functional check; the line says so (`shared_gpu`) and its value is not a measurement."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1024", "--no-extras"], {"NMPC_BENCH_SHARED_GPU": "1"})
    assert r.returncode == 0, (r.stderr + r.stdout)[-2500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["shared_gpu"] is True and out["gather_checked"] is True
    assert "functional check" in out["config"]["gather"] and out["config"]["batch_per_gpu"] == 1024
    assert [r_["rank"] for r_ in out["ranks"]] == [0, 1]
    assert out["ranks"][0]["shard"] == [0, 1024] and out["ranks"][1]["shard"] == [1024, 2048]
    assert all(r_["kernel_ms"] > 0 for r_ in out["ranks"]) and out["value"] > 0
