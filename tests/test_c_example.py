"""The C ABI from plain C (examples/c_api_demo.c): compiles with gcc against include/nmpc_solver.h,
links libnmpc_hip.so, fails loudly without a HIP device and solves with one."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpc_trajectory_generator_amd", "csrc")


def _build(tmp_path):
    from mpc_trajectory_generator_amd import _lib
    _lib.build_library()
    exe = str(tmp_path / "c_api_demo")
    r = subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_api_demo.c"), "-o", exe, "-L", CSRC, "-lnmpc_hip",
                        f"-Wl,-rpath,{CSRC}", "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_builds_and_needs_a_device(tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is checked on the CPU box")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "needs a HIP device" in r.stderr


@pytest.mark.gpu
def test_c_example_solves(tmp_path):
    import numpy as np
    from mpc_trajectory_generator_amd import named_config
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "instance 0: Converged" in r.stdout and "warm restart: instance 0 Converged" in r.stdout
    # a batch from file: the same numbers the Python binding gives
    cfg = named_config("cfg1")
    P = synthetic_batch(cfg, 11, 5, 77)
    path = tmp_path / "p.bin"
    P.tofile(path)
    r = subprocess.run([exe, str(path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from mpc_trajectory_generator_amd.solver import BatchSolver
    s = BatchSolver(cfg, max_batch=8)
    u, y, st = s.solve(P)
    s.close()
    for b in range(5):
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(f"instance {b}:")][0]
        assert f"{st['num_inner_iterations'][b]} inner iterations" in line
        assert f"u[0:2] = ({u[b, 0]:.6f}, {u[b, 1]:.6f})" in line
