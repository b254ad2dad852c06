"""GPU: the receding-horizon loop on device (nmpc_loop_*) against its host
mirror.  The mirror is ``VectorizedRecedingHorizon`` -- itself pinned to the loop version and, through
tests/test_harness.py, to the goldens recorded from the reference's PathGenerator.run -- driven by the
oracle and given the kernels' sin / cos, so parameter vectors, states, reference indices and solver
counters must agree bit for bit, step after step."""
import numpy as np
import pytest

from conftest import oracle_for
from mpc_trajectory_generator_amd import harness, named_config
from mpc_trajectory_generator_amd.config import load_config

pytestmark = pytest.mark.gpu


def _fleet(cfg, route, B, seed, K):
    rng = np.random.default_rng(seed)
    n = len(route.x_ref)
    i0 = rng.integers(0, max(1, n - 25), B)
    xr, yr, tr = np.array(route.x_ref), np.array(route.y_ref), np.array(route.theta_ref)
    starts = np.stack([xr[i0] + rng.normal(0, 0.05, B), yr[i0] + rng.normal(0, 0.05, B), tr[i0] + rng.normal(0, 0.1, B)], axis=1)
    dyn = None
    if K:
        jj = np.minimum(n - 1, i0[:, None] + rng.integers(0, 30, (B, K)))
        c = np.stack([xr[jj], yr[jj]], axis=2)
        dyn = (c + rng.uniform(-5, 5, (B, K, 2)), c + rng.uniform(-5, 5, (B, K, 2)), rng.uniform(0.05, 0.1, (B, K)),
               rng.uniform(0.3, 1.0, (B, K)), rng.uniform(0.3, 1.0, (B, K)), rng.uniform(0, np.pi, (B, K)))
    return i0, starts, dyn


@pytest.mark.parametrize("name,scene,K,steps,sinus", [("cfg4", 11, 3, 12, False), ("cfg4", 1, 2, 40, False), ("cfg1", 11, 0, 8, False),
                                                       ("nobs3", 11, 1, 6, False), ("cfg4", 11, 3, 10, True),
                                                       ("cfg2", 11, 2, 5, False)])
def test_device_loop_equals_host_mirror(name, scene, K, steps, sinus):
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.trajectory import DeviceRecedingHorizon, VectorizedRecedingHorizon
    # "nobs3": fewer circle slots than the scene has vertices -> the closest-vertex window is exercised; "cfg2": N_hor = 40, the
    # two-stages-per-lane kernel inside the loop
    cfg = load_config(Nobs=3) if name == "nobs3" else named_config(name)
    route = harness.scene_route(cfg, scene)
    B = 24
    i0, starts, dyn = _fleet(cfg, route, B, 7 + scene, K)
    o = oracle_for(cfg)
    s = BatchSolver(cfg, max_batch=32)
    try:
        dev = DeviceRecedingHorizon(s, route, starts, dyn, max_steps=steps, idx0=i0, sinus_object=sinus)
        host = VectorizedRecedingHorizon(route, starts, dyn, sincos=o.sincos_array, sinus_object=sinus)   # (third ellipse: sinusoidal law)
        host.idx = i0.astype(np.int64)
        for k in range(steps):
            dev.step()
            P, st = host.step(lambda P, U, Y: o.solve_batch(P, u0=U, y0=Y, threads=8))
            Pd, Ud, Yd = dev.params()
            state, last_u, idx, done, std = dev.read()
            assert np.array_equal(Pd, P), f"step {k}: parameter vectors differ at columns {np.unique(np.nonzero(Pd != P)[1])[:10]}"
            assert np.array_equal(Ud, host.U) and np.array_equal(Yd, host.Y)
            assert np.array_equal(state, host.state) and np.array_equal(last_u, host.last_u)
            assert np.array_equal(idx, host.idx) and np.array_equal(done, host.done)
            assert np.array_equal(std["num_inner_iterations"], st["num_inner_iterations"])
            assert np.array_equal(std["exit_status"], st["exit_status"])
        T = dev.trajectory()
        assert T.shape == (steps * cfg.num_steps_taken + 1, B, 3)
        assert np.array_equal(T, np.stack(host.traj))
        dev.close()
    finally:
        s.close()


def test_device_loop_reaches_goal_and_brakes():
    """Closed loop to the goal on scene 1 (reference config 0): the braking branch (velocity reference
    from the distance table inside the last sample) and the terminal test are reached on device."""
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.trajectory import DeviceRecedingHorizon, VectorizedRecedingHorizon
    cfg = named_config("cfg1")
    route = harness.scene_route(cfg, 1)
    o = oracle_for(cfg)
    n = len(route.x_ref)
    B = 6
    i0 = np.array([n - 40, n - 30, n - 22, n - 12, n - 5, n - 2])
    xr, yr, tr = np.array(route.x_ref), np.array(route.y_ref), np.array(route.theta_ref)
    starts = np.stack([xr[i0], yr[i0], tr[i0]], axis=1)
    s = BatchSolver(cfg, max_batch=8)
    try:
        dev = DeviceRecedingHorizon(s, route, starts, None, max_steps=120, idx0=i0)
        host = VectorizedRecedingHorizon(route, starts, None, sincos=o.sincos_array)
        host.idx = i0.astype(np.int64)
        for k in range(120):
            dev.step()
            P, st = host.step(lambda P, U, Y: o.solve_batch(P, u0=U, y0=Y, threads=6))
            if k % 10 == 9 or k < 3:
                Pd, _, _ = dev.params()
                assert np.array_equal(Pd, P), f"step {k}"
        state, last_u, idx, done, _ = dev.read()
        assert np.array_equal(state, host.state) and np.array_equal(done, host.done)
        assert done.all()
        dev.close()
    finally:
        s.close()
