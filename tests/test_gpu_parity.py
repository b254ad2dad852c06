"""GPU: the HIP path, called through the C ABI, against the CPU oracle and the committed goldens.

The bar is bit-exact: kernels and oracle implement the same canonical f64 arithmetic (DESIGN.md
section 4), so controls, multipliers, iteration counts and status words must be identical.  Against the
reference-derived goldens (different summation order, libm sin/cos) the tolerance is RTOL."""
import numpy as np
import pytest

from conftest import STATUS_FIELDS, VARIANTS, oracle_for
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.harness import synthetic_batch

pytestmark = pytest.mark.gpu
RTOL = 1e-11


@pytest.fixture(scope="module")
def solvers():
    from mpc_trajectory_generator_amd.solver import BatchSolver
    made = {}

    def get(name, **opts):
        key = (name, tuple(sorted(opts.items())))
        if key not in made:
            made[key] = BatchSolver(named_config(name), max_batch=8192, **opts)
        return made[key]
    yield get
    for s in made.values():
        s.close()


def assert_same_solution(gpu, cpu):
    (u, y, st), (uo, yo, sto) = gpu, cpu
    for f in STATUS_FIELDS:
        assert np.array_equal(st[f], sto[f]), f
    assert np.array_equal(u, uo)
    assert np.array_equal(y, yo)


def test_native_library_loaded(solvers):
    s = solvers("cfg1")
    s.ping()
    maps = open("/proc/self/maps").read()
    assert "libnmpc_hip.so" in maps


def test_primitives_bit_exact(solvers):
    s, o = solvers("cfg1"), oracle_for(named_config("cfg1"))
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-20, 20, 4000), rng.uniform(-1e3, 1e3, 500), [0.0, np.pi / 2, -np.pi, 1e-300]])
    sn, cs = s.test_sincos(x)
    ref = np.array([o.sincos(v) for v in x])
    assert np.array_equal(sn, ref[:, 0]) and np.array_equal(cs, ref[:, 1])
    a = np.abs(rng.normal(0, 1, 50000)) * 10.0 ** rng.integers(-30, 30, 50000)
    b = rng.normal(0, 1, 50000) * 10.0 ** rng.integers(-30, 30, 50000)
    q, r = s.test_divsqrt(a, b)
    assert np.array_equal(q, a / b) and np.array_equal(r, np.sqrt(a))       # IEEE division and sqrt


@pytest.mark.parametrize("name", list(VARIANTS))
def test_cost_layer_vs_oracle_and_golden(golden, solvers, name):
    d, cfg = golden[name], named_config(VARIANTS[name])
    s, o = solvers(VARIANTS[name]), oracle_for(cfg)
    n = len(d["u"])
    f, g, F1, F2 = s.evaluate(d["p"], d["u"])
    assert np.max(np.abs(f - d["f"]) / np.abs(d["f"])) <= RTOL
    assert np.max(np.abs(g - d["grad_f"]) / np.max(np.abs(d["grad_f"]), axis=1, keepdims=True)) <= RTOL
    assert np.max(np.abs(F1 - d["F1"])) <= RTOL * max(1.0, np.max(np.abs(d["F1"])))
    assert np.max(np.abs(F2 - d["F2"])) <= RTOL * max(1.0, np.max(np.abs(d["F2"])))
    for j, (c, y) in enumerate(zip(d["xi_c"], d["xi_y"])):
        psi, gp, F1, F2 = s.evaluate(d["p"], d["u"], np.full(n, c), np.tile(y, (n, 1)))
        assert np.max(np.abs(psi - d["psi"][:, j]) / np.abs(d["psi"][:, j])) <= RTOL
        assert np.max(np.abs(gp - d["grad_psi"][:, j]) /
                      np.max(np.abs(d["grad_psi"][:, j]), axis=1, keepdims=True)) <= RTOL
        for i in range(n):
            po, go, F1o, F2o = o.eval(d["p"][i], d["u"][i], c, y)
            assert psi[i] == po and np.array_equal(gp[i], go)
            assert np.array_equal(F1[i], F1o) and np.array_equal(F2[i], F2o)


@pytest.mark.parametrize("name,scene,B", [("cfg1", 11, 96), ("cfg1", 1, 33), ("cfg2", 11, 24)])
def test_solve_cold_start_bit_exact(solvers, name, scene, B):
    cfg = named_config(name)
    P = synthetic_batch(cfg, scene, B, 12345)
    assert_same_solution(solvers(name).solve(P), oracle_for(cfg).solve_batch(P, threads=8))


def test_pass_model_of_the_oracle_equals_the_kernels_pass_counter(solvers):
    """The oracle also counts how many evaluation passes a three-points-per-pass schedule needs for its (sequential)
    run -- u and u + h together; u_bar with the first two trials; further trials three at a time; a pass per Lipschitz
    back-off -- and the hybrid kernel counts the passes it actually executed.  Equal on every instance: the kernel
    wastes no pass, and the 6-point figure the same model gives (helper waves) can be trusted."""
    cfg = named_config("cfg1")
    P = synthetic_batch(cfg, 11, 96, 4711)
    _, _, st = solvers("cfg1").solve(P)
    _, _, sto = oracle_for(cfg).solve_batch(P, threads=8)
    assert np.array_equal(st["reserved"], sto["reserved"])
    assert st["reserved"].min() >= 3 and np.all(st["reserved"] >= st["num_outer_iterations"] * 2)


def test_solve_nobs50_and_dynamic_obstacles(solvers):
    cfg = named_config("cfg3")
    P = synthetic_batch(cfg, 11, 32, 12345, synthetic_circles=True)
    assert_same_solution(solvers("cfg3").solve(P), oracle_for(cfg).solve_batch(P, threads=8))
    cfg = named_config("cfg4")
    P = synthetic_batch(cfg, 11, 32, 12345, random_dyn=True)
    assert_same_solution(solvers("cfg4").solve(P), oracle_for(cfg).solve_batch(P, threads=8))


SHAPES = [(20, 10, 2), (20, 0, 0), (19, 10, 3), (17, 4, 1), (16, 0, 0), (15, 3, 2), (5, 1, 3), (2, 0, 0),
          (21, 10, 3), (32, 64, 3), (33, 10, 3), (34, 0, 0), (37, 3, 1), (39, 12, 2), (40, 4, 3), (40, 64, 0), (36, 0, 3)]


@pytest.mark.parametrize("N,nobs,ndyn", SHAPES)
def test_shape_sweep_bit_exact(N, nobs, ndyn):
    """Every accepted horizon class (one and two stages per lane, the shape-specialised and the run-time-shape kernels) with full and
    partial horizons, padded and empty obstacle tables: cost layer and solve against the oracle, bit for bit."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = load_config(N_hor=N, Nobs=nobs, Ndynobs=ndyn)
    P = synthetic_batch(cfg, 11, 10, 100 * N + nobs, random_dyn=ndyn > 0)
    s, o = BatchSolver(cfg, max_batch=16), oracle_for(cfg)
    try:
        gpu = s.solve(P)
        assert_same_solution(gpu, o.solve_batch(P, threads=8))
        rng = np.random.default_rng(N)
        U = gpu[0] + rng.normal(0, 0.05, gpu[0].shape)
        c = rng.choice([0.0, 1.0, 125.0], size=len(P))
        Y = rng.normal(0, 1.0, (len(P), cfg.n1))
        psi, g, F1, F2 = s.evaluate(P, U, c, Y)
        for i in range(len(P)):
            po, go, F1o, F2o = o.eval(P[i], U[i], c[i], Y[i])
            assert psi[i] == po and np.array_equal(g[i], go)
            assert np.array_equal(F1[i], F1o) and np.array_equal(F2[i], F2o)
    finally:
        s.close()


@pytest.mark.parametrize("name,env,kernel", [("cfg1", {}, "nmpc_solve_hyb_kernel<ShapeDefault>"),
                                             ("cfg1", {"NMPC_SHAPE": "any"}, "nmpc_solve_hyb_kernel<ShapeAny>"),
                                             ("cfg3", {}, "nmpc_solve_hyb_kernel<ShapeNobs50>"),
                                             ("cfg3", {"NMPC_SHAPE": "any"}, "nmpc_solve_hyb_kernel<ShapeAny>"),
                                             ("cfg2", {}, "nmpc_solve_hyb2_kernel<ShapeN40>"),
                                             ("cfg2", {"NMPC_SHAPE": "any"}, "nmpc_solve_hyb2_kernel<ShapeAny>"),
                                             ("cfg2", {"NMPC_TEAM_HELP": "0"}, "nmpc_solve_hyb2_kernel<ShapeN40>"),
                                             ("cfg2", {"NMPC_TEAM_OWNERS": "4"}, "nmpc_solve_hyb2_kernel<ShapeN40>")])
def test_alternative_kernels_same_bits(monkeypatch, name, env, kernel):
    """Shapes with a specialised kernel: the run-time-shape kernel must give the same bits on them; the handle reports which
    kernel runs.  (The two-point kernel that used to be a third alternative is retired: nmpc_solve_common.h.)"""
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = named_config(name)
    P = synthetic_batch(cfg, 11, 40, 4242, synthetic_circles=(name == "cfg3"))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = BatchSolver(cfg, max_batch=64, experiments=bool(env))
    try:
        assert s.kernel_name == kernel
        assert_same_solution(s.solve(P), oracle_for(cfg).solve_batch(P, threads=8))
    finally:
        s.close()


# restatement switches (include/nmpc_solver.h): every value of every switch, alone and
# combined, in every solve kernel -- hybrid (N <= 20), two-stage hybrid (20 < N <= 40; N = 24 and 33 the run-time shape, 40 the specialised one)
SWITCHES = [dict(akkt_gradient=0), dict(akkt_gradient=2), dict(ls_failure=1), dict(inner_status=1),
            dict(akkt_gradient=0, ls_failure=1, inner_status=1), dict(max_total_inner=150),
            dict(max_total_inner=700, ls_failure=1, akkt_gradient=0), dict(lbfgs_memory=7), dict(lbfgs_memory=2)]


@pytest.mark.parametrize("opts", SWITCHES, ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
@pytest.mark.parametrize("N", [20, 24, 40, 33])
def test_restatement_switches_bit_exact(N, opts):
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = load_config(N_hor=N)
    P = synthetic_batch(cfg, 11, 48 if N == 20 else 16, 777 + N)
    s = BatchSolver(cfg, max_batch=64, **opts)
    try:
        assert s.variant["akkt_gradient"] == ("per_trial", "step_top", "off")[opts.get("akkt_gradient", 1)]
        gpu = s.solve(P)
        cpu = oracle_for(cfg, **s.oracle_opts()).solve_batch(P, threads=8)
        assert_same_solution(gpu, cpu)
        if "max_total_inner" in opts:       # the deterministic max_duration: NotConvergedOutOfTime, never more than the budget
            assert gpu[2]["num_inner_iterations"].max() <= opts["max_total_inner"]
            assert (gpu[2]["exit_status"] == 2).any()
            assert np.all(np.isfinite(gpu[0]))
        if opts.get("inner_status") == 1:   # outer criteria met -> Converged: f2 and dy/c within delta on all of those
            ok = gpu[2]["exit_status"] == 0
            assert np.all(gpu[2]["f2_norm"][ok] <= 1e-4 + 1e-12)
    finally:
        s.close()


def test_line_search_exhaustion_paths_are_exercised():
    """ls_failure only matters when all 11 trials fail; make sure the batch used above really gets there
    (the oracle counts differ between the two settings), so the bit-exact test covers the tau = 0 branch."""
    cfg = named_config("cfg1")
    P = synthetic_batch(cfg, 11, 48, 797)
    a = oracle_for(cfg).solve_batch(P, threads=8)[2]
    b = oracle_for(cfg, ls_failure=1).solve_batch(P, threads=8)[2]
    assert not np.array_equal(a["num_grad_evals"], b["num_grad_evals"])


@pytest.mark.parametrize("m", [1, 5])
@pytest.mark.parametrize("N", [20, 24, 40])
def test_short_lbfgs_memory_bit_exact(N, m):
    """lbfgs_memory < 10: the pair that reaches age m leaves the ring (its slot, its products and its rho go back to zero), and no LDS
    write lands past the ring (the zero column of the hybrid kernel is per allocated slot)."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = load_config(N_hor=N)
    P = synthetic_batch(cfg, 11, 24, 31 * m + N)
    s = BatchSolver(cfg, max_batch=32, lbfgs_memory=m)
    try:
        assert_same_solution(s.solve(P), oracle_for(cfg, lbfgs_memory=m).solve_batch(P, threads=8))
    finally:
        s.close()


def test_migration_between_wave_slots_is_invisible(monkeypatch):
    """More instances than resident waves: long-running instances on the unfavoured wave slot of a SIMD are
    parked at outer-iteration boundaries and resumed by favoured waves (nmpc_solve_hyb.h).  With an aggressive
    threshold nearly every multi-outer-iteration instance migrates, some several times; results and counters
    must still be those of the sequential oracle."""
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = named_config("cfg1")
    P = synthetic_batch(cfg, 11, 2600, 99)
    monkeypatch.setenv("NMPC_PARK_MIN", "20")
    monkeypatch.setenv("NMPC_PARK_DEPTH", "64")
    s = BatchSolver(cfg, max_batch=2600, experiments=True)
    try:
        gpu = s.solve(P)
    finally:
        s.close()
    assert_same_solution(gpu, oracle_for(cfg).solve_batch(P, threads=8))
    monkeypatch.setenv("NMPC_PARK_MIN", "0")                      # migration off: the same bits
    s = BatchSolver(cfg, max_batch=2600, experiments=True)
    try:
        off = s.solve(P)
    finally:
        s.close()
    assert np.array_equal(off[0], gpu[0]) and np.array_equal(off[2]["reserved"], gpu[2]["reserved"])
    monkeypatch.setenv("NMPC_ORDER", "0")                         # ... and in index order instead of the launch order, scheduler off as well
    monkeypatch.setenv("NMPC_SCHED", "0")
    s = BatchSolver(cfg, max_batch=2600, experiments=True)
    try:
        plain = s.solve(P)
    finally:
        s.close()
    assert np.array_equal(plain[0], gpu[0]) and np.array_equal(plain[1], gpu[1]) and np.array_equal(plain[2]["reserved"], gpu[2]["reserved"])


def test_solve_warm_start_multipliers_penalty(solvers):
    cfg = named_config("cfg1")
    s, o = solvers("cfg1"), oracle_for(cfg)
    P = synthetic_batch(cfg, 11, 40, 7)
    u, y, st = s.solve(P)
    rng = np.random.default_rng(5)
    c0 = rng.choice([1.0, 5.0, 25.0], size=len(P))
    gpu = s.solve(P, u0=u, y0=y, c0=c0)
    cpu = o.solve_batch(P, u0=u, y0=y, c0=c0, threads=8)
    assert_same_solution(gpu, cpu)
    assert gpu[2]["num_inner_iterations"].sum() < st["num_inner_iterations"].sum()


def test_edge_cases(solvers):
    cfg = named_config("cfg1")
    s, o = solvers("cfg1"), oracle_for(cfg)
    # ragged batch sizes around the two-instances-per-wave packing, and B = 1
    for B in (1, 2, 3, 65):
        P = synthetic_batch(cfg, 1, B, 100 + B)
        assert_same_solution(s.solve(P), o.solve_batch(P, threads=4))
    # empty batch
    u, y, st = s.solve(np.zeros((0, cfg.n_p)))
    assert u.shape == (0, cfg.n_u) and st.shape == (0,)
    # iteration caps: tiny budgets must stop with NotConvergedIterations, identically
    s2, o2 = solvers("cfg1", max_inner=7, max_outer=3), oracle_for(cfg, max_inner=7, max_outer=3)
    P = synthetic_batch(cfg, 11, 16, 9)
    gpu, cpu = s2.solve(P), o2.solve_batch(P, threads=4)
    assert_same_solution(gpu, cpu)
    assert np.all(gpu[2]["exit_status"] == 1) and np.all(gpu[2]["num_outer_iterations"] <= 3)
    assert np.all(gpu[2]["num_inner_iterations"] <= 3 * 7)
    # wrong parameter count is an error, not a crash (OpEn error code 3003)
    from mpc_trajectory_generator_amd.solver import SolverError
    with pytest.raises(SolverError):
        s.solve(np.zeros((2, cfg.n_p - 1)))


def test_tcp_shim_sequential_semantics():
    """OptimizerTcpManager-shaped handle: warm start carried by the manager, OpEn error codes."""
    from mpc_trajectory_generator_amd.tcp_shim import OptimizerTcpManager
    cfg = named_config("cfg1")
    o = oracle_for(cfg)
    P = synthetic_batch(cfg, 1, 3, 11)
    mng = OptimizerTcpManager("mpc_build/navigation", config=cfg, max_batch=8)
    with pytest.raises(ConnectionRefusedError):
        mng.call(list(P[0]))
    mng.start()
    assert mng.ping() == {"Pong": 1}
    u_prev, y_prev = np.zeros((1, cfg.n_u)), np.zeros((1, cfg.n1))
    for k in range(3):
        r = mng.call(list(P[k]))
        assert r.is_ok()
        s = r.get()
        uo, yo, sto = o.solve_batch(P[k:k + 1], u0=u_prev, y0=y_prev)          # previous solution / multipliers persist
        assert s.solution == list(uo[0]) and s.lagrange_multipliers == list(yo[0])
        assert s.exit_status in ("Converged", "NotConvergedIterations")
        assert s.num_inner_iterations == sto["num_inner_iterations"][0] and s.solve_time_ms > 0
        u_prev, y_prev = uo, yo
    bad = mng.call(list(P[0][:-1]))
    assert not bad.is_ok() and bad.get().code == 3003
    assert mng.call(list(P[0]), initial_guess=[0.0] * 3).get().code == 1600
    assert mng.call(list(P[0]), initial_y=[0.0] * 3).get().code == 1700
    r = mng.call(list(P[0]), initial_guess=[0.0] * cfg.n_u, initial_y=[0.0] * cfg.n1, initial_penalty=5.0)
    uo, yo, sto = o.solve_batch(P[0:1], c0=np.array([5.0]))
    assert r.get().solution == list(uo[0])
    mng.kill()
    with pytest.raises(ConnectionRefusedError):
        mng.ping()


def test_closed_loop_scene1_reaches_goal():
    """BASELINE config 0 through the HIP path: default.yaml, scene 1, closed loop to the goal."""
    from mpc_trajectory_generator_amd import harness
    from mpc_trajectory_generator_amd.trajectory import TrajectoryGenerator
    cfg = named_config("cfg1")
    route = harness.scene_route(cfg, 1)
    xx, xy, uv, uw, solver_times, overhead = TrajectoryGenerator(cfg).run(route)
    assert abs(xx[-1] - 19.0) <= 0.05 and abs(xy[-1] - 10.0) <= 0.05 and abs(uv[-1]) < 0.005
    assert len(solver_times) == len(uv) < 400
    # clearance from the three NMPC vertices of the scene (circle radius 0.5, soft constraint)
    for vx, vy in route.vertices:
        assert np.min(np.hypot(np.array(xx) - vx, np.array(xy) - vy)) > 0.45
    d = np.load("tests/golden/harness_scene1.npz")       # the reference's loop with the oracle as its solver
    assert np.array_equal(xx, d["xx"]) and np.array_equal(xy, d["xy"])


def test_non_finite_inputs_are_contained(solvers):
    """A NaN / Inf parameter vector must end with a non-converged status (OpEn:
    NotConvergedNotFiniteComputation -> solver error 2000), never hang, and never disturb the other
    instances of the batch."""
    cfg = named_config("cfg1")
    s, o = solvers("cfg1"), oracle_for(cfg)
    P = synthetic_batch(cfg, 11, 12, 21)
    clean = o.solve_batch(P, threads=4)
    Pb = P.copy()
    Pb[3, 0] = np.nan            # state x
    Pb[7, 45] = np.inf           # a circle coordinate
    u, y, st = s.solve(Pb)
    assert st["exit_status"][3] in (1, 4) and st["exit_status"][7] in (1, 4)
    keep = [i for i in range(12) if i not in (3, 7)]
    assert np.array_equal(u[keep], clean[0][keep]) and np.array_equal(y[keep], clean[1][keep])
    for f in STATUS_FIELDS:
        assert np.array_equal(st[f][keep], clean[2][f][keep]), f
    from mpc_trajectory_generator_amd.tcp_shim import OptimizerTcpManager
    mng = OptimizerTcpManager(config=cfg, max_batch=4)
    mng.start()
    r = mng.call(list(Pb[3]))
    if not r.is_ok():
        assert r.get().code == 2000
    mng.kill()


def test_closed_loop_scene11_with_own_frontend():
    """End to end: plan on scene 11's polygons, drive the robot to the goal through the HIP solver."""
    from mpc_trajectory_generator_amd import harness
    from mpc_trajectory_generator_amd.frontend import scene_planner
    from mpc_trajectory_generator_amd.trajectory import TrajectoryGenerator
    cfg = named_config("cfg1")
    s = harness.SCENES[11]
    route = scene_planner(cfg, 11).route(s["start"], s["end"])
    out = TrajectoryGenerator(cfg).run(route, max_steps=600)
    assert out is not None
    xx, xy, uv, uw = out[:4]
    assert abs(xx[-1] - s["end"][0]) <= 0.05 and abs(xy[-1] - s["end"][1]) <= 0.05 and abs(uv[-1]) < 0.005
    assert max(uv) <= cfg.lin_vel_max + 1e-12 and max(abs(w) for w in uw) <= cfg.ang_vel_max + 1e-12
    acc = np.diff(np.array([0.0] + list(uv))) / cfg.ts
    assert acc.max() <= cfg.lin_acc_max + 5e-3 and acc.min() >= cfg.lin_acc_min - 5e-3      # ALM constraint, delta 1e-4 on ||dy||/c
    for vx, vy in route.vertices:
        assert np.min(np.hypot(np.array(xx) - vx, np.array(xy) - vy)) > 0.4


def test_batched_receding_horizon_on_gpu_equals_oracle(solvers):
    """Config-4 style lock-step batch: HIP and oracle driven through the same vectorised assembly give
    identical parameter vectors and states step after step (warm starts carried)."""
    from mpc_trajectory_generator_amd import harness
    from mpc_trajectory_generator_amd.trajectory import VectorizedRecedingHorizon
    cfg = named_config("cfg4")
    s, o = solvers("cfg4"), oracle_for(cfg)
    route = harness.scene_route(cfg, 1)
    rng = np.random.default_rng(5)
    B, n = 12, len(route.x_ref)
    i0 = rng.integers(0, n - 30, B)
    starts = np.stack([np.array(route.x_ref)[i0] + rng.normal(0, 0.05, B), np.array(route.y_ref)[i0] + rng.normal(0, 0.05, B),
                       np.array(route.theta_ref)[i0]], axis=1)
    c = np.stack([np.array(route.x_ref)[i0 + 8], np.array(route.y_ref)[i0 + 8]], axis=1)[:, None, :].repeat(3, axis=1)
    dyn = (c + rng.uniform(-3, 3, (B, 3, 2)), c + rng.uniform(-3, 3, (B, 3, 2)), rng.uniform(0.05, 0.1, (B, 3)),
           rng.uniform(0.3, 1.0, (B, 3)), rng.uniform(0.3, 1.0, (B, 3)), rng.uniform(0, np.pi, (B, 3)))
    a, b = VectorizedRecedingHorizon(route, starts, dyn), VectorizedRecedingHorizon(route, starts, dyn)
    a.idx, b.idx = i0.astype(np.int64), i0.astype(np.int64)
    for k in range(5):
        Pa, sta = a.step(lambda P, U, Y: s.solve(P, u0=U, y0=Y))
        Pb, stb = b.step(lambda P, U, Y: o.solve_batch(P, u0=U, y0=Y, threads=8))
        assert np.array_equal(Pa, Pb) and np.array_equal(a.state, b.state)
        assert np.array_equal(sta["num_inner_iterations"], stb["num_inner_iterations"])


def test_fuzz_shapes_and_options():
    """scripts/fuzz_parity.py: random shapes (both solve kernels, shape-specialised and run-time-shape) x random solver options and restatement switches x
    random small batches, cold and warm-started with user penalties -- every case bit-identical to the oracle."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "32", "7"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatches: 0 of 32" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


def test_nonfinite_cost_with_finite_controls_matches_the_oracle(solvers):
    """A penalty that overflows psi: the projected half step u stays finite (it is clamped to U) while the cost and
    the residual are not.  Kernels and oracle both end such a solve with NotConvergedNotFiniteComputation (OpEn itself
    checks u only -- the deviation is listed in include/nmpc_solver.h); the instances are isolated here so that the
    branch is covered by a bit-exact parity case of its own."""
    cfg = named_config("cfg1")
    P = synthetic_batch(cfg, 11, 16, 4242)
    for c0 in (1e306, 1e308):
        gpu = solvers("cfg1").solve(P, c0=np.full(16, c0))
        cpu = oracle_for(cfg).solve_batch(P, c0=np.full(16, c0), threads=8)
        assert (gpu[2]["exit_status"] == 4).any() and np.all(np.isfinite(gpu[0]))
        assert np.array_equal(gpu[2]["exit_status"], cpu[2]["exit_status"])
        assert np.array_equal(gpu[0], cpu[0]) and np.array_equal(gpu[2]["num_inner_iterations"], cpu[2]["num_inner_iterations"])
        bad = gpu[2]["exit_status"] == 4
        assert not np.all(np.isfinite(gpu[2]["cost"][bad]) & np.isfinite(gpu[2]["last_problem_norm_fpr"][bad]))


@pytest.mark.parametrize("name,B", [("cfg1", 4096), ("cfg1", 1), ("cfg2", 512), ("n27", 512)])
def test_per_instance_solve_time(solvers, name, B, monkeypatch):
    """status.solve_time_ms is THIS instance's first-start -> finish time on the device clock (the reference reads
    it per solve, src/mpc/mpc_generator.py:214, and derives its loop overhead from it, src/path_generator.py:387,402-403):
    positive, never above the kernel time of the batch, and growing with the work the instance needed."""
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.config import load_config
    cfg = load_config(N_hor=27) if name == "n27" else named_config(name)      # (20 < N_hor <= 32: the run-time-shape two-stage kernel)
    P = synthetic_batch(cfg, 11, B, 31337)
    s = BatchSolver(cfg, max_batch=B)
    try:
        s.solve(P)
        _, _, st = s.solve(P)
        batch_ms = s.last_batch_ms
    finally:
        s.close()
    t = st["solve_time_ms"]
    assert np.all(t > 0.0) and np.all(t <= batch_ms * 1.02 + 0.05), (t.min(), t.max(), batch_ms)
    assert t.max() >= 0.5 * batch_ms - 0.05                   # somebody ran (nearly) to the end of the launch
    if B > 1:
        work = st["reserved"].astype(np.float64)
        assert np.corrcoef(work, t)[0, 1] > 0.8
        assert t[np.argmax(work)] > np.median(t)


@pytest.mark.parametrize("env", [{"NMPC_TEAM_HELP": "0"}, {"NMPC_TEAM_OWNERS": "4"}, {"NMPC_TEAM_OWNERS": "2"},
                                 {"NMPC_TEAM_OWNERS": "1"}, {}],
                         ids=["no-help", "4-owners", "2-owners", "1-owner", "auto"])
@pytest.mark.parametrize("name,B", [("cfg1", 160), ("cfg3", 40), ("cfg4", 40), ("n17", 24), ("cfg2", 24), ("n35", 12)])
def test_team_modes_same_bits(monkeypatch, name, B, env):
    """The teams of the hybrid kernel (nmpc_solve_hyb.h): whether nobody helps, helpers appear only as the waves of a
    workgroup run out of work (4 owners), or every instance has helpers from its first iteration (1 owner, the
    small-batch mode), the owner consumes line-search trials in the sequential order -- the oracle's bits and counters,
    including the pass count of the three-point schedule."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = {"n17": lambda: load_config(N_hor=17, Nobs=4, Ndynobs=1), "n35": lambda: load_config(N_hor=35, Nobs=6, Ndynobs=2)}.get(
        name, lambda: named_config(name))()
    P = synthetic_batch(cfg, 11, B, 2718, synthetic_circles=(name == "cfg3"), random_dyn=(name in ("cfg4", "n17", "n35")))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = BatchSolver(cfg, max_batch=B, experiments=True)
    try:
        gpu = s.solve(P)
        gpu2 = s.solve(P, u0=gpu[0], y0=gpu[1])            # a warm start: short solves, helpers racing with fast owners
    finally:
        s.close()
    o = oracle_for(cfg)
    cpu = o.solve_batch(P, threads=8)
    assert_same_solution(gpu, cpu)
    assert np.array_equal(gpu[2]["reserved"], cpu[2]["reserved"])
    assert_same_solution(gpu2, o.solve_batch(P, u0=cpu[0], y0=cpu[1], threads=8))


@pytest.mark.parametrize("shape", [(5, 1, 3), (2, 0, 0), (10, 10, 0), (14, 0, 0)], ids=lambda s: "N%d-obs%d-dyn%d" % s)
def test_four_owners_on_short_horizons(monkeypatch, shape):
    """Short horizons make the LDS slice of a wave smaller than the twelve result areas a helper keeps in its own slice (N_hor <= 14):
    with four owners per workgroup a helper's areas for owner 3 then landed in the next wave's tables (round-3 advisor).  lds_layout
    now sizes the slice for the areas; enough instances here that waves finish at different times and help each other."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    N, nobs, ndyn = shape
    cfg = load_config(N_hor=N, Nobs=nobs, Ndynobs=ndyn)
    P = synthetic_batch(cfg, 11, 96, 4242, random_dyn=ndyn > 0)
    monkeypatch.setenv("NMPC_TEAM_OWNERS", "4")
    s = BatchSolver(cfg, max_batch=96, experiments=True)
    try:
        gpu = s.solve(P)
        gpu2 = s.solve(P, u0=gpu[0], y0=gpu[1])
    finally:
        s.close()
    o = oracle_for(cfg)
    cpu = o.solve_batch(P, threads=8)
    assert_same_solution(gpu, cpu)
    assert_same_solution(gpu2, o.solve_batch(P, u0=cpu[0], y0=cpu[1], threads=8))


def test_team_switches_and_budget_bit_exact(monkeypatch):
    """Line-search exhaustion (ls_failure = 1: the eleventh trial comes from a helper's result area), the per-trial
    AKKT gradient cache and the iteration budget, with helpers from the first iteration on."""
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = named_config("cfg1")
    P = synthetic_batch(cfg, 11, 48, 797)
    for opts in (dict(ls_failure=1), dict(akkt_gradient=0, ls_failure=1), dict(max_total_inner=300), dict(lbfgs_memory=3)):
        s = BatchSolver(cfg, max_batch=64, **opts)
        try:
            assert_same_solution(s.solve(P), oracle_for(cfg, **s.oracle_opts()).solve_batch(P, threads=8))
        finally:
            s.close()


@pytest.mark.parametrize("radius", ["", "0.5", "3.0"], ids=["default-radius", "radius-0.5m", "radius-3m"])
@pytest.mark.parametrize("name,B", [("cfg3", 96), ("n18", 40)])
def test_circle_culling_is_exact(monkeypatch, name, B, radius):
    """Shapes with many circle slots scan only the circles whose edge lies within a radius of the start position, and fall
    back to all of them for an evaluation in which a stage leaves that radius (eval_psi, CULL): with the production radius,
    with 3 m (both cases mixed) and with 0.5 m (the fall-back runs all the time, the reduced set is nearly empty) the
    oracle's bits -- on a batch in which half of the instances have a circle dropped right onto their reference path."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = load_config(N_hor=18, Nobs=37, Ndynobs=2) if name == "n18" else named_config(name)
    P = synthetic_batch(cfg, 11, B, 1234, synthetic_circles=True, random_dyn=(name == "n18"))
    N, off_c, off_r = cfg.N_hor, 20 + cfg.N_hor, cfg.n_p - 3 * cfg.N_hor
    for b in range(0, B, 2):                       # circle slot 7 <- on the reference sample in the middle of the horizon
        P[b, off_c + 21:off_c + 24] = (P[b, off_r + 3 * (N // 2)] + 0.2, P[b, off_r + 3 * (N // 2) + 1], 0.6)
    if radius:
        monkeypatch.setenv("NMPC_CULL_RADIUS", radius)
    s = BatchSolver(cfg, max_batch=B, experiments=True)
    try:
        gpu = s.solve(P)
    finally:
        s.close()
    cpu = oracle_for(cfg).solve_batch(P, threads=8)
    assert_same_solution(gpu, cpu)
    assert (cpu[2]["penalty"] > 1.0).any() and (cpu[2]["num_outer_iterations"] > 2).any()      # the circles did matter


@pytest.mark.parametrize("case", ["on-path", "edge", "rings", "ellipses", "warm-jump"])
@pytest.mark.parametrize("name", ["cfg1", "cfg3", "cfg4", "cfg2"])
def test_obstacle_certificate_is_exact(monkeypatch, name, case):
    """The shape-specialised three-point kernels skip the circle / ellipse activity scan of an evaluation while every stage has moved
    less than its clearance from the obstacles the last scan found untouched (eval_psi / eval_psi2, ObsCert) -- exact only if no
    obstacle can be entered unnoticed.  Obstacle fields built to defeat it -- circles dropped ONTO the reference (stages start
    inside), circles whose edge passes through reference samples (clearance ~ 0, sign of h at the rounding level), concentric
    rings of tiny and huge circles around the robot, fat ellipses crossing the horizon, warm starts from a far-away control
    sequence (every stage jumps by metres between the first evaluations) -- give the oracle's bits, and for N_hor <= 20 those of
    the run-time-shape kernel, which has no certificate."""
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = named_config(name)
    B, N = 48, cfg.N_hor
    P = synthetic_batch(cfg, 11, B, 4711, synthetic_circles=(name == "cfg3"), random_dyn=(name in ("cfg4", "cfg2") or case == "ellipses"))
    off_c, off_d, off_r = 20 + N, 20 + N + 3 * cfg.Nobs, cfg.n_p - 3 * N
    rng = np.random.default_rng(11)
    u0 = None
    for b in range(B):
        ref = P[b, off_r:].reshape(N, 3)
        if case == "on-path":
            for k in range(min(4, cfg.Nobs)):
                t = int(rng.integers(1, N))
                P[b, off_c + 3 * k:off_c + 3 * k + 3] = (ref[t, 0] + rng.normal(0, 0.1), ref[t, 1] + rng.normal(0, 0.1), rng.uniform(0.2, 0.8))
        elif case == "edge":
            for k in range(min(6, cfg.Nobs)):
                t = int(rng.integers(0, N)); r = rng.uniform(0.3, 1.5); a = rng.uniform(0, 2 * np.pi)
                P[b, off_c + 3 * k:off_c + 3 * k + 3] = (ref[t, 0] + r * np.cos(a), ref[t, 1] + r * np.sin(a), r)
        elif case == "rings":
            for k in range(cfg.Nobs):
                r = 10.0 ** rng.uniform(-3, 1.2)
                P[b, off_c + 3 * k:off_c + 3 * k + 3] = (P[b, 0] + rng.normal(0, 0.5), P[b, 1] + rng.normal(0, 0.5), r)
        elif case == "ellipses":
            for k in range(cfg.Ndynobs):
                for t in range(N):
                    o = off_d + (k * N + t) * 5
                    P[b, o:o + 5] = (ref[t, 0] + 0.3 * np.sin(0.4 * t + k), ref[t, 1] + 0.3 * np.cos(0.3 * t), rng.uniform(0.2, 2.5), rng.uniform(0.05, 0.6), rng.uniform(0, np.pi))
    if case == "warm-jump":
        u0 = np.tile(np.array([cfg.lin_vel_max, cfg.ang_vel_max]), (B, N)) * rng.choice([-0.3, 1.0], size=(B, 1))
    s = BatchSolver(cfg, max_batch=B)
    try:
        assert "Shape" in s.kernel_name and "ShapeAny" not in s.kernel_name
        gpu = s.solve(P, u0=u0)
    finally:
        s.close()
    cpu = oracle_for(cfg).solve_batch(P, u0=u0, threads=8)
    assert_same_solution(gpu, cpu)
    if case in ("on-path", "edge", "rings"):
        assert (cpu[2]["penalty"] > 1.0).any()                      # the obstacles did matter
    if N <= 20:
        monkeypatch.setenv("NMPC_SHAPE", "any")
        s = BatchSolver(cfg, max_batch=B, experiments=True)
        try:
            assert s.kernel_name.endswith("<ShapeAny>")
            assert_same_solution(s.solve(P, u0=u0), gpu)
        finally:
            s.close()


@pytest.mark.parametrize("shape", ["folded", "stationary", "far", "zigzag", "short"])
@pytest.mark.parametrize("name", ["cfg1", "cfg2", "n6"])
def test_windowed_cross_track_search_is_exact(name, shape):
    """The three-point kernels measure only the segments around each stage's previous nearest segment and accept that only
    if the rest of the reference is provably farther (eval_psi / eval_psi2, WIN); otherwise the full scan runs.  References
    built to defeat the test -- a path folded back onto itself (zero clearance), a robot parked at the goal (all segments
    the same point: every distance ties and the FIRST index must win), a robot 3 m off its path, a zig-zag whose far
    corners come close again, horizons with fewer segments than a window -- give the oracle's bits."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver
    cfg = load_config(N_hor=6 if shape != "short" else 4, Nobs=3, Ndynobs=1) if name == "n6" else named_config(name)
    if shape == "short" and name != "n6":
        pytest.skip("window narrower than the table: the small-horizon case covers it")
    B, N = 24, cfg.N_hor
    P = synthetic_batch(cfg, 11, B, 77)
    off_r = cfg.n_p - 3 * N
    rng = np.random.default_rng(5)
    for b in range(B):
        x0, y0, th = P[b, 0], P[b, 1], P[b, 2]
        step = 0.2 * (0.5 + rng.random())
        k = np.arange(N)
        if shape == "folded":            # out along the heading, back over the same points
            along = step * np.where(k < N // 2, k, N - 1 - k)
            rx, ry = x0 + along * np.cos(th), y0 + along * np.sin(th)
        elif shape == "stationary":      # the reference pads with its last point near the goal (src/path_generator.py:328-331)
            hold = rng.integers(0, N)
            along = step * np.minimum(k, hold)
            rx, ry = x0 + along * np.cos(th), y0 + along * np.sin(th)
        elif shape == "far":             # the robot starts 3 m beside its path
            rx = x0 + 3.0 * np.sin(th) + step * k * np.cos(th)
            ry = y0 - 3.0 * np.cos(th) + step * k * np.sin(th)
        elif shape == "zigzag":          # sharp corners every three samples, 0.3 m wide: far segments lie close together
            side = 0.3 * ((k // 3) % 2)
            rx = x0 + 0.05 * k * np.cos(th) - side * np.sin(th)
            ry = y0 + 0.05 * k * np.sin(th) + side * np.cos(th)
        else:
            rx, ry = x0 + step * k * np.cos(th), y0 + step * k * np.sin(th)
        P[b, off_r + 0:off_r + 3 * N:3] = rx
        P[b, off_r + 1:off_r + 3 * N:3] = ry
    s = BatchSolver(cfg, max_batch=B)
    try:
        gpu = s.solve(P)
        u, y, _ = gpu
        gpu2 = s.solve(P, u0=u, y0=y)                # warm restart: other centres, other trial points
    finally:
        s.close()
    o = oracle_for(cfg)
    cpu = o.solve_batch(P, threads=8)
    assert_same_solution(gpu, cpu)
    assert_same_solution(gpu2, o.solve_batch(P, u0=cpu[0], y0=cpu[1], threads=8))


def test_results_do_not_depend_on_timing():
    """Helpers, migration and the work queue decide WHERE and WHEN work runs, never what comes out: repeated solves of the same
    batches -- sizes that put every instance in a team from the start, that mix owners and helpers, and that exceed the resident
    waves -- give identical bits every time (scripts/stress_teams.py is the long version)."""
    import subprocess, sys, os
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_teams.py"), "6"], capture_output=True, text=True,
                       cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "STRESS_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_horizon_beyond_forty_is_refused():
    """include/nmpc_solver.h: NMPC_MAX_HORIZON = 40 -- one L-BFGS arithmetic and one set of certificates behind the ABI."""
    from mpc_trajectory_generator_amd.config import load_config
    from mpc_trajectory_generator_amd.solver import BatchSolver, SolverError
    for N in (41, 48, 64):
        with pytest.raises(SolverError) as e:
            BatchSolver(load_config(N_hor=N), max_batch=4)
        assert e.value.code == -1
