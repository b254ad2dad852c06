"""CPU: the oracle's PANOC/ALM restatement.  OpEn itself cannot run here (parity unpinned), so the
checks are optimality conditions and an independent solver."""
import numpy as np
import pytest
from scipy.optimize import minimize

from conftest import oracle_for
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.harness import synthetic_batch


@pytest.fixture(scope="module")
def solved():
    cfg = named_config("default")
    o = oracle_for(cfg)
    P = synthetic_batch(cfg, 11, 48, 12345)
    u, y, st = o.solve_batch(P, threads=8)
    return cfg, o, P, u, y, st


def test_solutions_feasible_and_status_consistent(solved):
    cfg, o, P, u, y, st = solved
    v, w = u[:, 0::2], u[:, 1::2]
    assert v.min() >= cfg.lin_vel_min and v.max() <= cfg.lin_vel_max          # PANOC returns the projected point
    assert np.abs(w).max() <= cfg.ang_vel_max
    assert set(np.unique(st["exit_status"])) <= {0, 1}
    conv = st["exit_status"] == 0
    assert conv.sum() >= len(P) // 3
    assert np.all(st["f2_norm"][conv] <= 1e-4 + 1e-12)                        # (epsilon, delta)-AKKT exit
    assert np.all(st["delta_y_norm_over_c"][conv] <= 1e-4 + 1e-12)
    assert np.all(st["num_outer_iterations"] >= 2)                            # criterion 1 needs nu > 0
    assert np.all(st["num_grad_evals"] >= st["num_inner_iterations"])


def test_inner_problem_against_scipy(solved):
    """At a converged point the final (c, y) sub-problem min_{u in U} psi(u) must not be improvable
    by an independent bound-constrained solver (L-BFGS-B on the oracle's psi / grad psi)."""
    cfg, o, P, u, y, st = solved
    idx = [i for i in np.where(st["exit_status"] == 0)[0] if st["num_outer_iterations"][i] == 2][:4]
    assert idx
    bounds = [(cfg.lin_vel_min, cfg.lin_vel_max), (-cfg.ang_vel_max, cfg.ang_vel_max)] * cfg.N_hor
    for i in idx:
        c = st["penalty"][i]
        # multipliers in force during the last inner solve: y_out is y+ of that solve; at an AKKT
        # exit ||y+ - y|| <= c*delta, so y+ serves as y to within the tolerance
        fun = lambda z: o.eval(P[i], z, c, y[i])[:2]                                   # noqa: E731
        psi_star = fun(u[i])[0]
        res = minimize(fun, u[i], jac=True, method="L-BFGS-B", bounds=bounds,
                       options=dict(maxiter=2000, ftol=1e-15, gtol=1e-9))
        assert res.fun >= psi_star - 1e-5 * max(1.0, abs(psi_star))
        assert np.max(np.abs(res.x - u[i])) < 2e-2


def test_warm_start_is_cheaper(solved):
    cfg, o, P, u, y, st = solved
    idx = np.where(st["exit_status"] == 0)[0][:8]
    u2, y2, st2 = o.solve_batch(P[idx], u0=u[idx], y0=y[idx], threads=4)
    assert st2["num_inner_iterations"].sum() < 0.5 * st["num_inner_iterations"][idx].sum()
    assert np.max(np.abs(u2 - u[idx])) < 1e-2


def test_box_projection_known_answer():
    """Convex known answer: with only the velocity-tracking weight and a reference whose slope the
    acceleration box admits, the optimum is v_t = clip(vref_t, U) and omega stays at its start."""
    cfg = named_config("default")
    o = oracle_for(cfg)
    P = synthetic_batch(cfg, 11, 2, 3)
    vref = np.linspace(0.2, 2.5, 20)              # 0.12 per step < lin_acc_max * ts = 0.2
    P[:, 10:20] = 0.0
    P[:, 11] = 10.0                               # qv only
    P[:, 3] = vref[0]                             # v_{-1}
    P[:, 4] = 0.0
    P[:, 20:40] = vref[None, :]
    P[:, 40:70] = 0.0                             # no circles
    u, y, st = o.solve_batch(P, threads=2)
    assert np.all(st["exit_status"] == 0)
    np.testing.assert_allclose(u[:, 0::2], np.tile(np.minimum(vref, cfg.lin_vel_max), (2, 1)), atol=2e-4)
    assert np.abs(u[:, 1::2]).max() <= 1e-12


def test_nonfinite_cost_ends_the_solve():
    """A penalty that overflows psi while the clamped controls stay finite: NotConvergedNotFiniteComputation (4).  OpEn
    checks u only; the deviation is listed in include/nmpc_solver.h and the kernels do the same (tests/test_gpu_parity.py)."""
    cfg = named_config("default")
    P = synthetic_batch(cfg, 11, 16, 4242)
    u, y, st = oracle_for(cfg).solve_batch(P, c0=np.full(16, 1e308), threads=4)
    assert (st["exit_status"] == 4).sum() >= 12 and np.all(np.isfinite(u))
    assert not np.any(np.isfinite(st["cost"][st["exit_status"] == 4]))


def _two_loop(S, Y, r):
    """The two-loop recursion of OpEn's lbfgs crate (apply_hessian) in plain numpy; S, Y: the stored pairs, NEWEST first."""
    q = r.copy()
    rho = [1.0 / float(y @ s) for s, y in zip(S, Y)]
    alpha = []
    for s, y, rh in zip(S, Y, rho):
        a = rh * float(s @ q)
        alpha.append(a)
        q -= a * y
    if len(S):
        q *= float(S[0] @ Y[0]) / float(Y[0] @ Y[0])
    for s, y, rh, a in reversed(list(zip(S, Y, rho, alpha))):
        b = rh * float(y @ q)
        q += (a - b) * s
    return q


@pytest.mark.parametrize("N", [20, 24, 27, 40, 7])
@pytest.mark.parametrize("m", [2, 7, 10])
def test_gram_form_lbfgs_is_the_two_loop_recursion(N, m):
    """The oracle (and the kernels it mirrors) applies the L-BFGS operator in the Gram form: one batch of inner products, two ten-step
    recurrences (nmpc_solve_hyb.h, nmpc_solve_hyb2.h).  Algebraically that IS the two-loop recursion of the lbfgs crate -- checked here
    deterministically on random well-conditioned pairs: partly filled buffers, full ones, and the eviction of the oldest pair (npush > m)."""
    from mpc_trajectory_generator_amd.config import load_config
    cfg = load_config(N_hor=N)
    orc = oracle_for(cfg)
    rng = np.random.default_rng(100 * N + m)
    for npush in (0, 1, m - 1, m, m + 3, 2 * m + 1):
        S = rng.standard_normal((npush, 2 * N))
        A = rng.standard_normal((2 * N, 2 * N))
        Hs = A @ A.T / (2 * N) + np.eye(2 * N)                      # y = H s with H positive definite: <y, s> > 0
        Y = S @ Hs
        r = rng.standard_normal(2 * N)
        d = orc.lbfgs_gram(m, S, Y, r)
        keep = min(m, npush)
        ref = _two_loop(list(S[::-1][:keep]), list(Y[::-1][:keep]), r)
        assert np.allclose(d, ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max()), (npush, np.abs(d - ref).max())


def test_horizons_beyond_the_kernels_are_refused():
    """One arithmetic behind the ABI: N_hor <= 40 (include/nmpc_solver.h NMPC_MAX_HORIZON); the oracle refuses what no kernel serves."""
    from mpc_trajectory_generator_amd.config import load_config
    cfg = load_config(N_hor=41)
    with pytest.raises(RuntimeError):
        oracle_for(cfg).solve_batch(synthetic_batch(cfg, 11, 2, 7))
