"""CPU: the oracle against the golden vectors produced by executing the reference's own
MpcModule.build() (tests/golden/make_golden.py), plus known answers from SURVEY.md App. F."""
import numpy as np
import pytest

from conftest import VARIANTS, oracle_for
from mpc_trajectory_generator_amd import named_config

RTOL = 1e-11   # oracle vs reference-derived goldens: differs only by summation order / sincos rounding


@pytest.mark.parametrize("name", list(VARIANTS))
def test_sizes_match_reference(golden, name):
    d, cfg = golden[name], named_config(VARIANTS[name])
    o = oracle_for(cfg)
    assert (o.n_u, o.n_p) == (d["u"].shape[1], d["p"].shape[1])
    assert (o.n1, o.n2) == (d["F1"].shape[1], d["F2"].shape[1])
    kv = dict(zip(d["cfg_keys"], d["cfg_vals"]))
    for k in ("N_hor", "Nobs", "Ndynobs", "ts", "lin_vel_min", "lin_vel_max", "lin_acc_min", "lin_acc_max",
              "ang_vel_max", "ang_acc_max"):
        assert float(cfg[k]) == kv[k], k
    # U and C boxes as the reference's og.constraints.Rectangle received them (mpc_generator.py:151-168)
    N = cfg.N_hor
    assert np.array_equal(d["umin"], np.tile([cfg.lin_vel_min, -cfg.ang_vel_max], N))
    assert np.array_equal(d["umax"], np.tile([cfg.lin_vel_max, cfg.ang_vel_max], N))
    assert np.array_equal(d["cmin"], np.r_[[cfg.lin_acc_min] * N, [-cfg.ang_acc_max] * N])
    assert np.array_equal(d["cmax"], np.r_[[cfg.lin_acc_max] * N, [cfg.ang_acc_max] * N])


@pytest.mark.parametrize("name", list(VARIANTS))
def test_cost_constraints_gradient(golden, name):
    d, cfg = golden[name], named_config(VARIANTS[name])
    o = oracle_for(cfg)
    for i in range(len(d["u"])):
        f, g, F1, F2 = o.eval(d["p"][i], d["u"][i])
        assert abs(f - d["f"][i]) <= RTOL * abs(d["f"][i])
        assert np.max(np.abs(g - d["grad_f"][i])) <= RTOL * np.max(np.abs(d["grad_f"][i]))
        assert np.max(np.abs(F1 - d["F1"][i])) <= RTOL * max(1.0, np.max(np.abs(d["F1"][i])))
        assert np.max(np.abs(F2 - d["F2"][i])) <= RTOL * max(1.0, np.max(np.abs(d["F2"][i])))
        for j, (c, y) in enumerate(zip(d["xi_c"], d["xi_y"])):
            psi, gp, _, _ = o.eval(d["p"][i], d["u"][i], c, y)
            assert abs(psi - d["psi"][i, j]) <= RTOL * abs(d["psi"][i, j])
            assert np.max(np.abs(gp - d["grad_psi"][i, j])) <= RTOL * np.max(np.abs(d["grad_psi"][i, j]))


def test_survey_known_answer(golden):
    # default.yaml, rng(0), u~U(-0.5,1.5)^40, p~U(0.1,2.0)^430
    d = golden["default"]
    assert abs(d["f"][0] - 393.0070140420273) < 1e-9
    np.testing.assert_allclose(d["grad_f"][0][:4], [51.27815109, 14.78224757, -56.61174672, -4.38617904], rtol=1e-8)
    o = oracle_for(named_config("default"))
    f, g, _, _ = o.eval(d["p"][0], d["u"][0])
    assert abs(f - 393.0070140420273) < 1e-9


def test_gradient_finite_differences(golden):
    d, cfg = golden["smooth"], named_config("cfg4")
    o = oracle_for(cfg)
    p, u = d["p"][2], d["u"][2].copy()
    c, y = 7.0, d["xi_y"][1]
    _, g, _, _ = o.eval(p, u, c, y)
    h = 1e-6
    for i in (0, 1, 7, 18, 39):
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd = (o.eval(p, up, c, y)[0] - o.eval(p, um, c, y)[0]) / (2 * h)
        assert abs(fd - g[i]) <= 1e-6 * max(1.0, abs(g[i]))


def test_sincos_primitive():
    o = oracle_for(named_config("default"))
    x = np.random.default_rng(0).uniform(-50, 50, 2000)
    sc = np.array([o.sincos(v) for v in x])
    assert np.max(np.abs(sc[:, 0] - np.sin(x))) < 3e-16
    assert np.max(np.abs(sc[:, 1] - np.cos(x))) < 3e-16


def test_tree_sum_shape():
    o = oracle_for(named_config("default"))
    v = np.random.default_rng(1).normal(size=20)
    w = np.zeros(32)
    w[:20] = v
    while len(w) > 1:
        w = w[0::2] + w[1::2]
    assert o.tree_sum(v) == w[0]
