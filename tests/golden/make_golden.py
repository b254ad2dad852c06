#!/usr/bin/env python3
"""Generate cost-layer golden vectors by EXECUTING the reference's own problem definition.

Runs only in the development container (needs /root/reference); the produced
``tests/golden/cost_*.npz`` files are data (inputs + expected outputs) and are what
travels.  Nothing from the reference is copied: its unmodified
``MpcModule.build()`` (reference src/mpc/mpc_generator.py:66-193) is imported and run with
stand-in ``casadi`` / ``opengen`` modules whose symbolic type is a thin wrapper over
torch float64 tensors, so that the reference code itself evaluates
f(u;p), F1(u;p), F2(u;p), U and C numerically and torch autograd differentiates them.

On top of the captured (f, F1, F2, C) the script forms the augmented cost the
OpEn code generator would have built (OpEn ALM documentation):

    psi(u; c, y, p) = f + c/2 * [ dist^2_C(F1 + y / max(c, 1)) + ||F2||^2 ]

and differentiates that too.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import math
import os
import sys
import types

import numpy as np
import torch
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)


# ----------------------------------------------------------------------------------------
# stand-in "casadi": the complete surface mpc_generator.py touches
# ----------------------------------------------------------------------------------------
class SX:
    """Numeric stand-in for casadi.SX backed by a torch f64 tensor (0-d or 1-d)."""

    __array_priority__ = 1000

    def __init__(self, t):
        self.t = t if isinstance(t, torch.Tensor) else torch.as_tensor(float(t))

    # casadi.SX.sym(name, n) -> vector of caller supplied values
    _values = {}

    @staticmethod
    def sym(name, n):
        v = SX._values[name]
        assert v.numel() == n, (name, v.numel(), n)
        return SX(v)

    @staticmethod
    def ones(n):
        return SX(torch.ones(n))

    @staticmethod
    def _raw(o):
        return o.t if isinstance(o, SX) else torch.as_tensor(float(o))

    def __getitem__(self, k):
        return SX(self.t[k])

    @property
    def T(self):
        return self

    def __add__(self, o): return SX(self.t + SX._raw(o))
    def __radd__(self, o): return SX(SX._raw(o) + self.t)
    def __sub__(self, o): return SX(self.t - SX._raw(o))
    def __rsub__(self, o): return SX(SX._raw(o) - self.t)
    def __mul__(self, o): return SX(self.t * SX._raw(o))
    def __rmul__(self, o): return SX(SX._raw(o) * self.t)
    def __truediv__(self, o): return SX(self.t / SX._raw(o))
    def __rtruediv__(self, o): return SX(SX._raw(o) / self.t)
    def __neg__(self): return SX(-self.t)

    def __pow__(self, e):
        assert e == 2
        return SX(self.t * self.t)
    # deliberately no __iadd__: the reference does ``x += ...`` on values sliced from z0


def _cat(*xs):
    return SX(torch.cat([torch.atleast_1d(SX._raw(x)) for x in xs]))


cs = types.ModuleType("casadi.casadi")
cs.SX = SX
cs.cos = lambda a: SX(torch.cos(SX._raw(a)))
cs.sin = lambda a: SX(torch.sin(SX._raw(a)))
cs.fmax = lambda a, b: SX(torch.maximum(*torch.broadcast_tensors(SX._raw(a), SX._raw(b))))
cs.fmin = lambda a, b: SX(torch.minimum(*torch.broadcast_tensors(SX._raw(a), SX._raw(b))))
cs.vertcat = _cat
cs.horzcat = _cat
cs.dot = lambda a, b: SX((SX._raw(a) * SX._raw(b)).sum())
cs.mtimes = lambda a, b: SX((SX._raw(a) * SX._raw(b)).sum())
cs.mmin = lambda a: SX(SX._raw(a).min())
casadi = types.ModuleType("casadi")
casadi.casadi = cs


# ----------------------------------------------------------------------------------------
# stand-in "opengen": captures what build() hands to the code generator
# ----------------------------------------------------------------------------------------
CAPTURED = {}


class _Chain:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: self


class _Rectangle:
    def __init__(self, xmin, xmax):
        self.xmin, self.xmax = list(xmin), list(xmax)


class _Problem:
    def __init__(self, u, p, cost):
        CAPTURED.clear()
        CAPTURED.update(u=u, p=p, cost=cost)

    def with_penalty_constraints(self, f2):
        CAPTURED["F2"] = f2
        return self

    def with_constraints(self, U):
        CAPTURED["U"] = U
        return self

    def with_aug_lagrangian_constraints(self, f1, C):
        CAPTURED["F1"], CAPTURED["C"] = f1, C
        return self


og = types.ModuleType("opengen")
og.builder = types.SimpleNamespace(Problem=_Problem, OpEnOptimizerBuilder=_Chain)
og.constraints = types.SimpleNamespace(Rectangle=_Rectangle)
og.config = types.SimpleNamespace(BuildConfiguration=_Chain, OptimizerMeta=_Chain,
                                  SolverConfiguration=_Chain)

sys.modules["casadi"] = casadi
sys.modules["casadi.casadi"] = cs
sys.modules["opengen"] = og
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))
from mpc.mpc_generator import MpcModule  # noqa: E402  (the reference, unmodified)


class dotdict(dict):
    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


def load_cfg(name, **over):
    with open(os.path.join(REF, "configs", name)) as fh:
        cfg = dotdict(yaml.safe_load(fh))
    cfg.update(over)
    return cfg


def variants():
    """The four problem shapes of BASELINE.md section 4 (overrides per SURVEY.md App. E)."""
    d = load_cfg("default.yaml")
    yield "default", d
    yield "n40", load_cfg("jconf_3.yaml", N_hor=40)
    yield "nobs50", load_cfg("default.yaml", Nobs=50)
    s = load_cfg("smooth_velocity.yaml")
    s.update(N_hor=20, nz=20, qv=d.qv, vel_red_steps=d.vel_red_steps, Ndynobs=d.Ndynobs)
    yield "smooth", s


def reference_eval(cfg, u, p):
    """Run the reference's build() on numeric (u, p); return torch scalars/vectors."""
    SX._values = {"u": u, "z0": p}
    MpcModule(cfg).build()
    c = CAPTURED
    return (c["cost"].t, c["F1"].t, c["F2"].t, c["U"], c["C"])


def psi_of(f, F1, F2, C, c, y):
    lo = torch.tensor(C.xmin)
    hi = torch.tensor(C.xmax)
    t = F1 + y / max(c, 1.0)
    s = t - torch.minimum(torch.maximum(t, lo), hi)
    return f + 0.5 * c * ((s * s).sum() + (F2 * F2).sum())


def weights_of(cfg):
    # reference src/path_generator.py:226-227
    return [cfg.q, cfg.qv, cfg.qtheta, cfg.lin_vel_penalty, cfg.ang_vel_penalty,
            cfg.qN, cfg.qthetaN, cfg.cte_penalty, cfg.lin_acc_penalty, cfg.ang_acc_penalty]


def realistic_case(cfg, rng, all_weights):
    """A parameter vector shaped like what the receding-horizon loop sends
    (layout: SURVEY.md App. A), with obstacles placed so penalties are active."""
    N, Nobs, Nd = cfg.N_hor, cfg.Nobs, cfg.Ndynobs
    x0, y0, th0 = rng.uniform(2, 18), rng.uniform(2, 18), rng.uniform(-math.pi, math.pi)
    last_u = [rng.uniform(0, 1.5), rng.uniform(-0.3, 0.3)]
    # reference polyline: starts near the state, heading near th0, two gentle bends
    step = 0.33
    hd = th0 + rng.normal(0, 0.2)
    px, py = x0 + rng.normal(0, 0.1), y0 + rng.normal(0, 0.1)
    n_real = N if rng.uniform() < 0.5 else int(rng.integers(N // 2, N))
    rx, ry, rt = [], [], []
    for i in range(N):
        if i < n_real:
            if i in (N // 3, 2 * N // 3):
                hd += rng.uniform(-0.9, 0.9)
            px, py = px + step * math.cos(hd), py + step * math.sin(hd)
        rx.append(px), ry.append(py), rt.append(hd)   # tail repeats the end pose
    xf = [rx[-1], ry[-1], rt[-1]]
    w = list(rng.uniform(0.5, 20.0, 10)) if all_weights else weights_of(cfg)
    base = cfg.lin_vel_max * cfg.throttle_ratio
    vel_ref = [base] * N
    if n_real < N:
        for i in range(n_real, N):
            vel_ref[i] = max(0.0, base * (1 - (i - n_real + 1) / max(1, N - n_real)))
    # a plausible rollout of plausible controls, to place obstacles on it
    u = np.empty(2 * N)
    u[0::2] = rng.uniform(0.2, 1.5, N)
    u[1::2] = rng.uniform(-0.5, 0.5, N)
    xs, ys, th = [], [], th0
    x, y = x0, y0
    for t in range(N):
        x += cfg.ts * u[2 * t] * math.cos(th)
        y += cfg.ts * u[2 * t] * math.sin(th)
        th += cfg.ts * u[2 * t + 1]
        xs.append(x), ys.append(y)
    stat = []
    n_act = int(rng.integers(1, Nobs + 1))
    for k in range(Nobs):
        if k < n_act:
            t = int(rng.integers(0, N))
            d, a = rng.uniform(0.1, 0.9), rng.uniform(0, 2 * math.pi)
            stat += [xs[t] + d * math.cos(a), ys[t] + d * math.sin(a),
                     cfg.vehicle_width / 2 + cfg.vehicle_margin]
        else:
            stat += [0.0, 0.0, 0.0]
    dyn = []
    for k in range(Nd):
        padded = (k == Nd - 1) and rng.uniform() < 0.5
        t0 = int(rng.integers(0, N))
        cx, cy = xs[t0] + rng.uniform(-1, 1), ys[t0] + rng.uniform(-1, 1)
        vx, vy = rng.uniform(-0.1, 0.1, 2)
        erx, ery, ang = rng.uniform(0.6, 1.6), rng.uniform(0.6, 1.6), rng.uniform(0, math.pi)
        for t in range(N):
            if padded:                      # reference src/path_generator.py:274-280
                dyn += [0.0, 0.0, 1.0, 1.0, 0.0]
            else:
                dyn += [cx + vx * (t - t0), cy + vy * (t - t0), erx, ery, ang]
    refs = [v for trip in zip(rx, ry, rt) for v in trip]
    p = [x0, y0, th0] + last_u + xf + last_u + w + vel_ref + stat + dyn + refs
    return u, np.array(p, dtype=np.float64)


def main():
    for name, cfg in variants():
        N = cfg.N_hor
        n_u = cfg.nu * N
        n_p = cfg.nz + N + cfg.Nobs * cfg.nobs + cfg.Ndynobs * cfg.ndynobs * N + cfg.nx * N
        n1, n2 = 2 * N, cfg.Nobs + cfg.Ndynobs
        cases_u, cases_p = [], []
        rng0 = np.random.default_rng(0)                       # SURVEY.md section 8c KAT
        cases_u.append(rng0.uniform(-0.5, 1.5, n_u))
        cases_p.append(rng0.uniform(0.1, 2.0, n_p))
        rng = np.random.default_rng(12345)                    # fixture seed
        for i in range(5):
            u, p = realistic_case(cfg, rng, all_weights=(i % 2 == 1))
            cases_u.append(u), cases_p.append(p)
        xis = []
        rngx = np.random.default_rng(777)
        xis.append((1.0, np.zeros(n1)))
        xis.append((25.0, rngx.normal(0, 1.0, n1)))
        xis.append((0.5, rngx.normal(0, 2.0, n1)))
        out = dict(u=np.array(cases_u), p=np.array(cases_p),
                   xi_c=np.array([c for c, _ in xis]), xi_y=np.array([y for _, y in xis]))
        F, G, F1s, F2s, PSI, GPSI = [], [], [], [], [], []
        for u_np, p_np in zip(cases_u, cases_p):
            u = torch.tensor(u_np, requires_grad=True)
            p = torch.tensor(p_np)
            f, F1, F2, U, C = reference_eval(cfg, u, p)
            assert F1.numel() == n1 and F2.numel() == n2
            (g,) = torch.autograd.grad(f, u, retain_graph=True)
            F.append(f.item()), G.append(g.numpy().copy())
            F1s.append(F1.detach().numpy().copy()), F2s.append(F2.detach().numpy().copy())
            ps, gps = [], []
            for c, y in xis:
                psi = psi_of(f, F1, F2, C, c, torch.tensor(y))
                (gp,) = torch.autograd.grad(psi, u, retain_graph=True)
                ps.append(psi.item()), gps.append(gp.numpy().copy())
            PSI.append(ps), GPSI.append(gps)
        out.update(f=np.array(F), grad_f=np.array(G), F1=np.array(F1s), F2=np.array(F2s),
                   psi=np.array(PSI), grad_psi=np.array(GPSI),
                   umin=np.array(U.xmin), umax=np.array(U.xmax),
                   cmin=np.array(C.xmin), cmax=np.array(C.xmax))
        keys = ["N_hor", "nu", "nx", "nz", "nobs", "Nobs", "Ndynobs", "ndynobs", "ts",
                "lin_vel_min", "lin_vel_max", "lin_acc_min", "lin_acc_max",
                "ang_vel_max", "ang_acc_max"]
        out["cfg_keys"] = np.array(keys)
        out["cfg_vals"] = np.array([float(cfg[k]) for k in keys])
        path = os.path.join(OUT, f"cost_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: n_u={n_u} n_p={n_p} n1={n1} n2={n2} f[0]={F[0]!r} "
              f"active F2 per case={[int((x > 0).sum()) for x in F2s]} -> {path}")
        if name == "default":
            assert abs(F[0] - 393.0070140420273) < 1e-9, F[0]


if __name__ == "__main__":
    main()
