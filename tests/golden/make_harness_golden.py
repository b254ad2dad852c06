#!/usr/bin/env python3
"""Generate harness golden vectors by RUNNING the reference's own receding-horizon driver.

Development container only (needs /root/reference).  The reference's ``PathGenerator.run``
(src/path_generator.py:197-437), ``MpcModule.run`` / ``rough_ref`` (src/mpc/mpc_generator.py) and
``PathPreProcessor`` helpers (src/visibility/visibility.py) are imported unmodified under stub
modules for the packages that are not installed (opengen, casadi, cv2, pyclipper,
extremitypathfinder); the A* front-end (``prepare`` / ``get_initial_guess``), which needs the real
pyclipper + extremitypathfinder, is replaced on the instance by a hand-derived route.  The solver
behind the stub ``og.tcp.OptimizerTcpManager`` is this repo's CPU oracle; every (parameters, solution)
exchange is recorded, so the tests can REPLAY the recorded solutions into this repo's driver and
demand the identical parameter sequence -- independent of any solver.

Outputs (data only): tests/golden/harness_*.npz

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_harness_golden.py
"""
import math
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import Oracle  # noqa: E402


# ------------------------------------------------------------------------------------------
# stub modules
# ------------------------------------------------------------------------------------------
class _Any:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


RECORD = {"p": [], "u": [], "exit": []}
MAX_CALLS = [10 ** 9]


class _Status:
    pass


class _Response:
    def __init__(self, st):
        self._st = st

    def is_ok(self):
        return True

    def get(self):
        return self._st


class FakeManager:
    """Server semantics of SURVEY.md App. C.4 on top of the oracle: u and y persist between calls."""
    oracle = None

    def __init__(self, path):
        self.u = None
        self.y = None

    def start(self):
        o = FakeManager.oracle
        self.u, self.y = np.zeros((1, o.n_u)), np.zeros((1, o.n1))

    def ping(self):
        return {"Pong": 1}

    def kill(self):
        pass

    def call(self, p):
        if len(RECORD["p"]) >= MAX_CALLS[0]:
            raise KeyboardInterrupt                      # reference returns the partial trajectory (:405-415)
        o = FakeManager.oracle
        p = np.asarray(p, dtype=np.float64)
        u, y, st = o.solve_batch(p[None, :], u0=self.u, y0=self.y)
        self.u, self.y = u, y
        s = _Status()
        s.solution = [float(v) for v in u[0]]
        s.exit_status = ("Converged", "NotConvergedIterations")[int(st["exit_status"][0])]
        s.solve_time_ms = 1.0
        RECORD["p"].append(p.copy()), RECORD["u"].append(u[0].copy()), RECORD["exit"].append(int(st["exit_status"][0]))
        return _Response(s)


og = _mod("opengen", tcp=types.SimpleNamespace(OptimizerTcpManager=FakeManager))
_mod("casadi")
_mod("casadi.casadi")
sys.modules["casadi"].casadi = sys.modules["casadi.casadi"]
_mod("cv2")
_mod("pyclipper", PyclipperOffset=_Any, scale_to_clipper=lambda x: x, scale_from_clipper=lambda x: x,
     JT_MITER=0, ET_CLOSEDPOLYGON=0)
_mod("extremitypathfinder")
_mod("extremitypathfinder.extremitypathfinder", PolygonEnvironment=_Any)
_mod("extremitypathfinder.plotting", PlottingEnvironment=_Any, draw_prepared_map=_Any())
sys.modules["extremitypathfinder"].extremitypathfinder = sys.modules["extremitypathfinder.extremitypathfinder"]
import matplotlib  # noqa: E402
matplotlib.use("Agg")

sys.path.insert(0, os.path.join(REF, "src"))
from path_generator import PathGenerator  # noqa: E402   (the reference, unmodified)
from utils.config import Configurator  # noqa: E402
from visibility.graphs import Graphs  # noqa: E402


def ref_config(name, **over):
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        cfg = Configurator(os.path.join(REF, "configs", name)).configurate()
    cfg.update(over)
    return cfg


def run_reference(cfg, graph, start, end, path, vertices, sinus_object, max_calls):
    """-> recorded exchanges and the trajectory the reference's run() returns."""
    RECORD["p"].clear(), RECORD["u"].clear(), RECORD["exit"].clear()
    MAX_CALLS[0] = max_calls
    FakeManager.oracle = Oracle(cfg.N_hor, cfg.Nobs, cfg.Ndynobs, cfg.ts, cfg.lin_vel_min, cfg.lin_vel_max,
                                cfg.ang_vel_max, cfg.lin_acc_min, cfg.lin_acc_max, cfg.ang_acc_max)
    pg = PathGenerator(cfg, build=False, verbose=False, sinus_object=sinus_object)
    ppp = pg.ppp

    def prepare(g):                                      # what visibility.py:49-57 stores, minus the inflation
        ppp.dyn_obs_list = g.dyn_obs_list.copy()
        ppp.original_obstacle_list = g.obstacle_list.copy()
        ppp.original_boundary_coordinates = g.boundary_coordinates.copy()

    def get_initial_guess(s, e):                         # what visibility.py:81-88 returns / stores
        ppp.path, ppp.vert, ppp.vert_copy = list(path), list(vertices), list(vertices)
        return list(path), list(vertices)

    ppp.prepare, ppp.get_initial_guess = prepare, get_initial_guess
    pg.runtime_analysis = lambda *a, **k: None
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        out = pg.run(graph, list(start), list(end))
    xx, xy, uv, uw = out[0], out[1], out[2], out[3]
    return (np.array(RECORD["p"]), np.array(RECORD["u"]), np.array(RECORD["exit"]),
            np.array(xx), np.array(xy), np.array(uv), np.array(uw), pg)


def main():
    graphs = Graphs()
    # ---- scene 1, default.yaml: the geometry-derived route of SURVEY.md section 8d (config 0) ----
    cfg = ref_config("default.yaml")
    g1 = graphs.get_graph(1)
    path1 = [(1.0, 5.0), (4.5, 15.5), (7.5, 15.5), (11.5, 12.0), (19.0, 10.0)]
    vert1 = [(5.0, 15.0), (7.0, 15.0), (12.0, 12.5)]
    P, U, E, xx, xy, uv, uw, pg = run_reference(cfg, g1, g1.start, g1.end, path1, vert1, False, 10 ** 9)
    xr, yr, tr = pg.mpc_generator.rough_ref((g1.start[0], g1.start[1]), path1[1:])
    bv, bd = pg.get_brake_vel_ref()
    print("scene 1: steps", len(P), "final", xx[-1], xy[-1], "exit counts", np.bincount(E))
    np.savez_compressed(os.path.join(OUT, "harness_scene1.npz"), params=P, solutions=U, exit=E, xx=xx, xy=xy, uv=uv,
                        uw=uw, x_ref=xr, y_ref=yr, theta_ref=tr, brake_vel=bv, brake_dist=bd,
                        start=np.array(g1.start, dtype=float), end=np.array(g1.end, dtype=float),
                        path=np.array(path1), vertices=np.array(vert1))

    # ---- scene 12's dynamic obstacles (graphs.py:182-187), sinus_object, num_steps_taken = 2, more vertices than Nobs slots ----
    cfg2 = ref_config("default.yaml", num_steps_taken=2, Nobs=4)
    g12 = graphs.get_graph(12)
    start, end = g12.start, g12.end
    path12 = [(18.9, 7.0), (23.0, 11.0), (23.0, 26.0), (27.2, 26.3), (38.6, 26.3), (44.7, 6.8)]
    vert12 = [(22.2, 11.8), (22.2, 15.9), (22.2, 20.4), (22.2, 25.0), (28.0, 25.5), (37.8, 25.5)]
    P, U, E, xx, xy, uv, uw, pg = run_reference(cfg2, g12, start, end, path12, vert12, True, 45)
    print("scene 12: steps", len(P), "last", xx[-1], xy[-1], "exit counts", np.bincount(E))
    preds = pg.ppp.get_dyn_obstacle(1.4, 20, True)
    preds1 = pg.ppp.get_dyn_obstacle(3.0, 1, False)
    fcv = [pg.ppp.find_closest_vertices(pos, 4, 0) for pos in [(19.0, 7.0), (22.5, 18.0), (30.0, 26.0), (44.0, 8.0)]]
    fcv2 = [pg.ppp.find_closest_vertices(pos, 5, 2) for pos in [(19.0, 7.0), (22.5, 18.0), (30.0, 26.0), (44.0, 8.0)]]
    np.savez_compressed(os.path.join(OUT, "harness_scene12.npz"), params=P, solutions=U, exit=E, xx=xx, xy=xy, uv=uv, uw=uw,
                        start=np.array(start, dtype=float), end=np.array(end, dtype=float), path=np.array(path12),
                        vertices=np.array(vert12), dyn_pred_t1p4_h20=np.array(preds, dtype=float),
                        dyn_pred_t3_h1=np.array(preds1, dtype=float),
                        dyn_obs=np.array([[o[0][0], o[0][1], o[1][0], o[1][1], o[2], o[3], o[4], o[5]] for o in g12.dyn_obs_list]),
                        fcv_len=np.array([len(v) for v in fcv]), fcv_first=np.array([v[0] if len(v) else (np.nan, np.nan) for v in fcv]),
                        fcv2_len=np.array([len(v) for v in fcv2]), fcv2_first=np.array([v[0] if len(v) else (np.nan, np.nan) for v in fcv2]))
    # Appendix F anchors
    xr, yr, tr = pg.mpc_generator.rough_ref((1, 1), [(4, 1), (4, 3)])
    print("rough_ref anchor:", len(xr), xr[:2], (xr[-1], yr[-1], tr[-1]))
    assert not [d for d, _, _ in os.walk(REF) if d.endswith("__pycache__")], "bytecode written into the reference"


if __name__ == "__main__":
    main()
