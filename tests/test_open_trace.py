"""Pins the solver layer to OpEn WHEN a trace is available.

``scripts/capture_open_trace.py`` records, on a machine with cargo + opengen==0.6.4, every (parameters -> solution,
iteration counts, exit status, multipliers) exchange of the reference's own closed loop.  No such machine was
available to this project, so ``tests/golden/open_trace_*.npz`` does not exist yet and the
consuming tests SKIP -- the solver layer stays "parity unpinned" and says so.  When a trace is dropped in, the tests
replay it through the oracle under every combination of the restatement switches and demand that the combination
the project ships as default reproduces OpEn (iteration counts exactly or within the stated slack, solutions within
TOL); the replay logic itself is exercised on CPU against a synthetic trace made by the oracle.
"""
import glob
import itertools
import os

import numpy as np
import pytest

from conftest import GOLDEN, oracle_for
from mpc_trajectory_generator_amd import _lib, named_config
from mpc_trajectory_generator_amd.harness import synthetic_batch

TOL_U = 1e-6            # max |u - u_OpEn| per control: different summation order / libm sin, cos, same algorithm
ITER_SLACK = 0.02       # relative slack on iteration counts (round-off can flip a line-search decision late in a solve)

SWITCH_GRID = [dict(akkt_gradient=a, ls_failure=l, inner_status=i, keep_multipliers=k)
               for a, l, i, k in itertools.product((0, 1, 2), (0, 1), (0, 1), (True, False))]
EXIT = {s: i for i, s in enumerate(_lib.EXIT_STATUS)}


def replay(trace, cfg, switches):
    """The reference's sequential semantics: u persists between calls, y persists or resets,
    c restarts.  Returns per-call arrays for comparison with the trace."""
    sw = dict(switches)
    keep_y = sw.pop("keep_multipliers")
    o = oracle_for(cfg, **sw)
    u = np.zeros((1, cfg.n_u))
    y = np.zeros((1, cfg.n1))
    out = {"u": [], "inner": [], "outer": [], "exit": []}
    for p in trace["p"]:
        if not keep_y:
            y = np.zeros_like(y)
        u, y, st = o.solve_batch(p[None, :], u0=u, y0=y)
        out["u"].append(u[0].copy())
        out["inner"].append(int(st["num_inner_iterations"][0]))
        out["outer"].append(int(st["num_outer_iterations"][0]))
        out["exit"].append(int(st["exit_status"][0]))
        # the reference applies OpEn's solution, so the next warm start is OpEn's u, not ours
        if "solution" in trace:
            u = np.array(trace["solution"][len(out["u"]) - 1], dtype=np.float64)[None, :]
            if keep_y and "lagrange_multipliers" in trace:
                y = np.array(trace["lagrange_multipliers"][len(out["u"]) - 1], dtype=np.float64)[None, :]
    return {k: np.array(v) for k, v in out.items()}


def score(trace, rep):
    """(fraction of calls whose exit status matches, mean relative iteration-count error, max control error)."""
    ex = np.array([EXIT.get(str(s), -1) for s in trace["exit_status"]])
    it = np.asarray(trace["num_inner_iterations"], dtype=np.float64)
    return (float((ex == rep["exit"]).mean()),
            float(np.mean(np.abs(rep["inner"] - it) / np.maximum(it, 1.0))),
            float(np.max(np.abs(rep["u"] - np.asarray(trace["solution"])))))


def rank_variants(trace, cfg):
    rows = []
    for sw in SWITCH_GRID:
        rows.append((score(trace, replay(trace, cfg, sw)), sw))
    rows.sort(key=lambda r: (-r[0][0], r[0][1], r[0][2]))
    return rows


def test_replay_logic_on_a_synthetic_trace():
    """A 'trace' produced by the oracle under a NON-default variant: the ranking must score exactly that variant
    (and only variants indistinguishable from it on this trace) with zero error, and the others worse -- so a real
    OpEn trace would identify the right switches."""
    cfg = named_config("cfg1")
    truth = dict(akkt_gradient=0, ls_failure=1, inner_status=0, keep_multipliers=False)
    P = synthetic_batch(cfg, 11, 3, 5)
    sw = dict(truth)
    sw.pop("keep_multipliers")
    o = oracle_for(cfg, **sw)
    trace = {k: [] for k in ("p", "solution", "exit_status", "num_inner_iterations", "lagrange_multipliers")}
    u = np.zeros((1, cfg.n_u))
    for p in P:
        u, y, st = o.solve_batch(p[None, :], u0=u, y0=np.zeros((1, cfg.n1)))
        trace["p"].append(p)
        trace["solution"].append(u[0].copy())
        trace["exit_status"].append(_lib.EXIT_STATUS[int(st["exit_status"][0])])
        trace["num_inner_iterations"].append(int(st["num_inner_iterations"][0]))
        trace["lagrange_multipliers"].append(y[0].copy())
    rows = rank_variants(trace, cfg)
    perfect = [sw_ for sc, sw_ in rows if sc == (1.0, 0.0, 0.0)]
    assert truth in perfect and rows[0][0] == (1.0, 0.0, 0.0)
    assert all(sw_["akkt_gradient"] == 0 and sw_["ls_failure"] == 1 for sw_ in perfect)     # the trace tells them apart
    assert len(perfect) < len(rows)


TRACES = sorted(glob.glob(os.path.join(GOLDEN, "open_trace_*.npz")))


@pytest.mark.skipif(not TRACES, reason="no OpEn trace committed (tests/golden/open_trace_*.npz): solver-layer parity is UNPINNED; "
                                       "run scripts/capture_open_trace.py on a machine with cargo + opengen==0.6.4")
@pytest.mark.parametrize("path", TRACES or ["-"])
def test_shipped_variant_reproduces_open(path):
    d = np.load(path, allow_pickle=False)
    trace = {k: d[k] for k in d.files}
    name = str(trace.get("config", "default.yaml"))
    cfg = named_config({"default.yaml": "cfg1", "jconf_3.yaml": "cfg1"}.get(name, "cfg1"))
    shipped = dict(akkt_gradient=1, ls_failure=0, inner_status=0, keep_multipliers=True)
    rows = rank_variants(trace, cfg)
    table = "\n".join(f"{sc}  {sw}" for sc, sw in rows[:8])
    sc = score(trace, replay(trace, cfg, shipped))
    assert sc[0] >= 0.98 and sc[1] <= ITER_SLACK and sc[2] <= TOL_U, \
        f"shipped restatement {shipped} scores {sc} against {os.path.basename(path)}; best variants:\n{table}"
