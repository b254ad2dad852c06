#!/usr/bin/env python3
"""Record what the reference's REAL solver does, so that this project's PANOC/ALM restatement can be pinned.

NOT runnable in the build container of this project (no cargo, no opengen, no casadi, no network).  Run it on any
machine that has the reference checkout plus its environment (env/environment.yml: python 3.7, opengen==0.6.4,
casadi, a Rust toolchain; extremitypathfinder + pyclipper for the reference's own A* front-end):

    PYTHONPATH=<reference>/src python scripts/capture_open_trace.py --reference <reference> \
        --scene 1 --config configs/default.yaml --out tests/golden/open_trace_scene1.npz

What it does: imports the reference's unmodified ``PathGenerator`` (src/path_generator.py), lets it build the OpEn
solver (``MpcModule.build``, src/mpc/mpc_generator.py:66-193) and run its closed loop on the scene
(``PathGenerator.run``, :197-437), while a thin wrapper around ``og.tcp.OptimizerTcpManager.call`` records every
request / response pair: the parameter vector ``p`` and, from OpEn's reply, ``solution``, ``exit_status``,
``num_outer_iterations``, ``num_inner_iterations``, ``last_problem_norm_fpr``, ``f2_norm``,
``delta_y_norm_over_c`` (``f1_infeasibility`` in later opengen versions), ``penalty``, ``lagrange_multipliers``,
``solve_time_ms``.  It also stores the versions found (opengen, and the ``optimization_engine`` / ``lbfgs`` crates
from the generated Cargo.lock) because the reference pins neither crate.

The output is data only (arrays + a few strings).  tests/test_open_trace.py consumes
``tests/golden/open_trace_*.npz`` when present: it replays the recorded parameter sequence through the oracle
under every combination of the restatement switches (include/nmpc_solver.h: akkt_gradient, ls_failure,
inner_status; tcp_shim: keep_multipliers) and reports which combination reproduces OpEn's iteration counts, exit
statuses and solutions -- that combination then becomes the default, and the kernels follow bit for bit.
Defaults in force (nmpc_default_opts / orc_default_opts): akkt_gradient = 1, ls_failure = 0, inner_status = 0, max_total_inner = 0;
akkt_gradient is the switch to settle first (the headline's mean iteration count hangs on it by a factor of fifteen).  Round-off level
choices of the restatement (Gram-form L-BFGS, ||r|| < eps gamma for the AKKT residual, the envelope's 0.5 / gamma
factor, the C-BFGS test without its division) are not switches: a trace agrees with them to solver tolerance, not bit for bit.
"""
import argparse
import glob
import os
import re
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of wljungbergh/mpc-trajectory-generator")
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--config", default="configs/default.yaml")
    ap.add_argument("--out", default="tests/golden/open_trace_scene1.npz")
    ap.add_argument("--no-build", action="store_true", help="reuse an existing mpc_build/ instead of regenerating the solver")
    ap.add_argument("--max-calls", type=int, default=0, help="stop the closed loop after this many solver calls (0: run to the goal)")
    args = ap.parse_args()

    ref = os.path.abspath(args.reference)
    sys.path.insert(0, os.path.join(ref, "src"))
    os.chdir(ref)                                      # the reference resolves build_directory relative to the cwd

    import opengen as og
    from path_generator import PathGenerator           # reference src/path_generator.py, unmodified
    from utils.config import Configurator              # reference src/utils/config.py
    from visibility.graphs import Graphs               # reference src/visibility/graphs.py

    rec = {k: [] for k in ("p", "solution", "exit_status", "num_outer_iterations", "num_inner_iterations",
                           "last_problem_norm_fpr", "f2_norm", "delta_y_norm_over_c", "penalty",
                           "lagrange_multipliers", "solve_time_ms")}

    real_call = og.tcp.OptimizerTcpManager.call

    class Stop(Exception):
        pass

    def recording_call(self, p, *a, **k):
        resp = real_call(self, p, *a, **k)
        if resp.is_ok():
            s = resp.get()
            rec["p"].append(np.array(p, dtype=np.float64))
            rec["solution"].append(np.array(s.solution, dtype=np.float64))
            rec["exit_status"].append(str(s.exit_status))
            rec["num_outer_iterations"].append(int(s.num_outer_iterations))
            rec["num_inner_iterations"].append(int(s.num_inner_iterations))
            rec["last_problem_norm_fpr"].append(float(s.last_problem_norm_fpr))
            rec["f2_norm"].append(float(s.f2_norm))
            dy = getattr(s, "f1_infeasibility", None)
            if dy is None:
                dy = getattr(s, "delta_y_norm_over_c", np.nan)
            rec["delta_y_norm_over_c"].append(float(dy))
            rec["penalty"].append(float(s.penalty))
            rec["lagrange_multipliers"].append(np.array(s.lagrange_multipliers, dtype=np.float64))
            rec["solve_time_ms"].append(float(s.solve_time_ms))
            if args.max_calls and len(rec["p"]) >= args.max_calls:
                raise KeyboardInterrupt          # the reference's loop handles this: kills the server, returns (:405-415)
        return resp

    og.tcp.OptimizerTcpManager.call = recording_call

    config = Configurator(os.path.join(ref, args.config)).configurate()
    g = Graphs().get_graph(complexity=args.scene)
    gen = PathGenerator(config, build=not args.no_build, verbose=True)
    gen.run(g, g.start, g.end)

    versions = {"opengen": getattr(og, "__version__", "unknown")}
    for lock in glob.glob(os.path.join(ref, config.build_directory, config.optimizer_name, "Cargo.lock")):
        text = open(lock).read()
        for crate in ("optimization_engine", "lbfgs"):
            m = re.search(r'name = "%s"\s+version = "([^"]+)"' % crate, text)
            if m:
                versions[crate] = m.group(1)
    out = {k: np.array(v) for k, v in rec.items()}
    out["versions"] = np.array([f"{k}={v}" for k, v in sorted(versions.items())])
    out["scene"] = np.array(args.scene)
    out["config"] = np.array(os.path.basename(args.config))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)) or ".", exist_ok=True)
    np.savez_compressed(args.out, **out)
    print(f"recorded {len(rec['p'])} solver calls -> {args.out}   versions: {versions}")


if __name__ == "__main__":
    main()
