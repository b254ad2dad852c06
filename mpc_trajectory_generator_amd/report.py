"""Run-time report of a trajectory run: counterpart of the reference's ``runtime_analysis`` and
``plot_solver_performance`` / ``plot_solver_hist`` (src/path_generator.py:190-195,479-568).

The reference prints a table of launch / front-end / MPC / solver times with mean, max and the number
of solver calls, optionally appends it to a file, and draws a histogram of per-step loop times.
Here the same quantities come back as text and as histogram arrays (plotting stays with the caller;
matplotlib is not a dependency of this package).
"""
from __future__ import annotations

import numpy as np


def runtime_analysis(time_dict: dict, solver_times, overhead_times=None, file_name: str = "") -> str:
    """Text block with the keys the reference reports (path_generator.py:479-533); appended to
    ``file_name`` when given, like the reference does."""
    st = np.asarray(solver_times, dtype=np.float64)
    lines = ["Runtime Analysis", "-" * 44]
    for key, label in (("opt_launch", "Launching optimizer"), ("prepare", "Prepare visibility graph"),
                       ("initial_guess", "Finding A* solution"), ("rough_ref", "Generating rough reference"),
                       ("mpc_time", "MPC loop"), ("solver_time", "  of which solver"), ("total_time", "Total")):
        if key in time_dict:
            lines.append(f"{label:<30}{float(time_dict[key]):>12.1f} ms")
    if st.size:
        lines += ["-" * 44, f"{'Solver calls':<30}{st.size:>12d}", f"{'Mean solver time':<30}{st.mean():>12.3f} ms",
                  f"{'Median solver time':<30}{np.median(st):>12.3f} ms", f"{'Max solver time':<30}{st.max():>12.3f} ms"]
    if overhead_times is not None and len(overhead_times):
        ov = np.asarray(overhead_times, dtype=np.float64)
        lines.append(f"{'Mean loop overhead':<30}{ov.mean():>12.3f} ms")
    text = "\n".join(lines) + "\n"
    if file_name:
        with open(file_name, "a") as fh:
            fh.write(text)
    return text


def loop_time_histogram(solver_times, overhead_times, bins: int = 30):
    """Histogram of per-step loop time = solver + overhead, what ``gen_runtime_plots.py:28-33`` plots.
    -> (counts, bin_edges)."""
    total = np.asarray(solver_times, dtype=np.float64) + np.asarray(overhead_times, dtype=np.float64)
    return np.histogram(total, bins=bins)


def batch_summary(status) -> dict:
    """Summary of a batched solve (status structured array): the numbers bench.py reports."""
    it = status["num_inner_iterations"].astype(np.float64)
    return {"solves": int(status.shape[0]), "converged_frac": float((status["exit_status"] == 0).mean()),
            "mean_inner_iters": float(it.mean()), "p50_inner_iters": float(np.median(it)),
            "p99_inner_iters": float(np.percentile(it, 99)), "max_inner_iters": int(it.max()),
            "mean_outer_iters": float(status["num_outer_iterations"].mean()),
            "exit_status_counts": {int(k): int(v) for k, v in zip(*np.unique(status["exit_status"], return_counts=True))}}
