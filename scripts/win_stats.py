"""Windowed cross-track search: share of evaluations that fall back to the full scan, and kernel ms, for stats builds of the
library (make -B libnmpc_ws_<tag>.so OUT=... EXTRA="-DNMPC_WIN_STATS ...").  usage: python scripts/win_stats.py cfg1 lib1.so lib2.so ..."""
import ctypes, json, os, subprocess, sys
sys.path.insert(0, ".")
if sys.argv[1] == "child":
    import numpy as np
    from mpc_trajectory_generator_amd import named_config, _lib
    from mpc_trajectory_generator_amd.solver import BatchSolver
    from mpc_trajectory_generator_amd.harness import synthetic_batch
    from mpc_trajectory_generator_amd.frontend import random_routes
    name = sys.argv[2]
    cfg = named_config(name)
    sol = BatchSolver(cfg, max_batch=8192)
    lib = _lib.load_library()
    out = {}
    P = synthetic_batch(cfg, 11, 8192, 0, routes=random_routes(cfg, 11, 32, seed=1000), synthetic_circles=name == "cfg3", random_dyn=name == "cfg4")
    sol.solve(P)
    buf = (ctypes.c_ulonglong * 4)()
    if hasattr(lib, "nmpc_debug_win_stats"):
        lib.nmpc_debug_win_stats(buf, 1)
    ms = []
    for _ in range(3):
        st = sol.solve(P)[2]
        ms.append(sol.last_batch_ms)
    if hasattr(lib, "nmpc_debug_win_stats"):
        lib.nmpc_debug_win_stats(buf, 0)
        out["searches"], out["fallbacks"] = int(buf[0]), int(buf[1])
        out["fallback_share"] = round(buf[1] / max(buf[0], 1), 4)
        out["obstacle_certificates"], out["obstacle_scans"] = int(buf[2]), int(buf[3])
        out["obstacle_scan_share"] = round(buf[3] / max(buf[2], 1), 4)
    out["kernel"], out["source_hash"], out["config"] = sol.kernel_name, _lib.source_hash(), name
    out["window_full_scan_share"] = out.get("fallback_share")
    out["ms"] = round(min(ms), 2)
    out["checksum"] = float(st["num_inner_iterations"].astype(np.float64).sum() + st["cost"].sum())
    print(json.dumps(out))
    sys.exit(0)
for lib in sys.argv[2:]:
    r = subprocess.run([sys.executable, __file__, "child", sys.argv[1]], env=dict(os.environ, NMPC_LIB_PATH=os.path.abspath(lib)), capture_output=True, text=True)
    print(lib, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-600:], flush=True)
