"""ctypes binding of libnmpc_hip.so (the C ABI of include/nmpc_solver.h).

There is no CPU fallback: if the HIP library is missing it is built with hipcc, and if that is
impossible or no MI355X is visible the constructor of the solver raises.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libnmpc_hip.so")
BUILD_INFO = os.path.join(_CSRC, "build_info.json")      # written by build_library: flags, codegen check, registers / LDS / scratch per kernel

# every symbol include/nmpc_solver.h declares
SYMBOLS = (
    "nmpc_default_opts", "nmpc_n_u", "nmpc_n_p", "nmpc_n1", "nmpc_n2", "nmpc_new", "nmpc_free",
    "nmpc_ping", "nmpc_last_error", "nmpc_abi_version", "nmpc_experiments_build", "nmpc_kernel_name", "nmpc_solve_batch_device",
    "nmpc_solve_batch_host", "nmpc_last_batch_ms", "nmpc_eval_batch_device", "nmpc_eval_batch_host",
    "nmpc_test_sincos_host", "nmpc_test_divsqrt_host",
    "nmpc_loop_new", "nmpc_loop_free", "nmpc_loop_step", "nmpc_loop_read", "nmpc_loop_params",
    "nmpc_loop_trajectory",
)

EXPECTED_ABI = 3      # the nmpc_opts / nmpc_status layouts below are written for this version of include/nmpc_solver.h

ERRORS = {0: "ok", -1: "bad problem", -2: "bad opts", -3: "bad argument", -4: "no HIP device",
          -5: "HIP runtime error", -6: "dead handle"}

EXIT_STATUS = ("Converged", "NotConvergedIterations", "NotConvergedOutOfTime", "NotConvergedCost",
               "NotConvergedNotFiniteComputation")


class NmpcProblem(C.Structure):
    _fields_ = [("N", C.c_int32), ("nobs", C.c_int32), ("ndyn", C.c_int32), ("reserved", C.c_int32),
                ("ts", C.c_double), ("vmin", C.c_double), ("vmax", C.c_double), ("wmax", C.c_double),
                ("amin", C.c_double), ("amax", C.c_double), ("awmax", C.c_double)]


class NmpcOpts(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("initial_tolerance", C.c_double),
                ("delta_tolerance", C.c_double), ("initial_penalty", C.c_double),
                ("penalty_update", C.c_double), ("tolerance_update", C.c_double),
                ("sufficient_decrease", C.c_double), ("lbfgs_memory", C.c_int32),
                ("max_inner", C.c_int32), ("max_outer", C.c_int32), ("max_total_inner", C.c_int32),
                ("akkt_gradient", C.c_int32), ("ls_failure", C.c_int32), ("inner_status", C.c_int32),
                ("reserved", C.c_int32)]

# the restatement switches (include/nmpc_solver.h) and what their values mean
VARIANT_FIELDS = {
    "akkt_gradient": ("per_trial", "step_top", "off"),
    "ls_failure": ("take_last_trial", "tau0_fb_step"),
    "inner_status": ("propagate_inner", "converged_if_outer_ok"),
}


class NmpcRoute(C.Structure):
    _dp = C.POINTER(C.c_double)
    _fields_ = [("n_ref", C.c_int32), ("n_vert", C.c_int32), ("n_brake", C.c_int32), ("num_steps_taken", C.c_int32),
                ("x_ref", _dp), ("y_ref", _dp), ("theta_ref", _dp), ("vert_xy", _dp),
                ("brake_vel", _dp), ("brake_dist", _dp),
                ("end", C.c_double * 3), ("base_speed", C.c_double), ("radius", C.c_double),
                ("dyn_pad", C.c_double), ("weights", C.c_double * 10)]


STATUS_DTYPE = np.dtype([("exit_status", "<i4"), ("num_outer_iterations", "<u4"),
                         ("num_inner_iterations", "<u4"), ("num_cost_evals", "<u4"),
                         ("num_grad_evals", "<u4"), ("reserved", "<u4"),
                         ("last_problem_norm_fpr", "<f8"), ("delta_y_norm_over_c", "<f8"),
                         ("f2_norm", "<f8"), ("penalty", "<f8"), ("cost", "<f8"),
                         ("solve_time_ms", "<f8")])
assert STATUS_DTYPE.itemsize == 72


def _sources():
    srcs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(_CSRC, "..", "..", "include", "nmpc_solver.h"))
    return srcs


def source_hash() -> str:
    """sha256 over the kernel sources: the key profiles/*/traffic.json files are stored under, so that a
    PMC figure measured on one version of the kernels is never quoted for another."""
    import hashlib
    h = hashlib.sha256()
    for s in _sources():
        if not s.endswith("Makefile"):
            with open(s, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 the kernels in-tree (cross-compiles without a GPU).  Safe to call from
    several processes at once (one rank per GPU): the build runs under a file lock into a temporary name
    and is renamed into place, the other processes find a fresh library when they get the lock."""
    import fcntl

    def stale():
        return (not os.path.exists(LIB_PATH)) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in _sources())

    if not (force or stale()):
        return LIB_PATH
    with open(os.path.join(_CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                tmp = f"libnmpc_hip.so.tmp{os.getpid()}"
                r = subprocess.run(["make", "-C", _CSRC, "-B", tmp, f"OUT={tmp}"], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("building libnmpc_hip.so failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
                # the compiler's output is checked before it is accepted: ROCm 7.2 has produced silently wrong code for these kernels twice
                # (codegen_check.py); the verdict, the flags and the kernels' resources go to build_info.json, which bench.py quotes
                from . import codegen_check
                res = codegen_check.verify()
                info = {"source_hash": source_hash(), "flags": res.get("flags"), "codegen_check": {k: v for k, v in res.items() if k != "resources"},
                        "resources": res.get("resources", {})}
                if not res["ok"]:
                    os.remove(os.path.join(_CSRC, tmp))
                    with open(BUILD_INFO + ".refused", "w") as fh:      # (build_info.json keeps describing the library that is in place)
                        json.dump(info, fh, indent=1)
                    raise RuntimeError("libnmpc_hip.so REFUSED: the compiler generated wrong code for these flags (codegen_check): " +
                                       json.dumps(info["codegen_check"])[:3000])
                os.replace(os.path.join(_CSRC, tmp), LIB_PATH)
                with open(BUILD_INFO, "w") as fh:
                    json.dump(info, fh, indent=1)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


# The same sources under the other machine-scheduler strategies: not shipped, but built and run by tests/test_gpu_strategies.py -- a kernel
# whose results depend on the schedule has a defect (or the compiler has: codegen_check.py), and the shipped strategy may only be hiding it.
STRATEGIES = {"default": [], "max-memory-clause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
              "max-ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
# The shipped library reads no environment variable.  The knobs tests and scripts need (force the run-time-shape kernel, team modes, scheduler
# and migration settings, culling radius) exist only in this variant: the shipped sources and scheduler flags + -DNMPC_EXPERIMENTS.
EXPERIMENTS = "experiments"
# Documented compile-time settings of the kernels, each with the shipped scheduler flags: tests demand the shipped library's bits from them.
# win0: -DNMPC_WIN=0 -DNMPC_WIN2=0 (both solve kernels), the cross-track search always as the full scan (the obstacle certificate and its helper-side guard stay on).
DEFINES = {"win0": ["-DNMPC_WIN=0", "-DNMPC_WIN2=0"]}


def variant_path(name: str) -> str:
    return os.path.join(_CSRC, "variants", f"libnmpc_{name}.so")


def _variant_key() -> str:
    """What a variant is fresh for: the kernel sources AND the Makefile (a flag change rebuilds the variants too)."""
    import hashlib
    with open(os.path.join(_CSRC, "Makefile"), "rb") as fh:
        return source_hash() + "+" + hashlib.sha256(fh.read()).hexdigest()[:8]


def build_variant(name: str, force: bool = False) -> dict:
    """Build csrc/variants/libnmpc_<name>.so with strategy `name` and run the code-generation check on ITS flags (the experiments variant:
    the shipped flags + -DNMPC_EXPERIMENTS); -> that check's result.  A variant whose check fails is not put in place."""
    from . import codegen_check
    import fcntl
    experiments = name == EXPERIMENTS
    defines = DEFINES.get(name)
    flags = None if (experiments or defines) else STRATEGIES[name]
    out, meta = variant_path(name), variant_path(name)[:-3] + ".json"
    os.makedirs(os.path.dirname(out), exist_ok=True)

    def fresh():
        if os.path.exists(out) and os.path.exists(meta) and not force:      # fresh = built from these very sources (content, not time stamps)
            try:
                with open(meta) as fh:
                    res = json.load(fh)
            except (OSError, ValueError):
                return None
            if res.get("variant_key") == _variant_key():
                return res
        return None

    res = fresh()
    if res is not None:
        return res
    with open(os.path.join(_CSRC, ".build.lock"), "w") as lock:      # (several test processes may ask for the same variant)
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            res = fresh()
            if res is not None:
                return res
            tmp = os.path.relpath(out, _CSRC) + f".tmp{os.getpid()}"
            cmd = ["make", "-C", _CSRC, "-B", tmp, f"OUT={tmp}"] + (["EXTRA=-DNMPC_EXPERIMENTS"] if experiments else
                                                                  ["EXTRA=" + " ".join(defines)] if defines else ["SCHED=" + " ".join(flags)])
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"building {out} failed:\n" + r.stdout[-2000:] + r.stderr[-4000:])
            res = (codegen_check.verify(codegen_check.makefile_flags() + (["-DNMPC_EXPERIMENTS"] if experiments else defines))
                   if (experiments or defines) else codegen_check.verify(flags))
            res.pop("resources", None)
            res["source_hash"] = source_hash()
            res["variant_key"] = _variant_key()
            if experiments and not res["ok"]:      # (the strategy variants are kept even when the check objects: tests/test_gpu_strategies.py reports on them)
                os.remove(os.path.join(_CSRC, tmp))
                raise RuntimeError(f"{out} REFUSED by the code-generation check: " + json.dumps({k: v for k, v in res.items() if k != "details"})[:2000])
            os.replace(os.path.join(_CSRC, tmp), out)
            with open(meta, "w") as fh:      # (under the lock: library and verdict change together)
                json.dump(res, fh)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return res


def build_info() -> dict:
    """What the library in use was built with (flags, code-generation check, resources per kernel); {} for a library of unknown origin."""
    try:
        with open(BUILD_INFO) as fh:
            info = json.load(fh)
        return info if info.get("source_hash") == source_hash() else {"stale": True, **info}
    except OSError:
        return {}


_lib = None
_lib_experiments = None


def load_library(experiments: bool = False) -> C.CDLL:
    """Load (building if needed) the HIP library; raises if that is impossible.  experiments=True: the variant with the environment knobs
    (tests and scripts only; the product never asks for it)."""
    global _lib, _lib_experiments
    if experiments:
        if _lib_experiments is None:
            build_library()
            build_variant(EXPERIMENTS)
            _lib_experiments = _bind(C.CDLL(variant_path(EXPERIMENTS)), variant_path(EXPERIMENTS))
            assert _lib_experiments.nmpc_experiments_build() == 1
        return _lib_experiments
    if _lib is not None:
        return _lib
    path = os.environ.get("NMPC_LIB_PATH") or build_library()      # (NMPC_LIB_PATH: instrumented builds, scripts/ only)
    _lib = _bind(C.CDLL(path), path)
    return _lib


def _bind(lib: C.CDLL, path: str) -> C.CDLL:
    lib.nmpc_abi_version.restype = C.c_int
    if lib.nmpc_abi_version() != EXPECTED_ABI:      # a stale / foreign .so would silently misread nmpc_opts
        raise RuntimeError(f"{path}: ABI version {lib.nmpc_abi_version()}, this package expects {EXPECTED_ABI} "
                           "(rebuild: make -C mpc_trajectory_generator_amd/csrc -B)")
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    lib.nmpc_default_opts.argtypes = [C.POINTER(NmpcOpts)]
    lib.nmpc_default_opts.restype = None
    for f in ("nmpc_n_u", "nmpc_n_p", "nmpc_n1", "nmpc_n2"):
        getattr(lib, f).argtypes = [C.POINTER(NmpcProblem)]
        getattr(lib, f).restype = C.c_int
    lib.nmpc_new.argtypes = [C.POINTER(NmpcProblem), C.POINTER(NmpcOpts), C.c_int, C.c_int, C.POINTER(vp)]
    lib.nmpc_free.argtypes = [vp]
    lib.nmpc_free.restype = None
    lib.nmpc_ping.argtypes = [vp]
    lib.nmpc_last_error.argtypes = [vp]
    lib.nmpc_last_error.restype = C.c_char_p
    lib.nmpc_kernel_name.argtypes = [vp]
    lib.nmpc_kernel_name.restype = C.c_char_p
    lib.nmpc_last_batch_ms.argtypes = [vp]
    lib.nmpc_last_batch_ms.restype = C.c_double
    lib.nmpc_solve_batch_device.argtypes = [vp, C.c_int] + [vp] * 7
    lib.nmpc_solve_batch_host.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, vp]
    lib.nmpc_eval_batch_device.argtypes = [vp, C.c_int] + [vp] * 9
    lib.nmpc_eval_batch_host.argtypes = [vp, C.c_int] + [dp] * 8
    lib.nmpc_test_sincos_host.argtypes = [vp, C.c_int, dp, dp, dp]
    lib.nmpc_test_divsqrt_host.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    lib.nmpc_loop_new.argtypes = [vp, C.POINTER(NmpcRoute), C.c_int, dp, C.POINTER(C.c_int32), C.c_int, dp, C.c_int,
                                  C.POINTER(vp)]
    lib.nmpc_loop_free.argtypes = [vp]
    lib.nmpc_loop_free.restype = None
    lib.nmpc_loop_step.argtypes = [vp, vp]
    lib.nmpc_loop_read.argtypes = [vp, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), vp]
    lib.nmpc_loop_params.argtypes = [vp, dp, dp, dp]
    lib.nmpc_loop_trajectory.argtypes = [vp, dp, C.c_int]
    lib.nmpc_experiments_build.restype = C.c_int
    return lib


def as_dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
