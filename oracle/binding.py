"""ctypes binding of oracle/libnmpc_oracle.so (test infrastructure, NOT the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnmpc_oracle.so")


class OrcProblem(C.Structure):
    _fields_ = [("N", C.c_int32), ("nobs", C.c_int32), ("ndyn", C.c_int32), ("reserved", C.c_int32),
                ("ts", C.c_double), ("vmin", C.c_double), ("vmax", C.c_double), ("wmax", C.c_double),
                ("amin", C.c_double), ("amax", C.c_double), ("awmax", C.c_double)]


class OrcOpts(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("initial_tolerance", C.c_double),
                ("delta_tolerance", C.c_double), ("initial_penalty", C.c_double),
                ("penalty_update", C.c_double), ("tolerance_update", C.c_double),
                ("sufficient_decrease", C.c_double), ("lbfgs_memory", C.c_int32),
                ("max_inner", C.c_int32), ("max_outer", C.c_int32), ("max_total_inner", C.c_int32),
                ("akkt_gradient", C.c_int32), ("ls_failure", C.c_int32), ("inner_status", C.c_int32),
                ("reserved", C.c_int32)]


class OrcStatus(C.Structure):
    _fields_ = [("exit_status", C.c_int32), ("num_outer_iterations", C.c_uint32),
                ("num_inner_iterations", C.c_uint32), ("num_cost_evals", C.c_uint32),
                ("num_grad_evals", C.c_uint32), ("reserved", C.c_uint32),
                ("last_problem_norm_fpr", C.c_double), ("delta_y_norm_over_c", C.c_double),
                ("f2_norm", C.c_double), ("penalty", C.c_double), ("cost", C.c_double),
                ("solve_time_ms", C.c_double)]


STATUS_DTYPE = np.dtype([("exit_status", "<i4"), ("num_outer_iterations", "<u4"),
                         ("num_inner_iterations", "<u4"), ("num_cost_evals", "<u4"),
                         ("num_grad_evals", "<u4"), ("reserved", "<u4"),
                         ("last_problem_norm_fpr", "<f8"), ("delta_y_norm_over_c", "<f8"),
                         ("f2_norm", "<f8"), ("penalty", "<f8"), ("cost", "<f8"),
                         ("solve_time_ms", "<f8")])
assert STATUS_DTYPE.itemsize == C.sizeof(OrcStatus) == 72


def build_oracle(force=False):
    """Compile the oracle with gcc if the .so is missing or older than its sources."""
    srcs = [os.path.join(_HERE, f) for f in ("nmpc_oracle.c", "nmpc_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libnmpc_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


class Oracle:
    """Thin, array-in/array-out view of the oracle for one problem shape."""

    def __init__(self, N, nobs, ndyn, ts, vmin, vmax, wmax, amin, amax, awmax, **opts):
        self.lib = C.CDLL(build_oracle())
        self.pb = OrcProblem(N, nobs, ndyn, 0, ts, vmin, vmax, wmax, amin, amax, awmax)
        self.opts = OrcOpts()
        self.lib.orc_default_opts(C.byref(self.opts))
        for k, v in opts.items():
            if k == "reserved" or not hasattr(self.opts, k):
                raise TypeError(f"unknown oracle option {k!r}")
            setattr(self.opts, k, v)
        for f in ("orc_n_u", "orc_n_p", "orc_n1", "orc_n2"):
            getattr(self.lib, f).restype = C.c_int
        self.lib.orc_tree_sum.restype = C.c_double
        self.lib.orc_tree_sum.argtypes = [C.POINTER(C.c_double), C.c_int]
        self.lib.orc_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self.lib.orc_sincos_n.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 3
        self.lib.orc_eval.argtypes = [C.POINTER(OrcProblem), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.c_double] + [C.POINTER(C.c_double)] * 5
        self.lib.orc_solve_batch.argtypes = [C.POINTER(OrcProblem), C.POINTER(OrcOpts), C.c_int,
                                             C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.c_void_p, C.c_int]
        self.n_u = self.lib.orc_n_u(C.byref(self.pb))
        self.n_p = self.lib.orc_n_p(C.byref(self.pb))
        self.n1 = self.lib.orc_n1(C.byref(self.pb))
        self.n2 = self.lib.orc_n2(C.byref(self.pb))

    def sincos_array(self, x):
        """(sin, cos) of an array with the canonical sin/cos of the kernels (not libm's)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        sn, cs = np.empty_like(x), np.empty_like(x)
        dp = C.POINTER(C.c_double)
        self.lib.orc_sincos_n(x.size, x.ctypes.data_as(dp), sn.ctypes.data_as(dp), cs.ctypes.data_as(dp))
        return sn, cs

    def sincos(self, x):
        s, c = C.c_double(), C.c_double()
        self.lib.orc_sincos(float(x), C.byref(s), C.byref(c))
        return s.value, c.value

    def lbfgs_gram(self, m, s_list, y_list, r):
        """H r of the Gram-form L-BFGS after the pairs (s_list[k], y_list[k]), oldest first, entered an empty buffer of memory m."""
        s_list = np.ascontiguousarray(s_list, dtype=np.float64).reshape(-1, self.n_u)
        y_list = np.ascontiguousarray(y_list, dtype=np.float64).reshape(-1, self.n_u)
        d = np.array(r, dtype=np.float64, order="C")
        self.lib.orc_test_lbfgs_gram.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 3
        rc = self.lib.orc_test_lbfgs_gram(self.pb.N, int(m), s_list.shape[0], _dp(s_list), _dp(y_list), _dp(d))
        if rc:
            raise RuntimeError(f"orc_test_lbfgs_gram failed: {rc}")
        return d

    def tree_sum(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        return self.lib.orc_tree_sum(_dp(v), len(v))

    def eval(self, p, u, c=0.0, y=None, grad=True):
        """-> psi, grad, F1, F2 for one (p, u)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert p.shape == (self.n_p,) and u.shape == (self.n_u,)
        y = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
        psi = C.c_double()
        g = np.zeros(self.n_u) if grad else None
        F1, F2 = np.zeros(self.n1), np.zeros(self.n2)
        rc = self.lib.orc_eval(C.byref(self.pb), _dp(p), _dp(u), float(c), _dp(y), C.byref(psi),
                               _dp(g), _dp(F1), _dp(F2))
        if rc:
            raise RuntimeError(f"orc_eval failed: {rc}")
        return psi.value, g, F1, F2

    def solve_batch(self, p, u0=None, y0=None, c0=None, threads=1):
        """-> u[B,n_u], y[B,n1], status (structured array)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        B = p.shape[0]
        assert p.shape == (B, self.n_p)
        u = np.zeros((B, self.n_u)) if u0 is None else np.array(u0, dtype=np.float64, order="C")
        y0 = None if y0 is None else np.ascontiguousarray(y0, dtype=np.float64)
        c0 = None if c0 is None else np.ascontiguousarray(c0, dtype=np.float64)
        y = np.zeros((B, self.n1))
        st = np.zeros(B, dtype=STATUS_DTYPE)
        rc = self.lib.orc_solve_batch(C.byref(self.pb), C.byref(self.opts), B, _dp(p), _dp(u), _dp(y0),
                                      _dp(c0), _dp(y), st.ctypes.data, int(threads))
        if rc:
            raise RuntimeError(f"orc_solve_batch failed: {rc}")
        return u, y, st
