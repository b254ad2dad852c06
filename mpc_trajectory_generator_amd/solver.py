"""BatchSolver: the batched NMPC solve on one MI355X, through the C ABI of include/nmpc_solver.h.

The counterpart of what the reference obtains from ``MpcModule.build()`` +
``og.tcp.OptimizerTcpManager`` (src/mpc/mpc_generator.py:66-193, src/path_generator.py:218-222):
a solver for one problem shape that maps parameter vectors ``p`` to control horizons ``u*``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .config import Config


class SolverError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"MPC Solver error: {message} (code {code})")
        self.code, self.message = code, message


def problem_from_config(cfg: Config) -> _lib.NmpcProblem:
    """The quantities the reference bakes in at build() time (src/mpc/mpc_generator.py:70-71,151-168)."""
    return _lib.NmpcProblem(int(cfg.N_hor), int(cfg.Nobs), int(cfg.Ndynobs), 0, float(cfg.ts),
                            float(cfg.lin_vel_min), float(cfg.lin_vel_max), float(cfg.ang_vel_max),
                            float(cfg.lin_acc_min), float(cfg.lin_acc_max), float(cfg.ang_acc_max))


class BatchSolver:
    """One handle = one problem shape on one GPU.  Not thread-safe; distinct solvers are."""

    def __init__(self, cfg: Config, max_batch: int = 8192, device: int = 0, experiments: bool = False, **opts):
        self.cfg = cfg
        # raises if the HIP library cannot be had.  (experiments: tests / scripts only -- the variant that reads the NMPC_* environment knobs)
        self.lib = _lib.load_library(experiments=experiments)
        self.pb = problem_from_config(cfg)
        self.opts = _lib.NmpcOpts()
        self.lib.nmpc_default_opts(C.byref(self.opts))
        for k, v in opts.items():
            if not hasattr(self.opts, k):
                raise TypeError(f"unknown solver option {k!r}")
            setattr(self.opts, k, v)
        self.n_u = self.lib.nmpc_n_u(C.byref(self.pb))
        self.n_p = self.lib.nmpc_n_p(C.byref(self.pb))
        self.n1 = self.lib.nmpc_n1(C.byref(self.pb))
        self.n2 = self.lib.nmpc_n2(C.byref(self.pb))
        assert (self.n_u, self.n_p) == (cfg.n_u, cfg.n_p)
        self.max_batch, self.device = int(max_batch), int(device)
        h = C.c_void_p()
        rc = self.lib.nmpc_new(C.byref(self.pb), C.byref(self.opts), self.device, self.max_batch, C.byref(h))
        if rc != 0:
            raise SolverError(rc, f"nmpc_new failed: {_lib.ERRORS.get(rc, rc)} "
                                  "(this package needs a HIP device; there is no CPU fallback)")
        self._h = h

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            self.lib.nmpc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def variant(self) -> dict:
        """The restatement switches in force, by name, plus the iteration budget."""
        v = {k: names[getattr(self.opts, k)] for k, names in _lib.VARIANT_FIELDS.items()}
        v["max_total_inner"] = int(self.opts.max_total_inner)
        return v

    def oracle_opts(self) -> dict:
        """This handle's options as keyword arguments of the test oracle (field names are shared)."""
        return {name: getattr(self.opts, name) for name, _ in self.opts._fields_ if name != "reserved"}

    @property
    def kernel_name(self) -> str:
        """Name of the solve kernel this handle launches (diagnostic; what a kernel trace shows)."""
        return self.lib.nmpc_kernel_name(self._h).decode()

    @property
    def last_batch_ms(self) -> float:
        """Kernel time (HIP events) of the last host-path ``solve`` on this handle; per-instance times are in
        ``status["solve_time_ms"]``."""
        return float(self.lib.nmpc_last_batch_ms(self._h))

    def ping(self):
        self._check(self.lib.nmpc_ping(self._h))

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.nmpc_last_error(self._h) if self._h else b""
            raise SolverError(rc, f"{_lib.ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    # ------------------------------------------------------------------ host-buffer path
    def solve(self, p, u0=None, y0=None, c0=None):
        """p [B, n_p] -> (u [B, n_u], y [B, n1], status structured array); numpy in, numpy out."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        B = p.shape[0]
        if p.ndim != 2 or p.shape[1] != self.n_p:
            raise SolverError(3003, f"wrong number of parameters: got {p.shape}, expected [B, {self.n_p}]")
        u = np.zeros((B, self.n_u)) if u0 is None else np.array(u0, dtype=np.float64, order="C")
        if u.shape != (B, self.n_u):
            raise SolverError(1600, "initial guess has incompatible dimensions")
        y0 = None if y0 is None else np.ascontiguousarray(y0, dtype=np.float64)
        if y0 is not None and y0.shape != (B, self.n1):
            raise SolverError(1700, "wrong dimension of Lagrange multipliers")
        c0 = None if c0 is None else np.ascontiguousarray(c0, dtype=np.float64).reshape(B)
        y = np.zeros((B, self.n1))
        st = np.zeros(B, dtype=_lib.STATUS_DTYPE)
        self._check(self.lib.nmpc_solve_batch_host(self._h, B, _lib.as_dp(p), _lib.as_dp(u), _lib.as_dp(y0),
                                                   _lib.as_dp(c0), _lib.as_dp(y), st.ctypes.data))
        return u, y, st

    def evaluate(self, p, u, c=None, y=None):
        """psi, grad psi, F1, F2 at (u; c, y, p) for a batch; c=None, y=None gives f and grad f."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        B = p.shape[0]
        assert p.shape == (B, self.n_p) and u.shape == (B, self.n_u)
        c = None if c is None else np.ascontiguousarray(c, dtype=np.float64).reshape(B)
        y = None if y is None else np.ascontiguousarray(y, dtype=np.float64).reshape(B, self.n1)
        psi, g = np.zeros(B), np.zeros((B, self.n_u))
        F1, F2 = np.zeros((B, self.n1)), np.zeros((B, max(self.n2, 1)))
        self._check(self.lib.nmpc_eval_batch_host(self._h, B, _lib.as_dp(p), _lib.as_dp(u), _lib.as_dp(c),
                                                  _lib.as_dp(y), _lib.as_dp(psi), _lib.as_dp(g),
                                                  _lib.as_dp(F1), _lib.as_dp(F2)))
        return psi, g, F1, F2[:, :self.n2]

    # ------------------------------------------------------------------ device-resident path
    def solve_device(self, d_p, d_u, d_y0=None, d_c0=None, d_y_out=None, d_status=None, stream=None):
        """Operands are torch CUDA tensors (float64, contiguous; status uint8 [B, 72]) already in
        HBM; enqueues on ``stream`` (default: torch's current stream) and returns immediately."""
        import torch
        B = d_p.shape[0]
        for tns, cols in ((d_p, self.n_p), (d_u, self.n_u)):
            assert tns.is_cuda and tns.dtype == torch.float64 and tns.is_contiguous() and tns.shape == (B, cols)
        s = torch.cuda.current_stream(d_p.device) if stream is None else stream
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None  # noqa: E731
        self._check(self.lib.nmpc_solve_batch_device(self._h, B, ptr(d_p), ptr(d_u), ptr(d_y0), ptr(d_c0),
                                                     ptr(d_y_out), ptr(d_status), C.c_void_p(s.cuda_stream)))

    # ------------------------------------------------------------------ primitives (tests)
    def test_sincos(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        s, c = np.zeros_like(x), np.zeros_like(x)
        self._check(self.lib.nmpc_test_sincos_host(self._h, x.size, _lib.as_dp(x), _lib.as_dp(s), _lib.as_dp(c)))
        return s, c

    def test_divsqrt(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        q, r = np.zeros_like(a), np.zeros_like(a)
        self._check(self.lib.nmpc_test_divsqrt_host(self._h, a.size, _lib.as_dp(a), _lib.as_dp(b),
                                                    _lib.as_dp(q), _lib.as_dp(r)))
        return q, r


def status_from_bytes(t):
    """torch uint8 [B, 72] (device or host) -> numpy structured array."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=_lib.STATUS_DTYPE)
