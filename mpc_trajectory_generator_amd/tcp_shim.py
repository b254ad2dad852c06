"""OptimizerTcpManager-shaped handle over the HIP solver.

The reference talks to its OpEn-generated solver through ``og.tcp.OptimizerTcpManager``
(src/path_generator.py:218-222,408,417 ; src/mpc/mpc_generator.py:206-221): ``start()``,
``ping()``, ``call(parameters)`` -> response with ``is_ok()`` / ``get()``, ``kill()``.  This class has
the same surface and the same sequential semantics, but the "server" is an
in-process C-ABI handle on the GPU -- no socket, no JSON:

* the solution buffer ``u`` lives with the manager, starts at zero and is left untouched between
  calls when no ``initial_guess`` is given, so consecutive calls warm-start from the previous
  solution, unshifted, exactly like the reference's driver experiences;
* the penalty restarts at its initial value on every call unless ``initial_penalty`` is given;
* the Lagrange multipliers persist between calls unless ``initial_y`` is given
  (``keep_multipliers=False`` resets them to zero per call instead; which of the two OpEn does is
  not verifiable here);
* a wrong parameter count / guess size / multiplier size yields an error response with OpEn's
  codes 3003 / 1600 / 1700, a solver failure 2000; non-convergence is NOT an error.

``call_batch`` is the batched path: B parameter vectors per call.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .config import Config, load_config
from .solver import BatchSolver, SolverError as _SolverError


class SolverStatus:
    """Success payload, attribute-compatible with opengen's ``SolverStatus``."""

    def __init__(self, u, y, st):
        self.exit_status = _lib.EXIT_STATUS[int(st["exit_status"])]
        self.num_outer_iterations = int(st["num_outer_iterations"])
        self.num_inner_iterations = int(st["num_inner_iterations"])
        self.last_problem_norm_fpr = float(st["last_problem_norm_fpr"])
        self.f1_infeasibility = float(st["delta_y_norm_over_c"])
        self.delta_y_norm_over_c = self.f1_infeasibility          # name used by opengen 0.6.x
        self.f2_norm = float(st["f2_norm"])
        self.solve_time_ms = float(st["solve_time_ms"])
        self.penalty = float(st["penalty"])
        self.cost = float(st["cost"])
        self.solution = [float(v) for v in u]
        self.lagrange_multipliers = [float(v) for v in y]


class SolverError:
    """Error payload, attribute-compatible with opengen's ``SolverError``."""

    def __init__(self, code, message):
        self.code, self.message = int(code), str(message)


class SolverResponse:
    def __init__(self, payload):
        self._payload = payload

    def is_ok(self):
        return isinstance(self._payload, SolverStatus)

    def get(self):
        return self._payload

    def __getitem__(self, key):                      # opengen also allows response["solution"]
        return getattr(self._payload, key)


class OptimizerTcpManager:
    def __init__(self, optimizer_path=None, config: Config | None = None, device: int = 0,
                 max_batch: int = 8192, keep_multipliers: bool = True, **solver_opts):
        self.optimizer_path = optimizer_path          # accepted for call-site compatibility, unused
        self.cfg = config if config is not None else load_config()
        self._device, self._max_batch = device, max_batch
        self._opts = solver_opts
        self._keep_y = keep_multipliers
        self._solver: BatchSolver | None = None
        self._u = self._y = None

    # -- lifecycle ------------------------------------------------------------------------
    def start(self):
        if self._solver is not None:
            raise RuntimeError("optimizer already started")
        self._solver = BatchSolver(self.cfg, max_batch=self._max_batch, device=self._device, **self._opts)
        self._u = np.zeros((1, self._solver.n_u))     # the server's solution buffer starts at zero
        self._y = np.zeros((1, self._solver.n1))

    def _need(self):
        if self._solver is None:
            raise ConnectionRefusedError("optimizer is not running (start() not called, or killed)")
        return self._solver

    def ping(self):
        self._need().ping()
        return {"Pong": 1}

    def kill(self):
        if self._solver is not None:
            self._solver.close()
            self._solver = None

    # -- one solve ------------------------------------------------------------------------
    def call(self, p, initial_guess=None, initial_y=None, initial_penalty=None,
             buffer_len=4096, max_data_size=1048576):
        s = self._need()
        p = np.asarray(p, dtype=np.float64).reshape(-1)
        if p.size != s.n_p:
            return SolverResponse(SolverError(3003, f"wrong number of parameters: provided {p.size}, expected {s.n_p}"))
        # validate everything first: an error response must leave the server's warm-start state untouched
        g = yy = None
        if initial_guess is not None:
            g = np.asarray(initial_guess, dtype=np.float64).reshape(-1)
            if g.size != s.n_u:
                return SolverResponse(SolverError(1600, f"initial guess has incompatible dimensions: provided {g.size}, expected {s.n_u}"))
        if initial_y is not None:
            yy = np.asarray(initial_y, dtype=np.float64).reshape(-1)
            if yy.size != s.n1:
                return SolverResponse(SolverError(1700, f"wrong dimension of Lagrange multipliers: provided {yy.size}, expected {s.n1}"))
        u_in, y_in = self._u.copy(), self._y.copy()
        if g is not None:
            u_in[0] = g
        if yy is not None:
            y_in[0] = yy
        elif not self._keep_y:
            y_in[0] = 0.0
        c0 = None if initial_penalty is None else np.array([float(initial_penalty)])
        try:
            u, y, st = s.solve(p[None, :], u0=u_in, y0=y_in, c0=c0)
        except _SolverError as e:
            return SolverResponse(SolverError(2000, f"problem solution failed: {e.message}"))
        if int(st["exit_status"][0]) == 4:
            return SolverResponse(SolverError(2000, "problem solution failed: NotFiniteComputation"))
        self._u, self._y = u, y
        return SolverResponse(SolverStatus(u[0], y[0], st[0]))

    # -- many solves ----------------------------------------------------------------------
    def call_batch(self, P, u0=None, y0=None, c0=None):
        """P [B, n_p] -> (U [B, n_u], Y [B, n1], status structured array).  Stateless: the initial
        guess / multipliers are what the caller passes (zeros when omitted)."""
        return self._need().solve(P, u0=u0, y0=y0, c0=c0)
