"""Receding-horizon drivers on top of the solver handle.

``TrajectoryGenerator.run`` follows the reference's ``PathGenerator.run`` loop
(src/path_generator.py:197-437) call for call -- manager start / ping, per-step parameter
assembly, ``mpc_step`` (= ``MpcModule.run``, src/mpc/mpc_generator.py:204-237), terminal test, kill
-- so a user of the reference finds the same control flow, with the OpEn TCP manager replaced by
``tcp_shim.OptimizerTcpManager``.  The visibility-graph A* front-end (extremitypathfinder /
pyclipper) is outside this project's scope; a ``harness.Route`` (waypoints + NMPC vertices) is
what ``run`` starts from.

``BatchedRecedingHorizon`` is the batched counterpart for BASELINE config 4: B independent robots
advance in lock step, one batched solve per step, controls and multipliers carried as warm starts.
"""
from __future__ import annotations

import math
import time

import numpy as np

from . import harness
from .config import Config
from .tcp_shim import OptimizerTcpManager


def mpc_step(cfg: Config, parameters, mng, take_steps, system_input, states):
    """One NMPC solve + state advance: src/mpc/mpc_generator.py:204-237."""
    solution = mng.call(parameters)                                          # :206
    if solution.is_ok():                                                     # :209-214
        data = solution.get()
        u, exit_status, solver_time = data.solution, data.exit_status, data.solve_time_ms
    else:                                                                    # :215-221
        err = solution.get()
        mng.kill()
        raise RuntimeError(f"MPC Solver error: {err.message}")
    system_input += u[:cfg.nu * take_steps]                                  # :223
    for i in range(take_steps):                                              # :225-235, Euler diff-drive
        u_v, u_omega = u[i * cfg.nu], u[1 + i * cfg.nu]
        x, y, theta = states[-3], states[-2], states[-1]
        states += [x + cfg.ts * (u_v * math.cos(theta)), y + cfg.ts * (u_v * math.sin(theta)),
                   theta + cfg.ts * u_omega]
    return exit_status, solver_time


class TrajectoryGenerator:
    """Counterpart of the reference's ``PathGenerator`` (plots and reports omitted)."""

    def __init__(self, config: Config, build: bool = False, verbose: bool = False, sinus_object: bool = False,
                 manager_factory=None):
        self.config, self.verbose, self.sinus_object = config, verbose, sinus_object
        self.time_dict, self.solver_times, self.overhead_times = {}, [], []
        self._factory = manager_factory or (lambda: OptimizerTcpManager(
            config.build_directory + "/" + config.optimizer_name, config=config))
        # build=True triggers OpEn code generation in the reference (:33-34); here the kernels are
        # compiled when the library is first loaded, nothing to do.

    def run(self, route: harness.Route, max_steps: int | None = None, record_parameters: list | None = None):
        """-> (xx, xy, uv, uomega, solver_times, overhead_times), src/path_generator.py:197-437."""
        cfg = self.config
        t_temp = time.time()
        mng = self._factory()                                                 # :218-222
        mng.start()
        mng.ping()
        self.time_dict["opt_launch"] = int(1000 * (time.time() - t_temp))
        tt = time.time()
        start, end = list(route.start), list(route.end)
        x_ref, y_ref = route.x_ref, route.y_ref
        terminal, t, idx = False, 0, 0
        self.solver_times, self.overhead_times = [], []
        system_input = []
        states = list(map(float, start))                                      # :267
        constraints = [0.0] * cfg.Nobs * cfg.nobs                             # :273
        dyn_constraints = harness.initial_dyn_constraints(cfg)                # :274-280
        params_per_dyn_obs = cfg.N_hor * cfg.ndynobs
        limit = 500.0 / cfg.ts if max_steps is None else max_steps
        t_temp = time.time()
        try:
            while (not terminal) and t < limit:                                   # :290
                t_overhead = time.time()
                x_init = states[-cfg.nx:]                                         # :293
                if len(route.vertices):                                           # :295-304
                    constraints = harness.static_constraints(route, (x_init[0], x_init[1]))
                if t == 0:                                                        # :306-309
                    for i, obs in enumerate(harness.dyn_obstacle(cfg, route.dyn_obs_list, t * cfg.ts, cfg.N_hor,
                                                                 self.sinus_object)):
                        dyn_constraints[i * params_per_dyn_obs:(i + 1) * params_per_dyn_obs] = \
                            [float(v) for tup in obs for v in tup]
                else:                                                             # :310-316 rotate left, refresh the tail
                    k = cfg.ndynobs * cfg.num_steps_taken
                    dyn_constraints = dyn_constraints[k:] + dyn_constraints[:k]
                    for i, obs in enumerate(harness.dyn_obstacle(cfg, route.dyn_obs_list,
                                                                 (t + cfg.N_hor - cfg.num_steps_taken) * cfg.ts,
                                                                 cfg.num_steps_taken, self.sinus_object)):
                        dyn_constraints[(i + 1) * params_per_dyn_obs - k:(i + 1) * params_per_dyn_obs] = \
                            [float(v) for tup in obs for v in tup]
                lb_idx = max(0, idx - 1 * cfg.num_steps_taken)                    # :320-325
                ub_idx = min(len(x_ref), idx + 5 * cfg.num_steps_taken)
                idx = harness.closest_index((x_init[0], x_init[1]), route.ref_points[lb_idx:ub_idx]) + lb_idx
                last_u = system_input[-cfg.nu:] if len(system_input) else [0.0] * cfg.nu     # :371-374
                parameters = harness.assemble_params(route, x_init, last_u, idx, constraints, dyn_constraints)
                if record_parameters is not None:
                    record_parameters.append(list(parameters))
                try:                                                              # :384-391
                    exit_status, solver_time = mpc_step(cfg, parameters, mng, cfg.num_steps_taken, system_input, states)
                    self.solver_times.append(solver_time)
                except RuntimeError as err:
                    if self.verbose:
                        print(err)
                    return None
                if exit_status in cfg.bad_exit_codes and self.verbose:            # :393-394
                    print(f"[MPC] Bad converge status: {exit_status}")
                if np.allclose(states[-3:-1], end[0:2], atol=0.05, rtol=0) and abs(system_input[-2]) < 0.005:   # :397
                    terminal = True
                t += cfg.num_steps_taken
                self.overhead_times.append((time.time() - t_overhead) * 1000.0 - solver_time)
        except KeyboardInterrupt:                                             # :405-415: kill the server, return what was driven so far
            if self.verbose:
                print("[MPC] killing TCP connection to MCP solver...")
            mng.kill()
            nx = cfg.nx
            return (states[0::nx], states[1::nx], system_input[0::2], system_input[1::2],
                    self.solver_times, self.overhead_times)
        mng.kill()                                                            # :417
        self.time_dict["mpc_time"] = int(1000 * (time.time() - t_temp))
        self.time_dict["solver_time"] = sum(self.solver_times)
        self.time_dict["mean_solver_time"] = float(np.mean(self.solver_times)) if self.solver_times else 0.0
        self.time_dict["total_time"] = int(1000 * (time.time() - tt))
        nx = cfg.nx
        return (states[0::nx], states[1::nx], system_input[0::2], system_input[1::2],
                self.solver_times, self.overhead_times)


class BatchedRecedingHorizon:
    """B robots on one route, advanced in lock step with one batched solve per step.

    Per robot and step the parameter vector is filled as ``TrajectoryGenerator.run`` does
    (closest reference sample in the sliding window, horizon padded with the end pose, braking
    ``vel_ref``, dynamic block rotated left and refreshed).  ``solve_fn(P, u0, y0) -> (U, Y, status)``
    is the batched solver (``BatchSolver.solve``); controls and multipliers are carried over as
    warm starts, the penalty restarts at its initial value, like the sequential path.
    """

    def __init__(self, route: harness.Route, starts, dyn_obs_lists=None, sinus_object=False):
        self.route, self.cfg = route, route.cfg
        self.B = len(starts)
        self.states = [list(map(float, s)) for s in starts]       # per robot: flat [x, y, theta, ...]
        self.inputs = [[] for _ in range(self.B)]
        self.idx = [0] * self.B
        self.t = 0
        self.dyn_lists = dyn_obs_lists if dyn_obs_lists is not None else [route.dyn_obs_list] * self.B
        self.sinus_object = sinus_object
        self.dyn = [harness.initial_dyn_constraints(self.cfg) for _ in range(self.B)]
        self.constraints = [[0.0] * self.cfg.Nobs * self.cfg.nobs for _ in range(self.B)]
        self.U = np.zeros((self.B, self.cfg.n_u))
        self.Y = np.zeros((self.B, self.cfg.n1))
        self.done = np.zeros(self.B, dtype=bool)

    def assemble(self):
        cfg, route = self.cfg, self.route
        per = cfg.N_hor * cfg.ndynobs
        k = cfg.ndynobs * cfg.num_steps_taken
        P = np.empty((self.B, cfg.n_p))
        for b in range(self.B):
            x_init = self.states[b][-cfg.nx:]
            if len(route.vertices):
                self.constraints[b] = harness.static_constraints(route, (x_init[0], x_init[1]))
            if self.t == 0:
                preds = harness.dyn_obstacle(cfg, self.dyn_lists[b], 0.0, cfg.N_hor, self.sinus_object)
                for i, obs in enumerate(preds):
                    self.dyn[b][i * per:(i + 1) * per] = [float(v) for tup in obs for v in tup]
            else:
                self.dyn[b] = self.dyn[b][k:] + self.dyn[b][:k]
                preds = harness.dyn_obstacle(cfg, self.dyn_lists[b], (self.t + cfg.N_hor - cfg.num_steps_taken) * cfg.ts,
                                             cfg.num_steps_taken, self.sinus_object)
                for i, obs in enumerate(preds):
                    self.dyn[b][(i + 1) * per - k:(i + 1) * per] = [float(v) for tup in obs for v in tup]
            lb = max(0, self.idx[b] - cfg.num_steps_taken)
            ub = min(len(route.x_ref), self.idx[b] + 5 * cfg.num_steps_taken)
            self.idx[b] = harness.closest_index((x_init[0], x_init[1]), route.ref_points[lb:ub]) + lb
            last_u = self.inputs[b][-cfg.nu:] if self.inputs[b] else [0.0] * cfg.nu
            P[b] = harness.assemble_params(route, x_init, last_u, self.idx[b], self.constraints[b], self.dyn[b])
        return P

    def advance(self, U):
        cfg = self.cfg
        for b in range(self.B):
            u = U[b]
            self.inputs[b] += [float(v) for v in u[:cfg.nu * cfg.num_steps_taken]]
            st = self.states[b]
            for i in range(cfg.num_steps_taken):
                x, y, th = st[-3], st[-2], st[-1]
                st += [x + cfg.ts * (u[i * cfg.nu] * math.cos(th)), y + cfg.ts * (u[i * cfg.nu] * math.sin(th)),
                       th + cfg.ts * u[1 + i * cfg.nu]]
            end = self.route.end
            self.done[b] = (abs(st[-3] - end[0]) <= 0.05 and abs(st[-2] - end[1]) <= 0.05
                            and abs(self.inputs[b][-2]) < 0.005)
        self.t += cfg.num_steps_taken

    def step(self, solve_fn):
        P = self.assemble()
        U, Y, st = solve_fn(P, self.U, self.Y)
        self.U, self.Y = U, Y
        self.advance(U)
        return P, st


class VectorizedRecedingHorizon:
    """``BatchedRecedingHorizon`` with the per-robot Python loops replaced by NumPy array operations
    (BASELINE config 4: 8192 robots x 100 steps).  Same quantities, same order of operations per
    robot -- the test suite demands bit-identical parameter vectors against the loop version.

    All robots share the route; dynamic obstacles are per robot: ``dyn_obs`` is ``None`` or a tuple of
    arrays ``(p1 [B, K, 2], p2 [B, K, 2], freq [B, K], rx [B, K], ry [B, K], angle [B, K])``.
    """

    def __init__(self, route: harness.Route, starts, dyn_obs=None, sincos=None, sinus_object=False):
        cfg = self.cfg = route.cfg
        self.route = route
        self.sinus_object = bool(sinus_object)     # obstacle index 2 follows the sinusoidal law (visibility.py:183-196,210-212)
        # sin / cos used by the state advance and the obstacle predictor: libm's (as the reference) unless
        # a replacement is given -- the device loop's bit-level mirror passes the kernels' own sin / cos
        self.sincos = sincos if sincos is not None else (lambda x: (np.sin(x), np.cos(x)))
        self.B = B = len(starts)
        self.state = np.array(starts, dtype=np.float64).reshape(B, 3)
        self.traj = [self.state.copy()]
        self.last_u = np.zeros((B, cfg.nu))
        self.has_input = False
        self.idx = np.zeros(B, dtype=np.int64)
        self.t = 0
        self.dyn_obs = dyn_obs
        K = 0 if dyn_obs is None else dyn_obs[0].shape[1]
        assert K <= cfg.Ndynobs
        self.K = K
        d = np.zeros((B, cfg.Ndynobs, cfg.N_hor, cfg.ndynobs))
        d[..., 2] = 1.0
        d[..., 3] = 1.0                                           # padding: unit radii (path_generator.py:274-280)
        self.dyn = d
        self.U = np.zeros((B, cfg.n_u))
        self.Y = np.zeros((B, cfg.n1))
        self.x_ref = np.array(route.x_ref)
        self.y_ref = np.array(route.y_ref)
        self.th_ref = np.array(route.theta_ref)
        self.n = len(self.x_ref)
        self.vert = np.array(route.vertices, dtype=np.float64).reshape(-1, 2)
        self.done = np.zeros(B, dtype=bool)

    # dynamic-obstacle prediction for all robots: visibility.py:156-166,199-216 (linear law)
    def _predict(self, t0, horizon):
        cfg = self.cfg
        p1, p2, freq, rx, ry, ang = self.dyn_obs
        times = np.linspace(t0, t0 + horizon * cfg.ts, horizon)                       # (:204)
        s = np.abs(self.sincos(freq[:, :, None] * times[None, None, :])[0])        # [B, K, H]
        pos = s[..., None] * p1[:, :, None, :] + (1 - s[..., None]) * p2[:, :, None, :]
        if self.sinus_object and pos.shape[1] > 2:                                   # (:183-196), amplitude 1.5
            k = 2
            ang_d = np.arctan2(p2[:, k, 1] - p1[:, k, 1], p2[:, k, 0] - p1[:, k, 0])[:, None]       # [B, 1]
            add = 1.5 * self.sincos((10 * freq[:, k, None]) * times[None, :])[1]                      # [B, H]
            sa, ca = self.sincos(ang_d)
            dx, dy = pos[:, k, :, 0] - p1[:, k, None, 0], pos[:, k, :, 1] - p1[:, k, None, 1]
            ex = ca * dx - sa * dy
            ey = sa * dx + ca * dy
            ey = ey + add
            sm, cm = self.sincos(-ang_d)
            qx = cm * (ex - 0.0) - sm * (ey - 0.0)
            qy = sm * (ex - 0.0) + cm * (ey - 0.0)
            pos[:, k, :, 0] = qx + p1[:, k, None, 0]
            pos[:, k, :, 1] = qy + p1[:, k, None, 1]
        pad = cfg.vehicle_width / 2 + cfg.vehicle_margin
        out = np.empty(pos.shape[:3] + (5,))
        out[..., 0:2] = pos
        out[..., 2] = (rx + pad)[:, :, None]
        out[..., 3] = (ry + pad)[:, :, None]
        out[..., 4] = ang[:, :, None]
        return out

    def assemble(self):
        cfg, route, B, N, n = self.cfg, self.route, self.B, self.cfg.N_hor, self.n
        s = cfg.num_steps_taken
        x, y = self.state[:, 0], self.state[:, 1]
        # static circles (path_generator.py:295-304 + visibility.py:141-148 with look-back 0)
        cons = np.zeros((B, cfg.Nobs, cfg.nobs))
        nv = len(self.vert)
        if nv:
            if cfg.Nobs >= nv:
                cons[:, :nv, 0:2] = self.vert[None]
                cons[:, :nv, 2] = route.radius
            else:
                dist = np.linalg.norm(self.vert[None, :, :] - self.state[:, None, 0:2], axis=2)
                lb = np.argmin(dist, axis=1)
                ub = min(nv, cfg.Nobs)
                j = lb[:, None] + np.arange(cfg.Nobs)[None, :]
                ok = j < ub
                jj = np.minimum(j, nv - 1)
                cons[..., 0:2] = np.where(ok[..., None], self.vert[jj], 0.0)
                cons[..., 2] = np.where(ok, route.radius, 0.0)
        # dynamic ellipses (path_generator.py:306-316)
        if self.K:
            if self.t == 0:
                self.dyn[:, :self.K] = self._predict(0.0, N)
            else:
                # the reference rotates the WHOLE flat list left by ndynobs * s entries (:312): inside a block that is a
                # shift by s stages, and a block's last s stages take the next block's first s (the last block's take
                # block 0's -- a padding slot can so inherit stale ellipses of obstacle 0 when 0 < K < Ndynobs)
                flat = self.dyn.reshape(B, -1)
                self.dyn = np.roll(flat, -cfg.ndynobs * s, axis=1).reshape(self.dyn.shape)
                self.dyn[:, :self.K, N - s:] = self._predict((self.t + N - s) * cfg.ts, s)
        # closest reference sample in the sliding window (:320-325)
        lb = np.maximum(0, self.idx - s)
        ub = np.minimum(n, self.idx + 5 * s)
        w = np.arange(6 * s)
        j = lb[:, None] + w[None, :]
        ok = j < ub[:, None]
        jj = np.minimum(j, n - 1)
        d = np.linalg.norm(np.stack([self.x_ref[jj] - x[:, None], self.y_ref[jj] - y[:, None]], axis=2), axis=2)
        d = np.where(ok, d, np.inf)
        self.idx = lb + np.argmin(d, axis=1)
        idx = self.idx
        # horizon references and target (:326-341)
        end = np.array(route.end, dtype=np.float64)
        j = idx[:, None] + np.arange(N)[None, :]
        ok = j < n
        jj = np.minimum(j, n - 1)
        refs = np.empty((B, N, 3))
        refs[..., 0] = np.where(ok, self.x_ref[jj], end[0])
        refs[..., 1] = np.where(ok, self.y_ref[jj], end[1])
        refs[..., 2] = np.where(ok, self.th_ref[jj], end[2])
        far = idx + N < n
        jf = np.minimum(idx + N, n - 1)
        xf = np.where(far[:, None], np.stack([self.x_ref[jf], self.y_ref[jf], self.th_ref[jf]], axis=1), end[None, :])
        # velocity reference with the braking profile (:343-361)
        bv, bd, base = np.array(route.brake_velocities), np.array(route.brake_distances), route.base_speed
        vel = np.full((B, N), base)
        brake = (idx + N) >= n - bd[0] / base
        num_base = np.minimum(n - idx - 1, N)
        k = np.arange(N)[None, :]
        nb = num_base[:, None]
        tail = np.where(k - nb < len(bv), bv[np.clip(k - nb, 0, len(bv) - 1)], 0.0)
        vel_b = np.where(k < nb, base, tail)
        vel = np.where(brake[:, None], vel_b, vel)
        for b in np.where(brake & (num_base == 0))[0]:                      # inside the last sample: distance-based (:347-351)
            dist_to_goal = math.sqrt((self.state[b, 0] - end[0]) ** 2 + (self.state[b, 1] - end[1]) ** 2)
            vr = [v for (v, dd) in zip(bv, bd) if dd <= dist_to_goal][:N]
            vel[b] = np.array(vr + [0.0] * (N - len(vr)))
        W = np.tile(np.array(cfg.weights()), (B, 1))
        P = np.concatenate([self.state, self.last_u, xf, self.last_u, W, vel, cons.reshape(B, -1),
                            self.dyn.reshape(B, -1), refs.reshape(B, -1)], axis=1)
        assert P.shape[1] == cfg.n_p
        return P

    def advance(self, U):
        cfg = self.cfg
        s = cfg.num_steps_taken
        st = self.state.copy()
        for i in range(s):                                                  # mpc_generator.py:225-235
            v, w = U[:, i * cfg.nu], U[:, 1 + i * cfg.nu]
            th = st[:, 2]
            # per robot: x + ts*(v*cos(theta)) with math.cos -> np.cos is the same libm call
            sn, cs = self.sincos(np.ascontiguousarray(th))
            st = np.stack([st[:, 0] + cfg.ts * (v * cs), st[:, 1] + cfg.ts * (v * sn),
                           th + cfg.ts * w], axis=1)
            self.traj.append(st.copy())
        self.state = st
        self.last_u = U[:, (s - 1) * cfg.nu:s * cfg.nu].copy()
        self.has_input = True
        end = self.route.end
        self.done = (np.abs(st[:, 0] - end[0]) <= 0.05) & (np.abs(st[:, 1] - end[1]) <= 0.05) & (np.abs(self.last_u[:, 0]) < 0.005)
        self.t += s

    def step(self, solve_fn):
        P = self.assemble()
        U, Y, st = solve_fn(P, self.U, self.Y)
        self.U, self.Y = U, Y
        self.advance(U)
        return P, st


class DeviceRecedingHorizon:
    """``VectorizedRecedingHorizon`` with everything on the GPU: parameter assembly, the batched solve
    and the state advance are kernels of libnmpc_hip.so (``nmpc_loop_*``, include/nmpc_solver.h), and
    nothing crosses PCIe between steps.  Same quantities, same order of operations as the host class;
    sin / cos are the kernels' own (tests/test_gpu_loop.py compares bit for bit against the host class
    given the same sin / cos).

    ``solver`` is the ``BatchSolver`` whose handle runs the solves; ``dyn_obs`` as in
    ``VectorizedRecedingHorizon``; ``max_steps`` > 0 records the trajectory on device; ``idx0`` = the
    reference sample each robot starts at (default 0, as the reference).
    """

    def __init__(self, solver, route: harness.Route, starts, dyn_obs=None, max_steps: int = 0, idx0=None,
                 sinus_object=False):
        import ctypes as C
        from . import _lib
        cfg = self.cfg = route.cfg
        self.solver, self.route, self.lib = solver, route, solver.lib
        self.B = B = len(starts)
        self.t = 0
        self.steps = 0
        starts = np.ascontiguousarray(np.array(starts, dtype=np.float64).reshape(B, 3))
        K = 0 if dyn_obs is None else dyn_obs[0].shape[1]
        dyn = None
        if K:
            p1, p2, freq, rx, ry, ang = dyn_obs
            sinus = np.zeros((B, K))
            if sinus_object and K > 2:
                sinus[:, 2] = 1.0                  # obstacle index 2 follows the sinusoidal law (visibility.py:210-212)
            direction = np.arctan2(p2[..., 1] - p1[..., 1], p2[..., 0] - p1[..., 0])
            dyn = np.ascontiguousarray(np.concatenate(
                [p1, p2, freq[..., None], rx[..., None], ry[..., None], ang[..., None], sinus[..., None],
                 direction[..., None]], axis=2), dtype=np.float64)
            assert dyn.shape == (B, K, 10)
        r = _lib.NmpcRoute()
        keep = []                                              # arrays the struct points to, until nmpc_loop_new returns

        def arr(v):
            a = np.ascontiguousarray(v, dtype=np.float64)
            keep.append(a)
            return _lib.as_dp(a)
        vert = np.array(route.vertices, dtype=np.float64).reshape(-1, 2)
        r.n_ref, r.n_vert, r.n_brake = len(route.x_ref), len(vert), len(route.brake_velocities)
        r.num_steps_taken = cfg.num_steps_taken
        r.x_ref, r.y_ref, r.theta_ref = arr(route.x_ref), arr(route.y_ref), arr(route.theta_ref)
        r.vert_xy = arr(vert) if len(vert) else None
        r.brake_vel, r.brake_dist = arr(route.brake_velocities), arr(route.brake_distances)
        r.end = (C.c_double * 3)(*[float(v) for v in route.end])
        r.base_speed, r.radius = float(route.base_speed), float(route.radius)
        r.dyn_pad = cfg.vehicle_width / 2 + cfg.vehicle_margin
        r.weights = (C.c_double * 10)(*cfg.weights())
        h = C.c_void_p()
        i0 = None if idx0 is None else np.ascontiguousarray(idx0, dtype=np.int32)
        solver._check(self.lib.nmpc_loop_new(solver._h, C.byref(r), B, _lib.as_dp(starts),
                                             None if i0 is None else i0.ctypes.data_as(C.POINTER(C.c_int32)),
                                             K, _lib.as_dp(dyn), int(max_steps), C.byref(h)))
        self._l = h
        self.max_steps = int(max_steps)

    def close(self):
        if getattr(self, "_l", None):
            self.lib.nmpc_loop_free(self._l)
            self._l = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, stream=None):
        """Enqueue one assemble -> solve -> advance; returns without synchronising."""
        self.solver._check(self.lib.nmpc_loop_step(self._l, stream))
        self.t += self.cfg.num_steps_taken
        self.steps += 1

    def read(self):
        """-> (state [B,3], last_u [B,2], idx [B], done [B] bool, status [B]) after synchronising."""
        import ctypes as C
        from . import _lib
        B = self.B
        state, last_u = np.empty((B, 3)), np.empty((B, 2))
        idx, done = np.empty(B, dtype=np.int32), np.empty(B, dtype=np.uint8)
        st = np.empty(B, dtype=_lib.STATUS_DTYPE)
        self.solver._check(self.lib.nmpc_loop_read(self._l, _lib.as_dp(state), _lib.as_dp(last_u),
                                                   idx.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   done.ctypes.data_as(C.POINTER(C.c_uint8)), st.ctypes.data))
        return state, last_u, idx, done.astype(bool), st

    def params(self):
        """-> (P [B,n_p] of the last step, U [B,n_u], Y [B,n1])."""
        from . import _lib
        P, U, Y = np.empty((self.B, self.cfg.n_p)), np.empty((self.B, self.cfg.n_u)), np.empty((self.B, self.cfg.n1))
        self.solver._check(self.lib.nmpc_loop_params(self._l, _lib.as_dp(P), _lib.as_dp(U), _lib.as_dp(Y)))
        return P, U, Y

    def trajectory(self):
        """-> [rows, B, 3]: the start poses and every pose reached so far (needs ``max_steps`` > 0)."""
        from . import _lib
        rows = self.steps * self.cfg.num_steps_taken + 1
        T = np.empty((rows, self.B, 3))
        n = self.lib.nmpc_loop_trajectory(self._l, _lib.as_dp(T), rows)
        if n < 0:
            self.solver._check(n)
        return T
