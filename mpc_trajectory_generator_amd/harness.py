"""Host-side inputs of the NMPC solve: reference path, braking profile, obstacle selection and
the parameter vector ``p`` -- restated from the reference's receding-horizon driver so that the
batched solver is fed exactly what ``PathGenerator.run`` feeds OpEn.

Every function cites the reference lines it follows (paths relative to the reference repo).
Layout of ``p``: SURVEY.md Appendix A / reference src/path_generator.py:378-379.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .config import Config


# --------------------------------------------------------------------------------------------
# reference trajectory and braking tables
# --------------------------------------------------------------------------------------------
def rough_ref(cfg: Config, pos, node_list):
    """Constant-speed samples along the polyline ``pos -> node_list[0] -> ...``, one per ``ts``.

    Follows src/mpc/mpc_generator.py:17-57: speed 1.1 * throttle_ratio * lin_vel_max (:20), a
    sample is emitted after each full time step or when the last node is reached, heading is the
    direction of the last leg travelled in that step (:55).  If the first node equals ``pos`` the
    reference dies with an unbound local (:31-33,55); here that is a ValueError.
    """
    v = cfg.throttle_ratio * 1.1 * cfg.lin_vel_max
    x, y = float(pos[0]), float(pos[1])
    nodes = [(float(a), float(b)) for a, b in node_list]
    i = 0
    xt, yt = nodes[0]
    xs, ys, ths = [], [], []
    x_dir = y_dir = None
    traveling = True
    while traveling:
        t = cfg.ts
        while t > 0:
            dist = math.hypot(xt - x, yt - y)
            if dist == 0:
                traveling = False
                break
            x_dir, y_dir = (xt - x) / dist, (yt - y) / dist
            time_to_node = dist / v
            if time_to_node > t:
                x, y = x + x_dir * v * t, y + y_dir * v * t
                t = 0
            else:
                x, y = x + x_dir * v * time_to_node, y + y_dir * v * time_to_node
                t = t - time_to_node
                i += 1
                if i > len(nodes) - 1:
                    traveling = False
                    break
                xt, yt = nodes[i]
        if x_dir is None:
            raise ValueError("rough_ref: first node coincides with the start position")
        xs.append(x), ys.append(y), ths.append(math.atan2(y_dir, x_dir))
    return xs, ys, ths


def brake_vel_ref(cfg: Config):
    """Braking velocity / distance-to-goal tables, src/path_generator.py:439-477."""
    base_speed = cfg.lin_vel_max * cfg.throttle_ratio
    brake_acc = -base_speed / (cfg.ts * cfg.vel_red_steps)
    brake_acc = max(cfg.lin_acc_min, brake_acc)
    brake_time = -base_speed / brake_acc
    brake_dist = base_speed * brake_time + 0.5 * brake_acc * brake_time ** 2
    steps = math.ceil(brake_time / cfg.ts)
    vel = [base_speed - base_speed / (steps - 1) * i for i in range(steps)]
    dist = [0.0] * len(vel)
    dist[0] = brake_dist
    for i, vv in enumerate(vel):
        if i < len(dist) - 1:
            dist[i + 1] = dist[i] - vv * cfg.ts
    return vel, dist


def closest_index(pos, pts) -> int:
    """argmin of Euclidean distance, first minimum (src/visibility/visibility.py:111-124)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    d = np.linalg.norm(pts - np.asarray(pos, dtype=np.float64)[None, :], axis=1)
    return int(np.argmin(d))


def find_closest_vertices(vert, pos, n_vertices=10, look_back=2):
    """src/visibility/visibility.py:141-148, including its slice quirk: ``ub`` is not relative
    to ``lb`` so the window shrinks as the robot advances."""
    if n_vertices >= len(vert):
        return list(vert)
    idx = closest_index(pos, vert)
    lb = max(0, idx - look_back)
    ub = min(len(vert), n_vertices - look_back)
    return list(vert[lb:ub])


# --------------------------------------------------------------------------------------------
# dynamic obstacles (src/visibility/visibility.py:156-216)
# --------------------------------------------------------------------------------------------
def _linear_obstacle(p1, p2, freq, t):
    s = abs(math.sin(freq * t))                                  # :160
    return s * p1[0] + (1 - s) * p2[0], s * p1[1] + (1 - s) * p2[1]   # :164


def _rotate(origin, point, angle):                               # :169-175
    ox, oy = origin
    px, py = point
    return (math.cos(angle) * (px - ox) - math.sin(angle) * (py - oy),
            math.sin(angle) * (px - ox) + math.cos(angle) * (py - oy))


def _sinus_obstacle(p1, p2, freq, t, ampl=1.5):                  # :183-196
    angle = math.atan2(p2[1] - p1[1], p2[0] - p1[0])
    p3 = _linear_obstacle(p1, p2, freq, t)
    add = ampl * math.cos(10 * freq * t)
    rx, ry = _rotate(p1, p3, angle)                              # rotate_and_add :177-181
    ry += add
    qx, qy = _rotate((0.0, 0.0), (rx, ry), -angle)
    return qx + p1[0], qy + p1[1]


def dyn_obstacle(cfg: Config, dyn_obs_list, t, horizon, sinus_object=False):
    """Predicted ellipses ``[(x, y, rx, ry, angle)] * horizon`` per obstacle (:199-216).

    Sample times are ``linspace(t, t + horizon*ts, horizon)`` (:204), i.e. spaced
    horizon*ts/(horizon-1), not ts; radii are padded by
    vehicle_width/2 + vehicle_margin (:208-209); obstacle index 2 follows the sinusoidal
    law when ``sinus_object`` (:210-212)."""
    if len(dyn_obs_list) == 0:
        return []
    times = np.linspace(t, t + horizon * cfg.ts, horizon)
    pad = cfg.vehicle_width / 2 + cfg.vehicle_margin
    out = []
    for i, (p1, p2, freq, rx, ry, angle) in enumerate(dyn_obs_list):
        fn = _sinus_obstacle if (sinus_object and i == 2) else _linear_obstacle
        out.append([(*fn(p1, p2, freq, float(tt)), rx + pad, ry + pad, angle) for tt in times])
    return out


# --------------------------------------------------------------------------------------------
# parameter vector, one receding-horizon step
# --------------------------------------------------------------------------------------------
@dataclass
class Route:
    """What ``PathGenerator.run`` holds constant over a trajectory (src/path_generator.py:244-286)."""
    cfg: Config
    start: tuple
    end: tuple
    waypoints: list                      # A* path incl. start (``path`` at :244)
    vertices: list = field(default_factory=list)        # ppp.vert: NMPC circle centres (visibility.py:82-86)
    dyn_obs_list: list = field(default_factory=list)
    sinus_object: bool = False

    def __post_init__(self):
        cfg = self.cfg
        self.x_ref, self.y_ref, self.theta_ref = rough_ref(cfg, self.start[:2], self.waypoints[1:])   # :251
        self.ref_points = np.column_stack([self.x_ref, self.y_ref])                                # :269
        self.brake_velocities, self.brake_distances = brake_vel_ref(cfg)                          # :286
        self.base_speed = cfg.lin_vel_max * cfg.throttle_ratio                                    # :284
        self.radius = cfg.vehicle_width / 2 + cfg.vehicle_margin                                  # :301


def initial_dyn_constraints(cfg: Config):
    """Padding for the dynamic block: zeros with unit radii (src/path_generator.py:274-280)."""
    d = [0.0] * (cfg.Ndynobs * cfg.ndynobs * cfg.N_hor)
    d[2::cfg.ndynobs] = [1.0] * (cfg.Ndynobs * cfg.N_hor)
    d[3::cfg.ndynobs] = [1.0] * (cfg.Ndynobs * cfg.N_hor)
    return d


def static_constraints(route: Route, pos):
    """(x, y, r) per selected vertex, zero padded to Nobs*nobs (src/path_generator.py:295-304).
    Returns None when the scene has no obstacles (the reference then keeps its previous list)."""
    cfg = route.cfg
    origin = find_closest_vertices(route.vertices, pos, cfg.Nobs, 0)
    c = [val for (vx, vy) in origin for val in (float(vx), float(vy), route.radius)]
    c += [0.0] * (cfg.Nobs * cfg.nobs - len(c))
    return c


def horizon_refs(route: Route, idx: int, state):
    """x_finish, the N reference samples and vel_ref for reference index ``idx``
    (src/path_generator.py:326-361).  ``state`` is only read for the distance-to-goal branch."""
    cfg = route.cfg
    N = cfg.N_hor
    x_ref, y_ref, theta_ref, end = route.x_ref, route.y_ref, route.theta_ref, route.end
    n = len(x_ref)
    if idx + N >= n:                                                           # :326-333
        x_finish = [float(end[0]), float(end[1]), float(end[2])]
        pad = N - (n - idx)
        tmpx = x_ref[idx:] + [float(end[0])] * pad
        tmpy = y_ref[idx:] + [float(end[1])] * pad
        tmpt = theta_ref[idx:] + [float(end[2])] * pad
    else:                                                                      # :334-341
        x_finish = [x_ref[idx + N], y_ref[idx + N], theta_ref[idx + N]]
        tmpx, tmpy, tmpt = x_ref[idx:idx + N], y_ref[idx:idx + N], theta_ref[idx:idx + N]
    bv, bd, base = route.brake_velocities, route.brake_distances, route.base_speed
    if (idx + N) >= n - bd[0] / base:                                          # :344
        num_base = min(n - idx - 1, N)
        vel_ref = [base] * num_base
        if num_base == 0:                                                      # :347-351
            dist_to_goal = math.sqrt((state[0] - end[0]) ** 2 + (state[1] - end[1]) ** 2)
            vel_ref = [v for (v, d) in zip(bv, bd) if d <= dist_to_goal]
        else:                                                                  # :352-355
            vel_ref += bv[:min(len(bv), N - num_base)]
        vel_ref += [0.0] * (N - len(vel_ref))                                  # :358
        vel_ref = vel_ref[:N]
    else:
        vel_ref = [base] * N                                                   # :361
    refs = [0.0] * (N * cfg.nx)                                                # :363-365
    refs[0::cfg.nx] = tmpx
    refs[1::cfg.nx] = tmpy
    refs[2::cfg.nx] = tmpt
    return x_finish, vel_ref, refs


def assemble_params(route: Route, state, last_u, idx, constraints, dyn_constraints):
    """One parameter vector, concatenated as src/path_generator.py:378-379."""
    x_finish, vel_ref, refs = horizon_refs(route, idx, state)
    p = (list(map(float, state)) + list(map(float, last_u)) + x_finish + list(map(float, last_u))
         + route.cfg.weights() + vel_ref + list(constraints) + list(dyn_constraints) + refs)
    assert len(p) == route.cfg.n_p, (len(p), route.cfg.n_p)
    return p


# --------------------------------------------------------------------------------------------
# scenes: data of the reference's hard-coded maps that the benchmark configurations name
# --------------------------------------------------------------------------------------------
# src/visibility/graphs.py:34-43 (scene 1) and :161-170 (scene 11): obstacle polygons, start, end.
# Waypoints are the miter-offset (vehicle_width = 0.5) corner points of the shortest path between the
# default start/end: scene 1 as derived by hand in SURVEY.md section 8d (config 0), scene 11 as
# returned by this repo's own visibility-graph planner (frontend.py; the reference's front-end needs
# extremitypathfinder/pyclipper, which are not available).  tests/test_frontend.py re-derives both.
SCENES = {
    1: dict(
        start=(1.0, 5.0, math.radians(45)), end=(19.0, 10.0, math.radians(0)),
        waypoints=[(1.0, 5.0), (4.5, 15.5), (7.5, 15.5), (11.5, 12.0), (19.0, 10.0)],
        vertices=[(5.0, 15.0), (7.0, 15.0), (12.0, 12.5)],
    ),
    11: dict(
        start=(27.8, 2.7, math.radians(90)), end=(50.3, 45.9, math.radians(0)),
        waypoints=[(27.8, 2.7), (27.6, 5.5), (27.6, 33.5), (44.5, 33.6), (47.2, 36.69507401175115),
                   (55.8, 39.11309815931775), (55.8, 43.32261312113903), (50.3, 45.9)],
        vertices=[(28.1, 6.0), (28.1, 33.0), (44.0, 34.1), (47.7, 36.2), (55.3, 39.6), (55.3, 42.8)],
    ),
}


def scene_route(cfg: Config, scene: int) -> Route:
    s = SCENES[scene]
    return Route(cfg, s["start"], s["end"], list(s["waypoints"]), list(s["vertices"]))


# --------------------------------------------------------------------------------------------
# synthetic batches: the generator G(seed, B, cfg, scene) of SURVEY.md section 8d / BASELINE.md section 4
# --------------------------------------------------------------------------------------------
def synthetic_batch(cfg: Config, scene: int, B: int, seed: int, *, synthetic_circles: bool = False,
                    random_dyn: bool = False, routes=None):
    """-> P [B, n_p] float64: B independent instances at random points of a scene's route.

    Per instance: reference index idx ~ U{0..len-1}; state = reference sample
    idx + N(0, 0.05^2) m in x and y, heading theta_ref + N(0, 0.1^2); last_u = (U(0, v_max),
    U(-0.2, 0.2)); p filled exactly like the reference loop does for that idx; static circles by
    the find_closest_vertices rule (radius vehicle_width/2 + vehicle_margin), zero padded; dynamic
    block = the reference's padding unless ``random_dyn``.

    ``synthetic_circles`` (BASELINE config 3): all Nobs slots are filled with vertices of random
    convex polygons scattered over a 60 x 60 m box, rejected within 1.0 m of the route.
    ``random_dyn`` (BASELINE config 4): per-instance random ellipses crossing the route.
    ``routes``: a list of ``Route`` objects (e.g. ``frontend.random_routes``: randomised start/goal
    pairs planned on the scene's polygons); each instance then draws its route uniformly.  Default:
    the scene's own start -> end route.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    route_list = list(routes) if routes else [scene_route(cfg, scene)]
    route = route_list[0]
    N = cfg.N_hor
    P = np.empty((B, cfg.n_p), dtype=np.float64)
    circles = None
    if synthetic_circles:
        circles = _synthetic_circle_field(cfg, route, rng)
    pad_dyn = initial_dyn_constraints(cfg)
    for b in range(B):
        if len(route_list) > 1:
            route = route_list[int(rng.integers(0, len(route_list)))]
        n = len(route.x_ref)
        idx = int(rng.integers(0, n))
        state = [route.x_ref[idx] + rng.normal(0, 0.05), route.y_ref[idx] + rng.normal(0, 0.05),
                 route.theta_ref[idx] + rng.normal(0, 0.1)]
        last_u = [rng.uniform(0, cfg.lin_vel_max), rng.uniform(-0.2, 0.2)]
        if circles is None:
            cons = static_constraints(route, state[:2])
        else:
            d = np.hypot(circles[:, 0] - state[0], circles[:, 1] - state[1])
            sel = circles[np.argsort(d, kind="stable")[:cfg.Nobs]]
            cons = [val for (cx, cy) in sel for val in (float(cx), float(cy), route.radius)]
            cons += [0.0] * (cfg.Nobs * cfg.nobs - len(cons))
        dyn = pad_dyn
        if random_dyn:
            dyn = _random_dyn_block(cfg, route, idx, rng)
        P[b] = assemble_params(route, state, last_u, idx, cons, dyn)
    return P


def _synthetic_circle_field(cfg, route, rng):
    pts = []
    ref = route.ref_points
    while len(pts) < 400:
        cx, cy = rng.uniform(0, 60, 2)
        k = int(rng.integers(3, 9))
        rad = rng.uniform(0.5, 3.0)
        ang = np.sort(rng.uniform(0, 2 * math.pi, k))
        for a in ang:
            vx, vy = cx + rad * math.cos(a), cy + rad * math.sin(a)
            if np.min(np.hypot(ref[:, 0] - vx, ref[:, 1] - vy)) >= 1.0:
                pts.append((vx, vy))
    return np.array(pts)


def _random_dyn_block(cfg, route, idx, rng):
    """Ndynobs ellipses oscillating between two points near the route (scene-12 style,
    src/visibility/graphs.py:182-187), predicted over the horizon at t = 0."""
    n = len(route.x_ref)
    obs = []
    for _ in range(cfg.Ndynobs):
        j = min(n - 1, idx + int(rng.integers(0, cfg.N_hor + 10)))
        c = np.array([route.x_ref[j], route.y_ref[j]])
        p1 = (c + rng.uniform(-5, 5, 2)).tolist()
        p2 = (c + rng.uniform(-5, 5, 2)).tolist()
        obs.append([p1, p2, rng.uniform(0.05, 0.1), rng.uniform(0.3, 1.0), rng.uniform(0.3, 1.0),
                    rng.uniform(0, math.pi)])
    t0 = rng.uniform(0, 60.0)
    block = []
    for pred in dyn_obstacle(cfg, obs, t0, cfg.N_hor):
        for tup in pred:
            block += [float(v) for v in tup]
    return block
