"""Route front-end: polygon inflation + visibility-graph shortest path.

The reference gets its route from third-party code: ``pyclipper`` inflates every obstacle by the
vehicle width with mitred joins and deflates the boundary (src/visibility/visibility.py:49-67,
90-105), ``extremitypathfinder`` builds a visibility graph over the inflated polygons and runs A*
(:69-88), and the path corners are mapped back to the nearest original vertices, which become the
NMPC's circle centres (:126-139).  Neither package is available here, so this module is an own
implementation of the same geometry -- **unpinned** against the reference's dependencies
(validated geometrically: tests/test_frontend.py).  It runs once per trajectory on the CPU; it is
not on the GPU hot path.
"""
from __future__ import annotations

import heapq
import math

import numpy as np

from . import harness
from .config import Config

_EPS = 1e-9


def _signed_area(poly):
    a = 0.0
    for i in range(len(poly)):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % len(poly)]
        a += x1 * y2 - x2 * y1
    return 0.5 * a


def offset_polygon(poly, delta, miter_limit=2.0):
    """Mitred offset of a simple polygon by ``delta`` (> 0 grows it, < 0 shrinks it), the join
    style the reference asks pyclipper for (JT_MITER, visibility.py:92).  A corner whose mitre
    would reach further than ``miter_limit * |delta|`` is squared off with two points."""
    pts = [(float(x), float(y)) for x, y in poly]
    if _signed_area(pts) < 0:                      # work counter-clockwise; restore the order at the end
        pts = pts[::-1]
        flipped = True
    else:
        flipped = False
    n = len(pts)
    out = []
    for i in range(n):
        p0, p1, p2 = pts[i - 1], pts[i], pts[(i + 1) % n]
        e1 = (p1[0] - p0[0], p1[1] - p0[1])
        e2 = (p2[0] - p1[0], p2[1] - p1[1])
        l1, l2 = math.hypot(*e1), math.hypot(*e2)
        n1 = (e1[1] / l1, -e1[0] / l1)             # outward normals of a CCW polygon
        n2 = (e2[1] / l2, -e2[0] / l2)
        cos_t = n1[0] * n2[0] + n1[1] * n2[1]
        denom = 1.0 + cos_t
        if denom < 1e-12:                          # 180 degree turn-back: square off
            out.append((p1[0] + delta * n1[0], p1[1] + delta * n1[1]))
            out.append((p1[0] + delta * n2[0], p1[1] + delta * n2[1]))
            continue
        mx, my = (n1[0] + n2[0]) / denom, (n1[1] + n2[1]) / denom      # mitre vector per unit delta
        if math.hypot(mx, my) > miter_limit:
            out.append((p1[0] + delta * n1[0], p1[1] + delta * n1[1]))
            out.append((p1[0] + delta * n2[0], p1[1] + delta * n2[1]))
        else:
            out.append((p1[0] + delta * mx, p1[1] + delta * my))
    return out[::-1] if flipped else out


def _seg_intersect_strict(a, b, c, d):
    """Proper crossing of open segments ab and cd (touching at endpoints / collinear overlap is not a crossing)."""
    def orient(p, q, r):
        return (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
    o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
    return (o1 * o2 < -_EPS) and (o3 * o4 < -_EPS)


def _point_in_polygon(p, poly, strict=True):
    """Even-odd test; points on the boundary count as outside when ``strict``."""
    x, y = p
    n = len(poly)
    for i in range(n):                              # on an edge?
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % n]
        cross = (x2 - x1) * (y - y1) - (y2 - y1) * (x - x1)
        if abs(cross) <= 1e-9 * max(1.0, math.hypot(x2 - x1, y2 - y1)):
            if min(x1, x2) - 1e-9 <= x <= max(x1, x2) + 1e-9 and min(y1, y2) - 1e-9 <= y <= max(y1, y2) + 1e-9:
                return not strict
    inside = False
    for i in range(n):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % n]
        if (y1 > y) != (y2 > y):
            xi = x1 + (y - y1) * (x2 - x1) / (y2 - y1)
            if xi > x:
                inside = not inside
    return inside


class VisibilityPlanner:
    """Counterpart of ``PathPreProcessor.prepare`` + ``get_initial_guess`` (visibility.py:49-88,126-139)."""

    def __init__(self, cfg: Config, boundary, obstacles, dyn_obs_list=()):
        self.cfg = cfg
        self.original_boundary = [tuple(map(float, p)) for p in boundary]
        self.original_obstacles = [[tuple(map(float, p)) for p in o] for o in obstacles]
        self.dyn_obs_list = list(dyn_obs_list)
        w = float(cfg.vehicle_width)
        self.obstacles = [offset_polygon(o, +w) for o in self.original_obstacles]          # (:97)
        self.boundary = offset_polygon(self.original_boundary, -w)                          # (:59-61)
        self.nodes = []
        for poly in self.obstacles:
            self.nodes += self._extremities(poly, hole=True)
        self.nodes += self._extremities(self.boundary, hole=False)

    @staticmethod
    def _extremities(poly, hole):
        """Corners a shortest path can bend around: convex corners of obstacles, reflex corners of the boundary."""
        ccw = _signed_area(poly) > 0
        out = []
        n = len(poly)
        for i in range(n):
            p0, p1, p2 = poly[i - 1], poly[i], poly[(i + 1) % n]
            cross = (p1[0] - p0[0]) * (p2[1] - p1[1]) - (p1[1] - p0[1]) * (p2[0] - p1[0])
            convex = cross > 1e-12 if ccw else cross < -1e-12
            if convex == hole:
                out.append(p1)
        return out

    def _free(self, a, b):
        """Is the open segment ab collision free (outside every inflated obstacle, inside the boundary)?"""
        if math.hypot(a[0] - b[0], a[1] - b[1]) < 1e-12:
            return True
        polys = self.obstacles + [self.boundary]
        for poly in polys:
            n = len(poly)
            for i in range(n):
                if _seg_intersect_strict(a, b, poly[i], poly[(i + 1) % n]):
                    return False
        for s in (0.5, 0.25, 0.75, 0.0625, 0.9375):          # sample the interior against containment
            m = (a[0] + s * (b[0] - a[0]), a[1] + s * (b[1] - a[1]))
            if any(_point_in_polygon(m, o, strict=True) for o in self.obstacles):
                return False
            if not _point_in_polygon(m, self.boundary, strict=False):
                return False
        return True

    def shortest_path(self, start, goal):
        """A* over the visibility graph; -> (waypoints incl. start and goal, length)."""
        s, g = (float(start[0]), float(start[1])), (float(goal[0]), float(goal[1]))
        pts = [s, g] + self.nodes
        n = len(pts)
        vis = {}

        def visible(i, j):
            key = (i, j) if i < j else (j, i)
            if key not in vis:
                vis[key] = self._free(pts[i], pts[j])
            return vis[key]

        def h(i):
            return math.hypot(pts[i][0] - g[0], pts[i][1] - g[1])
        dist = {0: 0.0}
        prev = {}
        heap = [(h(0), 0)]
        done = set()
        while heap:
            _, i = heapq.heappop(heap)
            if i in done:
                continue
            done.add(i)
            if i == 1:
                break
            for j in range(n):
                if j == i or j in done or not visible(i, j):
                    continue
                d = dist[i] + math.hypot(pts[i][0] - pts[j][0], pts[i][1] - pts[j][1])
                if d < dist.get(j, math.inf) - 1e-12:
                    dist[j], prev[j] = d, i
                    heapq.heappush(heap, (d + h(j), j))
        if 1 not in dist:
            raise ValueError("no collision-free path between start and goal")
        path, i = [], 1
        while True:
            path.append(pts[i])
            if i == 0:
                break
            i = prev[i]
        return path[::-1], dist[1]

    def original_vertices(self, path):
        """Closest original (un-inflated) vertex, obstacles and boundary alike, for each interior
        path corner (visibility.py:126-139)."""
        if len(path) <= 2:
            return []
        allv = [v for o in self.original_obstacles for v in o] + list(self.original_boundary)
        return [allv[harness.closest_index(c, allv)] for c in path[1:-1]]

    def route(self, start, end, sinus_object=False) -> harness.Route:
        path, _ = self.shortest_path(start[:2], end[:2])
        return harness.Route(self.cfg, tuple(start), tuple(end), path, self.original_vertices(path),
                             self.dyn_obs_list, sinus_object)


# scene data of the reference's 13 maps (src/visibility/graphs.py:21-191): boundary, obstacle polygons, default
# start / end poses and dynamic-obstacle lists -- a table of coordinates (scenes.json beside this file, written by
# tests/golden/make_scene_fixtures.py)
def _load_scenes():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes.json")) as fh:
        raw = json.load(fh)
    out = {}
    for g in raw:
        out[g["index"]] = dict(boundary=[tuple(p) for p in g["boundary"]],
                               obstacles=[[tuple(p) for p in o] for o in g["obstacles"]],
                               start=tuple(g["start"]), end=tuple(g["end"]),
                               dyn_obs_list=[[tuple(o[0]), tuple(o[1])] + list(o[2:]) for o in g["dyn_obs_list"]])
    return out


SCENE_POLYGONS = _load_scenes()


def scene_planner(cfg: Config, scene: int) -> VisibilityPlanner:
    """Planner on scene 0..12 of the reference (``Graphs().get_graph(complexity)``, graphs.py:199-203)."""
    s = SCENE_POLYGONS[scene]
    return VisibilityPlanner(cfg, s["boundary"], s["obstacles"], s["dyn_obs_list"])


def random_routes(cfg: Config, scene: int, n: int, seed: int, min_length: float = 12.0):
    """``n`` routes between random collision-free start / goal points of a scene (BASELINE config 1:
    "randomized start/goal"), planned on the inflated polygons like the reference's front-end."""
    pl = scene_planner(cfg, scene)
    rng = np.random.Generator(np.random.PCG64(seed))
    xs = [p[0] for p in pl.boundary]
    ys = [p[1] for p in pl.boundary]

    def sample():
        while True:
            p = (rng.uniform(min(xs), max(xs)), rng.uniform(min(ys), max(ys)))
            if not _point_in_polygon(p, pl.boundary, strict=True):
                continue
            if any(_point_in_polygon(p, offset_polygon(o, 0.25), strict=False) for o in pl.obstacles):
                continue                                   # keep a little clear of the inflated obstacles
            return p
    out = []
    while len(out) < n:
        s, g = sample(), sample()
        try:
            path, length = pl.shortest_path(s, g)
        except ValueError:
            continue
        if length < min_length:
            continue
        th0 = math.atan2(path[1][1] - path[0][1], path[1][0] - path[0][0])
        th1 = math.atan2(path[-1][1] - path[-2][1], path[-1][0] - path[-2][0])
        out.append(harness.Route(cfg, (s[0], s[1], th0), (g[0], g[1], th1), path, pl.original_vertices(path)))
    return out
