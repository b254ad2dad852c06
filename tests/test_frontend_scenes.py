"""CPU: the visibility-graph front-end on ALL 13 scenes of the reference (data: tests/golden/scenes.json, dumped from
src/visibility/graphs.py:21-191 by tests/golden/make_scene_fixtures.py).  The reference delegates this stage to
pyclipper + extremitypathfinder (src/visibility/visibility.py:49-139), neither of which is installed, so the checks
are independent ones: an 8-connected grid Dijkstra over the same inflated map brackets the path length from both
sides, dense sampling proves the path collision-free, and the path-corner -> original-vertex mapping is checked
against the semantics of ``find_original_vertices`` (:126-139: nearest vertex among all obstacle AND boundary
vertices, start and goal excluded)."""
import heapq
import json
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN
from mpc_trajectory_generator_amd import named_config
from mpc_trajectory_generator_amd.frontend import SCENE_POLYGONS, _point_in_polygon, scene_planner

SCENES = json.load(open(os.path.join(GOLDEN, "scenes.json")))
RES = 0.25


def test_package_scene_table_equals_fixture():
    assert sorted(SCENE_POLYGONS) == list(range(13))
    for g in SCENES:
        s = SCENE_POLYGONS[g["index"]]
        assert [list(p) for p in s["boundary"]] == g["boundary"]
        assert [[list(p) for p in o] for o in s["obstacles"]] == g["obstacles"]
        assert list(s["start"]) == g["start"] and list(s["end"]) == g["end"]
        assert len(s["dyn_obs_list"]) == len(g["dyn_obs_list"])


def _free_point(pl, p):
    return _point_in_polygon(p, pl.boundary, strict=False) and not any(_point_in_polygon(p, o, strict=True) for o in pl.obstacles)


def _grid_shortest(pl, start, goal):
    """Dijkstra on an 8-connected grid of spacing RES whose nodes are the free points of the inflated map
    (start and goal added as extra nodes linked to the free grid points within 1.5 RES)."""
    xs = [p[0] for p in pl.boundary]
    ys = [p[1] for p in pl.boundary]
    x0, y0 = math.floor(min(xs) / RES) * RES, math.floor(min(ys) / RES) * RES
    nx, ny = int((max(xs) - x0) / RES) + 2, int((max(ys) - y0) / RES) + 2
    free = np.zeros((nx, ny), dtype=bool)
    for i in range(nx):
        for j in range(ny):
            free[i, j] = _free_point(pl, (x0 + i * RES, y0 + j * RES))
    def near(p):
        i0, j0 = int(round((p[0] - x0) / RES)), int(round((p[1] - y0) / RES))
        out = []
        for i in range(i0 - 2, i0 + 3):
            for j in range(j0 - 2, j0 + 3):
                if 0 <= i < nx and 0 <= j < ny and free[i, j]:
                    d = math.hypot(x0 + i * RES - p[0], y0 + j * RES - p[1])
                    if d <= 1.5 * RES:
                        out.append(((i, j), d))
        return out
    dist = {}
    heap = [(d, n) for n, d in near(start)]
    for d, n in heap:
        dist[n] = min(d, dist.get(n, math.inf))
    heapq.heapify(heap)
    goal_links = dict(near(goal))
    best = math.inf
    done = set()
    while heap:
        d, n = heapq.heappop(heap)
        if n in done:
            continue
        done.add(n)
        if n in goal_links:
            best = min(best, d + goal_links[n])
        if d > best:
            break
        i, j = n
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                if (di or dj) and 0 <= i + di < nx and 0 <= j + dj < ny and free[i + di, j + dj]:
                    nd = d + RES * math.hypot(di, dj)
                    m = (i + di, j + dj)
                    if nd < dist.get(m, math.inf):
                        dist[m] = nd
                        heapq.heappush(heap, (nd, m))
    return best


@pytest.mark.parametrize("g", SCENES, ids=lambda g: f"scene{g['index']}")
def test_every_scene_plans_a_valid_shortest_path(g):
    cfg = named_config("cfg1")
    pl = scene_planner(cfg, g["index"])
    start, goal = tuple(g["start"][:2]), tuple(g["end"][:2])
    if not (_free_point(pl, start) and _free_point(pl, goal)):
        # the scene's default poses lie inside the inflated map (the reference's own front-end would fail there too)
        with pytest.raises(ValueError):
            pl.shortest_path(start, goal)
        return
    path, length = pl.shortest_path(start, goal)
    assert path[0] == start and path[-1] == goal
    assert abs(length - sum(math.dist(a, b) for a, b in zip(path, path[1:]))) < 1e-9
    assert length >= math.dist(start, goal) - 1e-12
    # collision-free against the inflated polygons, inside the deflated boundary
    for a, b in zip(path, path[1:]):
        for k in range(1, 60):
            m = (a[0] + k / 60 * (b[0] - a[0]), a[1] + k / 60 * (b[1] - a[1]))
            assert not any(_point_in_polygon(m, o, strict=True) for o in pl.obstacles)
            assert _point_in_polygon(m, pl.boundary, strict=False)
    # interior corners are corners of inflated polygons, and a shortest path turns at each of them
    for c in path[1:-1]:
        assert any(math.dist(c, v) < 1e-9 for v in pl.nodes)
    # bracket by the grid search: a grid path is never shorter than the true shortest path minus the snapping slack and
    # at most 8.24 % longer (octile metric) plus slack
    lg = _grid_shortest(pl, start, goal)
    assert math.isfinite(lg)
    assert length <= lg + 3 * RES, (length, lg)
    assert length >= lg / 1.0824 - 4 * RES, (length, lg)
    # corner -> original vertex mapping (visibility.py:126-139)
    allv = [tuple(v) for o in g["obstacles"] for v in o] + [tuple(v) for v in g["boundary"]]
    want = [min(allv, key=lambda v: math.dist(v, c)) for c in path[1:-1]]
    got = pl.original_vertices(path)
    assert len(got) == len(path) - 2
    for c, w_, v in zip(path[1:-1], want, got):
        assert math.dist(v, c) == pytest.approx(math.dist(w_, c), abs=1e-12)
        # the mapped vertex is the one the corner was offset from: at most the mitre length away
        assert math.dist(v, c) <= 2.0 * cfg.vehicle_width + 1e-9
    # a route object (rough reference, braking tables) comes out of it
    r = pl.route(tuple(g["start"]), tuple(g["end"]))
    assert len(r.x_ref) >= 1 and np.all(np.isfinite(r.x_ref))
