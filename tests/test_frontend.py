"""CPU: own visibility-graph front-end (unpinned against pyclipper / extremitypathfinder, which are
absent): geometric validation."""
import math

import numpy as np

from mpc_trajectory_generator_amd import named_config, harness
from mpc_trajectory_generator_amd.frontend import (VisibilityPlanner, _point_in_polygon, offset_polygon,
                                                   random_routes, scene_planner)


def test_miter_offset_rectangles():
    assert offset_polygon([(5.0, 0.0), (5.0, 15.0), (7.0, 15.0), (7.0, 0.0)], 0.5) == \
        [(4.5, -0.5), (4.5, 15.5), (7.5, 15.5), (7.5, -0.5)]
    assert offset_polygon([(0.0, 0.0), (20.0, 0.0), (20.0, 20.0), (0.0, 20.0)], -0.5) == \
        [(0.5, 0.5), (19.5, 0.5), (19.5, 19.5), (0.5, 19.5)]
    tri = offset_polygon([(0, 0), (4, 0), (0, 3)], 0.5)
    for (x, y) in [(0, 0), (4, 0), (0, 3)]:
        assert _point_in_polygon((x, y), tri)
    # every offset edge is 0.5 away from the original edge line
    assert abs(tri[0][1] - (-0.5)) < 1e-12 or abs(tri[0][0] - (-0.5)) < 1e-12


def test_scene1_route_matches_geometry_of_the_survey():
    """(1,5)->(4.5,15.5)->(7.5,15.5)->(11.5,12)->(19,10), vertices (5,15),(7,15),(12,12.5)."""
    cfg = named_config("cfg1")
    pl = scene_planner(cfg, 1)
    path, length = pl.shortest_path((1, 5), (19, 10))
    assert path == [(1.0, 5.0), (4.5, 15.5), (7.5, 15.5), (11.5, 12.0), (19.0, 10.0)]
    assert pl.original_vertices(path) == [(5.0, 15.0), (7.0, 15.0), (12.0, 12.5)]
    assert path == harness.SCENES[1]["waypoints"] and pl.original_vertices(path) == harness.SCENES[1]["vertices"]
    assert abs(length - sum(math.dist(a, b) for a, b in zip(path, path[1:]))) < 1e-12


def test_paths_are_collision_free_and_locally_shortest():
    cfg = named_config("cfg1")
    for scene in (1, 11):
        pl = scene_planner(cfg, scene)
        s = harness.SCENES[scene]
        path, length = pl.shortest_path(s["start"], s["end"])
        hand = sum(math.dist(a, b) for a, b in zip(s["waypoints"], s["waypoints"][1:]))
        assert length <= hand + 1e-9                      # never longer than the hand-derived route
        for a, b in zip(path, path[1:]):
            for k in range(1, 40):
                m = (a[0] + k / 40 * (b[0] - a[0]), a[1] + k / 40 * (b[1] - a[1]))
                assert not any(_point_in_polygon(m, o, strict=True) for o in pl.obstacles)
                assert _point_in_polygon(m, pl.boundary, strict=False)
        for c in path[1:-1]:                              # interior corners are inflated-polygon vertices
            assert any(math.dist(c, v) < 1e-9 for v in pl.nodes)


def test_random_routes_feed_the_batch_generator():
    cfg = named_config("cfg1")
    routes = random_routes(cfg, 11, 4, seed=1)
    assert len(routes) == 4 and all(len(r.x_ref) > 20 for r in routes)
    P = harness.synthetic_batch(cfg, 11, 32, 0, routes=routes)
    assert P.shape == (32, cfg.n_p) and np.all(np.isfinite(P))
    assert np.array_equal(P, harness.synthetic_batch(cfg, 11, 32, 0, routes=random_routes(cfg, 11, 4, seed=1)))


def test_no_path_raises():
    cfg = named_config("cfg1")
    pl = VisibilityPlanner(cfg, [(0, 0), (10, 0), (10, 10), (0, 10)], [[(4, -1), (4, 11), (6, 11), (6, -1)]])
    try:
        pl.shortest_path((1, 5), (9, 5))
        assert False
    except ValueError:
        pass
