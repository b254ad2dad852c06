#!/usr/bin/env python3
"""Dump the DATA of the reference's 13 hard-coded scenes (src/visibility/graphs.py:21-191: boundary, obstacle
polygons, default start / end poses, dynamic-obstacle lists) into tests/golden/scenes.json and into the
package's scene table mpc_trajectory_generator_amd/scenes.json (what `frontend.scene_planner(cfg, k)` plans on).

Development container only (needs /root/reference; graphs.py imports nothing but numpy / matplotlib / math).
The output is data -- coordinates -- not code.   Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_scene_fixtures.py
"""
import json
import os
import sys

import matplotlib
matplotlib.use("Agg")

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "src"))

from visibility.graphs import Graphs  # noqa: E402

scenes = []
gs = Graphs()
for i, g in enumerate(gs.graphs):
    scenes.append(dict(index=i,
                       boundary=[[float(x), float(y)] for x, y in g.boundary_coordinates],
                       obstacles=[[[float(x), float(y)] for x, y in o] for o in g.obstacle_list],
                       start=[float(v) for v in g.start], end=[float(v) for v in g.end],
                       dyn_obs_list=[[[float(v) for v in o[0]], [float(v) for v in o[1]]] + [float(v) for v in o[2:]]
                                     for o in g.dyn_obs_list]))
PKG = os.path.join(os.path.dirname(os.path.dirname(OUT)), "mpc_trajectory_generator_amd", "scenes.json")
for path in (os.path.join(OUT, "scenes.json"), PKG):      # the fixture the tests check against + the package's scene table
    with open(path, "w") as fh:
        json.dump(scenes, fh, indent=0)
    print(len(scenes), "scenes ->", path)
