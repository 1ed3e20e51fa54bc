#!/usr/bin/env python3
"""What makes the slowest tiles slow: traces single tiles and single rays of the C3 pulse with the counting build.
   python tools/slow_tile.py [tile ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rts_amd import api, scenes  # noqa: E402
import rts_amd._lib
rts_amd._lib.require_built()        # a timed tool never builds, and never measures a stale library
args = sys.argv[1:]
which = args.pop(0) if args and not args[0].isdigit() else "c3"        # python tools/slow_tile.py [c3|c4|c5] [tile ...]
tiles = [int(x) for x in args] or ([83452, 82689, 81989] if which == "c3" else [786329, 786322])
spec = scenes.config4() if which == "c4" else scenes.config5() if which == "c5" else scenes.config3()
tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], count_traversal=True)
tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"]); tx = spec["tx"]
for t in tiles:
    for rep in range(2):
        st = tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"], ray_first=t * 64, ray_count=64)
    print("tile %d alone: %.1f us, segments %d nodes %d tris %d shaded %d" % (t, st["ms_trace"] * 1e3, st["segments"], st["node_visits"], st["tri_tests"], st["shaded"]))
    rows = []
    for r in range(64):
        st = tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"], ray_first=t * 64 + r, ray_count=1)
        rows.append((st["ms_trace"] * 1e3, st["segments"], st["node_visits"], st["tri_tests"]))
    rows = np.array(rows)
    o = np.argsort(rows[:, 0])[::-1]
    print("  per ray (us, segments, nodes, tris), slowest 8:", [tuple(int(v) for v in rows[i]) for i in o[:8]])
    print("  per ray totals: steps max %d mean %.0f; us max %.0f median %.0f; us per step (slowest ray) %.3f" % ((rows[:, 2] + rows[:, 3]).max(), (rows[:, 2] + rows[:, 3]).mean(), rows[:, 0].max(), np.median(rows[:, 0]), rows[o[0], 0] / max(rows[o[0], 2] + rows[o[0], 3], 1)))
