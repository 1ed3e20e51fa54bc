#!/usr/bin/env python3
"""node visits / triangle tests per segment of the bench scenes (counting build of the trace kernel)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rts_amd import api, scenes
import rts_amd._lib
rts_amd._lib.require_built()        # a timed tool never builds, and never measures a stale library
narrow = dict(scenes.config3(), tx=dict(scenes.config3()["tx"], span=(0.004, 0.004, 0.1)))
ecef = scenes.ecef_offset(lat=math.pi / 2)
cases = {"c3": lambda: scenes.config3(), "c2": lambda: scenes.config2(rx_radius=200.0), "c3narrow": lambda: narrow, "c3ecef": lambda: scenes.translate(scenes.config3(), ecef),
         "c3narrowecef": lambda: scenes.translate(narrow, ecef), "c4": lambda: scenes.config4(), "c5": lambda: scenes.config5()}
for name in (sys.argv[1:] or ["c3", "c2", "c3narrow", "c3ecef", "c3narrowecef"]):
    spec = cases[name]()
    tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], count_traversal=True)
    tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
    tx = spec["tx"]
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"])
    import ctypes as C
    ls = (C.c_uint64 * 5)(); rts_amd._lib.check(rts_amd._lib.lib().rts_get_walk_stats(tr.h, ls, 5))
    issued, alive, useful, _walked, never = [float(x) for x in ls]
    print("%-9s segs %d  nodes/seg %.2f  tri/seg %.2f  shaded/seg %.3f  trace %.3f ms | walk lane-steps: issued %.3e, to lanes in the round %.1f %%, taken %.1f %%  (lanes out of the round %.1f %%, waiting for the round's slowest %.1f %% -- of which lanes that never started a walk in the round %.1f %%, lanes whose walk was shorter %.1f %%)" %
          (name, st["segments"], st["node_visits"] / st["segments"], st["tri_tests"] / st["segments"], st["shaded"] / st["segments"], st["ms_trace"],
           issued, 100 * alive / max(issued, 1), 100 * useful / max(issued, 1), 100 * (1 - alive / max(issued, 1)), 100 * (alive - useful) / max(issued, 1), 100 * never / max(issued, 1), 100 * (alive - useful - never) / max(issued, 1)))
    tr.close()
