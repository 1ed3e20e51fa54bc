#!/usr/bin/env python3
"""node visits / triangle tests per segment of the bench scenes (counting build of the trace kernel)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rts_amd import api, scenes
import rts_amd._lib
rts_amd._lib.require_built()        # a timed tool never builds, and never measures a stale library
narrow = dict(scenes.config3(), tx=dict(scenes.config3()["tx"], span=(0.004, 0.004, 0.1)))
ecef = scenes.ecef_offset(lat=math.pi / 2)
for name, spec in (("c3", scenes.config3()), ("c2", scenes.config2(rx_radius=200.0)), ("c3narrow", narrow),
                   ("c3ecef", scenes.translate(scenes.config3(), ecef)), ("c3narrowecef", scenes.translate(narrow, ecef))):
    tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], count_traversal=True)
    tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
    tx = spec["tx"]
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"])
    print("%-9s segs %d  nodes/seg %.2f  tri/seg %.2f  shaded/seg %.3f  trace %.3f ms" % (name, st["segments"], st["node_visits"] / st["segments"], st["tri_tests"] / st["segments"], st["shaded"] / st["segments"], st["ms_trace"]))
    tr.close()
