#!/usr/bin/env python3
"""rts_scene_info of the bench scenes: records, bytes shared per device and per handle, build time   (python tools/scene_info.py [c3 c4 ...])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rts_amd import api, scenes
import rts_amd._lib
rts_amd._lib.require_built()
cases = {"c2": lambda: scenes.config2(rx_radius=200.0), "c3": lambda: scenes.config3(), "c4": lambda: scenes.config4(), "c5": lambda: scenes.config5()}
for name in (sys.argv[1:] or ["c3", "c4"]):
    spec = cases[name]()
    tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"])
    tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tr.scene_info().items()})
    tr.close()
