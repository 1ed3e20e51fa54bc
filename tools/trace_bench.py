#!/usr/bin/env python3
"""Kernel-level A/B harness: times the stages of one pulse (HIP events inside the library) on the
bench scenes and prints a checksum of the received set, so that a kernel change that alters any
result bit is caught immediately.   python tools/trace_bench.py [c2|c3|c3s] [reps]"""
import hashlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rts_amd import api, scenes  # noqa: E402
import rts_amd._lib
rts_amd._lib.require_built()        # a timed tool never builds, and never measures a stale library

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = {"c2": lambda: scenes.config2(rx_radius=200.0), "c3": lambda: scenes.config3(rx_radius=50.0),
        "c3s": lambda: scenes.config3(W=100, rx_radius=50.0), "c3ico": lambda: scenes.config3(rx_radius=50.0, ico=True), "c3iconarrow": lambda: scenes.config3(rx_radius=50.0, ico=True), "c3nomesh": lambda: scenes.config3(rx_radius=50.0),
        "c3norx": lambda: scenes.config3(rx_radius=50.0), "c3empty": lambda: scenes.config3(rx_radius=50.0), "c3narrow": lambda: scenes.config3(rx_radius=50.0),
        "c3ecef": lambda: scenes.translate(scenes.config3(rx_radius=50.0), scenes.ecef_offset(lat=math.pi / 2)),
        "c3narrowecef": lambda: scenes.translate(scenes.config3(rx_radius=50.0), scenes.ecef_offset(lat=math.pi / 2)),
        "c4s": lambda: scenes.config4(W=232), "c4": lambda: scenes.config4(), "c4empty": lambda: scenes.config4(), "c4norx": lambda: scenes.config4(), "c5": lambda: scenes.config5()}[which]()
if which == "c3nomesh":
    spec["meshes"] = []; spec["motion"] = []
if which == "c4empty":
    spec["meshes"] = []; spec["motion"] = []; spec["rx"] = []
if which == "c4norx":
    spec["rx"] = []
if which == "c3empty":
    spec["meshes"] = []; spec["motion"] = []; spec["rx"] = []
if which == "c3norx":
    spec["rx"] = []
if which in ("c3narrow", "c3narrowecef", "c3iconarrow"):          # beam squeezed onto the fuselage: nearly every ray hits
    spec["tx"] = dict(spec["tx"], span=(0.004, 0.004, 0.1))
tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"])
tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
tx = spec["tx"]
ms = []
shard = int(os.environ.get("RTS_SHARD", "1"))                     # trace part 0 of `shard` interleaved parts of the launch (C4: one GPU's share of an 8-way ray split, rts_plan_cpi's 4096-index tiles)
count = spec["W"] ** 3
il = (int(os.environ.get("RTS_SHARD_TILE", "4096")), shard, int(os.environ.get("RTS_SHARD_PART", "0"))) if shard > 1 else None      # (RTS_SHARD_TILE: launch indices per interleaved tile, rts_plan_cpi's `tile`)
for k in range(reps + 1):
    if "motion_fn" in spec:
        mo = spec["motion_fn"](k)
    else:
        mo = [dict(position=tuple(np.add(m["position"], (0.2 * k, 0.02 * k, 0.0))), velocity=m["velocity"]) for m in spec["motion"]]
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], mo, ray_first=0, ray_count=count, interleave=il)
    if k:
        ms.append((st["ms_scene"], st["ms_trace"], st["ms_compact"]))
rec = tr.received()
h = hashlib.sha1(np.ascontiguousarray(rec["results"]["power"]).tobytes() + np.ascontiguousarray(rec["slots"]).tobytes() +
                 np.ascontiguousarray(rec["path"]).tobytes()).hexdigest()[:16]
ms = np.array(ms)
if os.environ.get("RTS_VERBOSE"):
    print("trace ms per launch:", " ".join("%.3f" % x for x in ms[:, 1]))
print("%s: segs %d recv %d | scene %.3f trace %.3f (min %.3f) compact %.3f ms | %.2f Gseg/s | sha %s" %
      (spec["name"], st["segments"], st["received"], ms[:, 0].mean(), ms[:, 1].mean(), ms[:, 1].min(), ms[:, 2].mean(),
       st["segments"] / ms[:, 1].min() / 1e6, h))
