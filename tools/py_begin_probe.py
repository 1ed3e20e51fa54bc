#!/usr/bin/env python3
"""where does the Python side of bench.py's trace_begin go?  (sphere6: 0.26 ms per call against 0.05 ms inside the library)"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (as bench.py: torch's HIP runtime first)
from rts_amd import api, scenes
import rts_amd._lib as L
which = sys.argv[1] if len(sys.argv) > 1 else "sphere6"
spec = scenes.config_sphere6() if which == "sphere6" else scenes.config3()
tx = spec["tx"]
trs = []
for i in range(3):
    t = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"])
    if i == 0: t.set_scene(spec["meshes"])
    else: t.share_scene(trs[0])
    t.set_receivers(spec["rx"]); t.reserve(0); trs.append(t)
def motion(k):
    if "motion_fn" in spec: return spec["motion_fn"](k)
    return [dict(position=tuple(np.asarray(m["position"]) + np.asarray(m["velocity"]) * k * 1e-3), velocity=m["velocity"]) for m in spec["motion"]]
N = 200
mos = [motion(k) for k in range(N)]
acc = dict(marshal=0.0, call=0.0, end=0.0)
pend = []
for k in range(N):
    t = trs[k % 3]
    a = time.perf_counter()
    p = t._pulse(tx["origin"], tx["span"], tx["dir"], mos[k], 0, 0, None)
    b = time.perf_counter()
    rc = L.lib().rts_trace_pulse_begin(t.h, C.byref(p))
    c = time.perf_counter()
    assert rc == 0
    pend.append(t)
    if len(pend) == 3:
        q = pend.pop(0); q.trace_end(); q.finalise_uniform(None, 0.03, 1, 1, 1e10, 3e8); q.aggregate(3e8, 1e10, 0, fetch=False); q.groups()
    d = time.perf_counter()
    if k >= 20:
        acc["marshal"] += b - a; acc["call"] += c - b; acc["end"] += d - c
print(which, {k: round(v / (N - 20) * 1e6, 1) for k, v in acc.items()}, "us per pulse")
