#!/usr/bin/env python3
"""What the GPU makes of N trace launches that are ALL enqueued before any is waited for (no host in the loop, no post-processing):
N handles of one scene, one pulse each per round, begin x N then end x N; the time per pulse against N says whether the pipelined
bench (two or three pulses in flight behind a submitting thread) is bounded by the chip or by the depth of its pipeline.
   python tools/inflight_probe.py [c3|c2|c4|c5] [max handles] [rounds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rts_amd import api, scenes  # noqa: E402
import rts_amd._lib
rts_amd._lib.require_built()
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
spec = {"c2": lambda: scenes.config2(rx_radius=200.0), "c3": lambda: scenes.config3(rx_radius=50.0), "c4": scenes.config4, "c5": scenes.config5}[which]()
tx = spec["tx"]
trs = []
for i in range(nmax):
    t = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"])
    if i == 0:
        t.set_scene(spec["meshes"])
    else:
        t.share_scene(trs[0])
    t.set_receivers(spec["rx"]); trs.append(t)


def motion(k):
    if "motion_fn" in spec:
        return spec["motion_fn"](k)
    return [dict(position=tuple(np.add(m["position"], (0.2 * k, 0.02 * k, 0.0))), velocity=m["velocity"]) for m in spec["motion"]]


k = 0
for t in trs:                       # every handle has seen the pulse a few times (tile-cost history, streams, buffers)
    for _ in range(3):
        t.trace(tx["origin"], tx["span"], tx["dir"], motion(k)); k += 1
for n in [1, 2, 3, 4, 6, 8, 12, 16]:
    if n > nmax:
        break
    per = []
    for r in range(rounds + 1):
        mos = [motion(k + i) for i in range(n)]; k += n
        t0 = time.perf_counter()
        for i in range(n):
            trs[i].trace_begin(tx["origin"], tx["span"], tx["dir"], mos[i])
        t1 = time.perf_counter()
        for i in range(n):
            trs[i].trace_end()
        t2 = time.perf_counter()
        if r:
            per.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3))
    per = np.array(per)
    print("%s: %2d launches enqueued together: %.3f ms in all (min %.3f) = %.3f ms per pulse (min %.3f); enqueueing them took %.3f ms" %
          (which, n, per[:, 0].mean(), per[:, 0].min(), per[:, 0].mean() / n, per[:, 0].min() / n, per[:, 1].mean()), flush=True)
