#!/usr/bin/env python3
"""CPU baseline of SURVEY 8d: the oracle (C++ restatement, BVH mode, all host threads) on C1-C3, best of 5.
   python tools/cpu_baseline.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import helpers as H  # noqa: E402
from rts_amd import scenes  # noqa: E402

threads = min(os.cpu_count() or 1, 64)
for name, spec in (("C1", scenes.config1()), ("C2", scenes.config2(rx_radius=200.0)), ("C3", scenes.config3())):
    sc = H.oracle_scene(O, spec, spec["motion"]); tx = spec["tx"]; W = spec["W"]
    best = None
    for rep in range(5):
        t0 = time.time()
        r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], use_bvh=True, threads=threads, debug=False)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, r["counters"]["segments"])
    print("%s: %d launch indices, %d segments, %.1f ms/pulse, %.2f Mrays/s (%d threads, best of 5)" % (name, W ** 3, best[1], best[0] * 1e3, best[1] / best[0] / 1e6, threads), flush=True)
