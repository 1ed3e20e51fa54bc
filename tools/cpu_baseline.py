#!/usr/bin/env python3
"""CPU baseline of SURVEY 8d: the oracle (C++ restatement, BVH mode, -O3 -march=native -ffp-contract=off build made on this
machine, every hardware thread) on C1-C3 as whole pulses (best of 5) and on C4 / C5 EXTRAPOLATED from a 1 % sample of the
launch indices (every 100th index, labelled as such).  The reference has no CPU path; this is the restatement, reported
as a baseline, not a target.
   python tools/cpu_baseline.py [c1 c2 c2file c3 c4 c5]"""
import os, sys, time
os.environ["RTS_ORACLE_NATIVE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import helpers as H  # noqa: E402
from rts_amd import scenes  # noqa: E402


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


threads = os.cpu_count() or 1
which = [a.lower() for a in sys.argv[1:]] or ["c1", "c2", "c2file", "c3", "c4", "c5"]
specs = {"c1": lambda: scenes.config1(), "c2": lambda: scenes.config2(rx_radius=200.0), "c2file": lambda: scenes.config2_file(rx_radius=200.0),
         "c3": lambda: scenes.config3(), "c4": lambda: scenes.config4(), "c5": lambda: scenes.config5()}
print("CPU: %s, %d hardware threads; oracle/rts_oracle.cpp, g++ -O3 -march=native -ffp-contract=off, BVH mode" % (cpu_model(), threads), flush=True)
for name in which:
    spec = specs[name]()
    motion = spec["motion_fn"](3) if "motion_fn" in spec else spec["motion"]
    sc = H.oracle_scene(O, spec, motion); tx = spec["tx"]; W = spec["W"]; total = W ** 3
    sample = name in ("c4", "c5")
    stride = 100 if sample else 1
    n = total // stride
    best = None
    for rep in range(3 if sample else 5):
        t0 = time.time()
        r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=stride, n_rays=n, use_bvh=True, threads=threads, debug=False, reuse_buffers=True)
        dt = time.time() - t0
        if rep == 0:
            continue                                   # the first call builds the oracle's BVH
        if best is None or dt < best[0]:
            best = (dt, r["counters"]["segments"])
    rate = best[1] / best[0] / 1e6
    if sample:
        print("%s (%s): 1 %% sample (every 100th of %d launch indices): %d segments in %.1f ms -> %.2f Mrays/s; EXTRAPOLATED %.0f ms/pulse"
              % (name.upper(), spec["name"], total, best[1], best[0] * 1e3, rate, best[0] * 1e3 * stride), flush=True)
    else:
        print("%s (%s): %d launch indices, %d segments, %.1f ms/pulse, %.2f Mrays/s (best of 4)" % (name.upper(), spec["name"], total, best[1], best[0] * 1e3, rate), flush=True)
