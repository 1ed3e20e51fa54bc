#!/usr/bin/env python3
"""What the receivers cost a launch: lone launches of a scene whole and with its meshes removed (what is left is the pre-filter, the tile-level
screen and whatever capture tests they could not exclude), with the blocks' own clocks (RTS_TIMELINE_BLOCKS): launch time, median / 90th percentile / last block end.
   python tools/rx_probe.py [c3 c3rx1 c3rx2 c5 c2 c2r200 sphere6 c4 ...]     (c3rx1 / c5: the monostatic case, the transmitter ON the capture sphere)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["RTS_TIMELINE_BLOCKS"] = "1"
from rts_amd import api, scenes
import numpy as np
which = sys.argv[1:] or ["c3", "c3rx1", "c3rx2", "c5"]
cases = {"c3": ("c3 (4 rx)", lambda: scenes.config3()), "c3rx1": ("c3 n_rx=1", lambda: scenes.config3(n_rx=1)), "c3rx2": ("c3 n_rx=2", lambda: scenes.config3(n_rx=2)), "c5": ("c5", lambda: scenes.config5()),
         "c2": ("c2", lambda: scenes.config2()), "c2r200": ("c2 rx 200 m", lambda: scenes.config2(rx_radius=200.0)), "sphere6": ("sphere6", lambda: scenes.config_sphere6()), "c4": ("c4", lambda: scenes.config4())}
for name, spec in ((cases[w][0], cases[w][1]()) for w in which):
    for strip in (False, True):
        sp = dict(spec)
        if strip:
            sp["meshes"] = []; sp["motion"] = []
        tr = api.Tracer(sp["W"], sp["max_refl"], 0, sp["smooth"])
        tr.set_scene(sp["meshes"]); tr.set_receivers(sp["rx"]); tx = sp["tx"]
        ms = []
        for k in range(6):
            st = tr.trace(tx["origin"], tx["span"], tx["dir"], sp["motion"])
            b = tr.block_timeline()
            if k >= 2: ms.append((st["ms_trace"], b["end_p50"]/1e3, b["end_p90"]/1e3, b["end_last"]/1e3))
        ms = np.array(ms).mean(0)
        print("%-12s %s: launch %.3f ms | block end p50 %.3f p90 %.3f last %.3f | segments %d received %d" % (name, "no meshes" if strip else "whole    ", ms[0], ms[1], ms[2], ms[3], st["segments"], st["received"]), flush=True)
        tr.close()
