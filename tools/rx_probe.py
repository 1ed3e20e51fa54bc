import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["RTS_TIMELINE_BLOCKS"] = "1"
from rts_amd import api, scenes
import numpy as np
for name, spec in (("c3 (4 rx)", scenes.config3()), ("c3 n_rx=1", scenes.config3(n_rx=1)), ("c3 n_rx=2", scenes.config3(n_rx=2)), ("c5", scenes.config5())):
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
