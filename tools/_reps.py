import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rts_amd import api, scenes
spec = scenes.config3(rx_radius=50.0)
tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"]); tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"]); tx = spec["tx"]
for mode, step in (("moving 0.2 m/pulse", 0.2), ("static", 0.0), ("moving 0.02", 0.02)):
    ms = []
    for k in range(25):
        mo = [dict(position=tuple(np.add(m["position"], (step * k, 0.1 * step * k, 0.0))), velocity=m["velocity"]) for m in spec["motion"]]
        st = tr.trace(tx["origin"], tx["span"], tx["dir"], mo)
        ms.append(st["ms_trace"])
    print(mode, " ".join("%.3f" % m for m in ms))
