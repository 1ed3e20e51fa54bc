"""C3's scene with the refraction branch on (max_refr = 2, refractive index 1.3): ms per launch of the REFR trace kernel"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rts_amd import api, scenes
spec = scenes.config3(rx_radius=50.0)
for m in spec["meshes"]: m["refr_index"] = 1.3
tr = api.Tracer(spec["W"], spec["max_refl"], 2, spec["smooth"]); tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"]); tx = spec["tx"]
ms = []
for k in range(10):
    mo = [dict(position=tuple(np.add(m["position"], (0.2 * k, 0.02 * k, 0.0))), velocity=m["velocity"]) for m in spec["motion"]]
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], mo); ms.append(st["ms_trace"])
print("refraction on C3: segs %d recv %d | trace ms per launch: %s | settled min %.3f" % (st["segments"], st["received"], " ".join("%.3f" % x for x in ms), min(ms[2:])))
