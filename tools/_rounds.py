import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rts_amd import api, scenes, _lib as L
for name in ("c3", "c3narrow"):
    spec = scenes.config3(rx_radius=50.0)
    if name == "c3narrow": spec["tx"] = dict(spec["tx"], span=(0.004, 0.004, 0.1))
    tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], count_traversal=True); tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"]); tx = spec["tx"]
    tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"]); tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"])
    L.lib().rts_debug_zero()
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"])
    d = np.zeros(64, np.uint64); L.lib().rts_debug_read(d.ctypes.data_as(C.c_void_p))
    print(name, "segments", st["segments"])
    print(" round: wave-iterations, lane-steps, lanes/iter | waves-in-round, lanes alive at start, alive/wave | iters/wave, steps/lane")
    for r in range(10):
        I, Ls, A, T = [int(d[k + r]) for k in (0, 10, 20, 30)]
        if T: print("  %d: %9d %11d %5.1f | %8d %10d %5.1f | %6.1f %6.1f" % (r, I, Ls, Ls / max(I, 1), T, A, A / T, I / T, Ls / max(A, 1)))
    tr.close()
