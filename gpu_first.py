# scratch: first GPU contact
import sys, time, numpy as np
sys.path.insert(0, "tests")
from oracle import oracle as O
from rts_amd import api, scenes
import helpers as H
for mk in (scenes.config1, lambda: scenes.config2(subdiv=2, W=24, rx_radius=400.0), lambda: scenes.config3(W=24, detail=0.05, rx_radius=400.0)):
    spec = mk()
    n = spec["W"]**3
    t0=time.time(); o = H.oracle_trace(O, spec); t1=time.time()
    tr, st = H.gpu_trace(api, spec)
    g = tr.all_rays(n)
    print(spec["name"], "oracle %.2fs"%(t1-t0), st)
    print(" received oracle", int((o["results"]["received"]>=0).sum()), "gpu", st["received"])
    H.compare_full(o, g, n)
    rec = tr.received()
    idx = np.nonzero(o["results"]["received"]>=0)[0]
    assert np.array_equal(rec["slots"], idx.astype(np.uint64))
    H.assert_prd_equal(o["results"][idx], rec["results"], "received")
    print(" PARITY OK")
