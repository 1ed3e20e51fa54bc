"""Shared drivers for the parity tests: run the same scene spec through the oracle (checker)
and through the product (librts_amd.so via ctypes) and compare."""
import numpy as np

from rts_amd import scenes


def oracle_scene(O, spec, motion=None):
    motion = motion if motion is not None else spec["motion"]
    sc = O.Scene()
    for m, mo in zip(spec["meshes"], motion):
        vw, nw = scenes.world_vertices(m, mo)
        sc.add_mesh(m["tris"], vw, nw, m["refl_coeff"], m["refr_index"], mo.get("velocity", (0, 0, 0)))
    sc.set_receivers(spec["rx"])
    return sc


def oracle_trace(O, spec, motion=None, **kw):
    sc = oracle_scene(O, spec, motion)
    tx = spec["tx"]
    return sc.trace(tx["origin"], tx["span"], tx["dir"], spec["W"], spec["max_refl"], spec.get("max_refr", 0), spec["smooth"], **kw)


def gpu_tracer(api, spec, **kw):
    tr = api.Tracer(spec["W"], spec["max_refl"], spec.get("max_refr", 0), spec["smooth"], **kw)
    tr.set_scene(spec["meshes"])
    tr.set_receivers(spec["rx"])
    return tr


def gpu_trace(api, spec, tr=None, motion=None, **kw):
    own = tr is None
    if own:
        tr = gpu_tracer(api, spec, keep_all=True)
    tx = spec["tx"]
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], motion if motion is not None else spec["motion"], **kw)   # kw: ray_first, ray_count, interleave
    return tr, st


PRD_EXACT_FIELDS = ["rayLength", "refrIndex", "reflDepth", "refrDepth", "maxRayIndex", "rayDirection", "firstHitPoint",
                    "prevHitPoint", "power", "doppler", "received", "end"]


def assert_prd_equal(a, b, what=""):
    """bit-exact comparison of PerRayData arrays, field by field (NaN-safe through byte views)."""
    assert a.shape == b.shape, (what, a.shape, b.shape)
    for f in PRD_EXACT_FIELDS:
        xa = np.ascontiguousarray(a[f]); xb = np.ascontiguousarray(b[f])
        if xa.dtype.kind == "f":
            same = xa.view(np.uint64) == xb.view(np.uint64)
        else:
            same = xa == xb
        if not np.all(same):
            bad = np.argwhere(~same)
            i = tuple(bad[0])
            raise AssertionError("%s: field %s differs at %d places, first at %s: %r vs %r" % (what, f, len(bad), i, xa[i], xb[i]))


def compare_full(o, g, n):
    """o: oracle trace dict (all rows), g: product rts_get_all_rays dict; n = number of output rows."""
    nh = o["hit_prim"].shape[0]
    assert np.array_equal(o["hit_prim"], g["hit_prim"][:nh]), "closest-hit primitive ids differ"
    assert np.array_equal(o["hit_t"].view(np.uint32), g["hit_t"][:nh].view(np.uint32)), "f32 hit distances differ"
    assert np.array_equal(o["path"][:n], g["path"][:n]), "target paths differ"
    assert_prd_equal(o["results"][:n], g["results"][:n], "per-ray records")
    # RCS angles come from libm / OCML atan2: tolerance, not bits
    np.testing.assert_allclose(g["rcs_angle"][:n], o["rcs_angle"][:n], rtol=0, atol=1e-12)
