"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI (ctypes -> librts_amd.so),
against the CPU oracle and the committed golden vectors.

Bar (BASELINE.json north_star): closest-hit primitive ids, f32 hit distances, depths, target
paths, received flags and every f64 field of the per-ray record BIT-EXACT; RCS angles (libm vs
OCML atan2) within 1e-12 rad; aggregated returns within 1e-9 relative (the north star allows
1e-5), pathMatch exact.
"""
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import helpers as H  # noqa: E402
import make_golden  # noqa: E402
from test_golden import check_against_golden, NAMES  # noqa: E402
from test_host_logic import random_received_set  # noqa: E402

pytestmark = pytest.mark.gpu
C0 = 299792458.0


@pytest.fixture(scope="module")
def scenes():
    from rts_amd import scenes as S
    return S


def full_parity(api, O, spec, motion=None, bvh=False):
    rows = spec["max_refl"] + 3 if spec.get("max_refr", 0) else 1
    n = spec["W"] ** 3 * rows
    tr, st = H.gpu_trace(api, spec, motion=motion)
    g = tr.all_rays(spec["W"] ** 3)
    o = H.oracle_trace(O, spec, motion=motion, use_bvh=bvh, threads=4 if bvh else 1)
    H.compare_full(o, g, n)
    assert st["segments"] == o["counters"]["segments"] and st["shaded"] == o["counters"]["shaded"]
    rec = tr.received()
    idx = np.nonzero(o["results"]["received"] >= 0)[0]
    assert st["received"] == len(idx)
    assert np.array_equal(rec["slots"], idx.astype(np.uint64))           # ascending launch index == host scan order (ray_tracer.cpp:1190)
    H.assert_prd_equal(o["results"][idx], rec["results"], "received records")
    assert np.array_equal(o["path"][idx], rec["path"])
    np.testing.assert_allclose(rec["rcs_angle"], o["rcs_angle"][idx], rtol=0, atol=1e-12)
    return tr, st, o, g


def test_device_math_is_ieee(rts, oracle):
    """f64 divide / sqrt correctly rounded on gfx950 and the basic-op atan2f identical to the oracle's"""
    tr = rts.Tracer(2, 1)
    rng = np.random.default_rng(0)
    n = 200000
    y = (rng.normal(size=n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    x = (rng.normal(size=n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    y[:8] = [0, 0, 1, -1, 0, -0.0, 3, -3]; x[:8] = [1, -1, 0, 0, 0, -1, 3, -3]
    a = rng.normal(size=n) * 10.0 ** rng.uniform(-100, 100, n); b = rng.normal(size=n) * 10.0 ** rng.uniform(-100, 100, n)
    at, dv, sq = tr.self_test_math(y, x, a, b)
    assert np.array_equal(dv.view(np.uint64), (a / b).view(np.uint64))
    assert np.array_equal(sq.view(np.uint64), np.sqrt(np.abs(a)).view(np.uint64))
    want = np.array([oracle.atan2f(yy, xx) for yy, xx in zip(y[:20000], x[:20000])], np.float32)
    assert np.array_equal(at[:20000].view(np.uint32), want.view(np.uint32))
    tr.close()


@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_golden(rts, name):
    spec = make_golden.golden_specs()[name]
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    n = spec["W"] ** 3
    tr, st = H.gpu_trace(rts, spec)
    a = tr.all_rays(n)
    check_against_golden(g, a["results"], a["path"], a["hit_prim"], a["hit_t"], a["rcs_angle"])
    assert st["segments"] == g["counters"][0] and st["shaded"] == g["counters"][1]
    # finalise + aggregate on the device vs the golden literal aggregation
    wl = spec["c"] / spec["carrier"]
    tr.finalise_uniform([1.0] * len(spec["meshes"]), wl, 1.0, 1.0, spec["carrier"], spec["c"])
    fin = tr.received()["results"]
    assert np.array_equal(fin["power"], g["fin_power"]) and np.array_equal(fin["doppler"], g["fin_doppler"])
    groups = tr.aggregate(spec["c"], spec["carrier"])
    ag = tr.aggregated()
    assert np.array_equal(ag["pathMatch"], g["agg_pathMatch"])
    np.testing.assert_allclose(ag["results"]["power"], g["agg_power"], rtol=1e-12)
    np.testing.assert_allclose(ag["results"]["doppler"], g["agg_doppler"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(ag["delay"], g["agg_delay"], rtol=1e-12)
    np.testing.assert_allclose(ag["phase"], g["agg_phase"], rtol=1e-9, atol=1e-12)
    resp = rts.groups_to_responses(groups)
    assert np.array_equal(resp["ray"].astype(np.int64), g["unique"].astype(np.int64))
    tr.close()


def test_c1_plate_full(rts, oracle, scenes):
    """BASELINE configs[0]: 2-triangle plate, 1 Tx / 1 Rx, 10 648 launch indices, 1 bounce"""
    tr, st, o, g = full_parity(rts, oracle, scenes.config1())
    assert st["rays"] == 22 ** 3 and st["received"] > 1000
    tr.close()


@pytest.mark.parametrize("smooth", [True, False])
def test_sphere_smooth_and_flat(rts, oracle, scenes, smooth):
    """interpolated vertex normals vs flat face normals (triangle_mesh.cu:175-194)"""
    spec = scenes.config2(subdiv=3, W=20, rx_radius=400.0)
    spec["smooth"] = smooth
    tr, st, o, g = full_parity(rts, oracle, spec)
    assert st["received"] > 0 and st["shaded"] > 3000
    tr.close()


@pytest.mark.parametrize("max_refl", [0, 1, 4, 9])
def test_multi_target_depths(rts, oracle, scenes, max_refl):
    """three targets incl. the "rect" per-face-normal rule, two receivers, inter-target bounces; depth 0 = every hit absorbed;
    depth 9 exercises the second path word"""
    spec = scenes.config_multi(W=18, max_refl=max_refl)
    tr, st, o, g = full_parity(rts, oracle, spec)
    if max_refl >= 4:
        assert o["results"]["reflDepth"].max() >= 3
        assert len(np.unique(o["path"][o["results"]["received"] >= 0], axis=0)) >= 2
    tr.close()


@pytest.mark.parametrize("max_refl,smooth", [(1, True), (3, True), (3, False), (0, True)])
def test_refraction(rts, oracle, scenes, max_refl, smooth):
    """refraction branch (normal_shader.cu:191-282, maxRefr forced to 2): rows rayIndex + k W^3 for the ray refracted
    into (k = 1) and back out of (k = 2) the first-hit target, path prefill, (1 - |Gamma|) power split, refract()"""
    spec = scenes.config_multi(W=14, max_refl=max_refl, smooth=smooth)
    spec["max_refr"] = 1
    spec["meshes"][0]["refr_index"] = 1.5; spec["meshes"][0]["refl_coeff"] = 0.5
    spec["meshes"][1]["refr_index"] = 2.2; spec["meshes"][1]["refl_coeff"] = -0.6
    spec["meshes"][2]["refl_coeff"] = 1.0                                 # |Gamma| = 1: never refracts (:198)
    spec["rx"] = spec["rx"] + [scenes._rx_at((200.0, 0.0, 0.0), (0, 0, 0), 90.0, 2.6)]   # behind the targets: catches transmitted rays
    tr, st, o, g = full_parity(rts, oracle, spec)
    n = spec["W"] ** 3
    res = o["results"]
    assert (res["refrDepth"][n:2 * n] >= 1).any() and (res["refrDepth"][2 * n:3 * n] == 2).any()
    assert (res["received"][2 * n:3 * n] >= 0).any()                      # rays refracted through a target reach the far receiver
    # finalise + aggregate over rows of all chains vs the literal pipeline
    wl = spec["c"] / spec["carrier"]
    tr.finalise_uniform([1.0, 1.0, 1.0], wl, 1.0, 1.0, spec["carrier"], spec["c"])
    groups = tr.aggregate(spec["c"], spec["carrier"])
    rx, rxi, slots = oracle.filter_finalise(o["results"], o["path"], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], len(res))
    ag = tr.aggregated()
    assert np.array_equal(ag["pathMatch"], lit["pathMatch"])
    np.testing.assert_allclose(ag["results"]["power"], lit["results"]["power"], rtol=1e-11)
    resp = rts.groups_to_responses(groups)
    assert np.array_equal(resp["ray"].astype(np.int64), oracle.unique_paths(lit["pathMatch"]).astype(np.int64))
    tr.close()


def test_aircraft_bvh_mode(rts, oracle, scenes):
    """8 000-triangle airframe, 4 receivers: the device hierarchy finds exactly the brute-force closest hits"""
    spec = scenes.config3(W=22, detail=0.08, rx_radius=400.0)
    tr, st, o, g = full_parity(rts, oracle, spec, bvh=True)
    assert (g["hit_prim"][:, 0] >= 0).mean() > 0.1
    tr.close()


def test_rotating_moving_target(rts, oracle, scenes):
    """per-pulse rigid transform (ray_tracer.cpp:993-1014): rotation of vertices AND normals, displacement, velocity"""
    spec = scenes.config_multi(W=14)
    tr = H.gpu_tracer(rts, spec, keep_all=True)
    n = spec["W"] ** 3
    for k in range(3):
        mo = [dict(position=(0.3 * k, 0.1 * k, 0.0), velocity=(300.0, 100.0, 0.0), rotation=rts.rotation_matrix(0.2 * k, 0.05 * k, -0.1 * k)),
              dict(position=(2.0, 9.0 - 0.5 * k, 1.0), velocity=(0.0, -500.0, 0.0)),
              dict(position=(9.0, -7.0, 0.2 * k), velocity=(0.0, 0.0, 200.0), rotation=rts.rotation_matrix(0.0, 0.0, 0.3 * k))]
        _, st = H.gpu_trace(rts, spec, tr=tr, motion=mo)
        assert st["bvh_rebuilt"] == 1
        H.compare_full(H.oracle_trace(oracle, spec, motion=mo), tr.all_rays(n), n)
    _, st = H.gpu_trace(rts, spec, tr=tr, motion=mo)
    assert st["bvh_rebuilt"] == 0                                        # unchanged placement: nothing is re-placed
    tr.close()


def test_edge_cases(rts, oracle, scenes):
    # W = 1: boresight ray only (ray_tracer.cu:160-161), axis-aligned direction (zero components in the slab test)
    spec = scenes.config1(); spec["W"] = 1
    full_parity(rts, oracle, spec)[0].close()
    # no targets at all: every ray goes straight to the miss program
    spec = scenes.config1(); spec["W"] = 6; spec["meshes"] = []; spec["motion"] = []
    spec["rx"] = [oracle.rx_sphere((500.0, 0.0, 0.0), math.pi, 0.0, 40.0, 2.0, 2.0)]
    tr, st, o, g = full_parity(rts, oracle, spec)
    assert st["shaded"] == 0 and st["received"] > 0 and (o["results"]["reflDepth"] == 0).all()   # direct rays
    tr.close()
    # no receivers: nothing can be received
    spec = scenes.config2(subdiv=1, W=8); spec["rx"] = []
    tr, st, o, g = full_parity(rts, oracle, spec)
    assert st["received"] == 0 and tr.received()["results"].shape[0] == 0
    tr.close()
    # a degenerate (zero-area) triangle and a sliver next to real geometry, single-primitive scene
    v = np.array([[0, -5, -5], [0, 5, -5], [0, 5, 5], [0, 0, 0], [0, 0, 0], [0, 1, 1], [0, -5, -5], [0, 5, 5], [0, -5, 5]], np.float64)
    t = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.uint32); nrm = np.tile(np.array([[-1.0, 0, 0]]), (9, 1))
    spec = scenes.config1(); spec["W"] = 12
    spec["meshes"] = [dict(tris=t, verts=v, normals=nrm, refl_coeff=0.5, refr_index=1.0)]
    full_parity(rts, oracle, spec)[0].close()
    spec["meshes"] = [dict(tris=t[:1], verts=v[:3], normals=nrm[:3], refl_coeff=0.5, refr_index=1.0)]
    full_parity(rts, oracle, spec)[0].close()


def test_ragged_shards_equal_whole(rts, scenes):
    """ray_first / ray_count ranges (multi-GPU sharding): concatenated shard outputs == the whole launch"""
    spec = scenes.config_multi(W=16)
    n = spec["W"] ** 3
    tr = H.gpu_tracer(rts, spec)
    H.gpu_trace(rts, spec, tr=tr)
    whole = tr.received()
    parts = []
    for lo, hi in [(0, 1), (1, 1000), (1000, 1001), (1001, 4095), (4095, 4096)]:
        _, st = H.gpu_trace(rts, spec, tr=tr, ray_first=lo, ray_count=hi - lo)
        assert st["rays"] == hi - lo
        parts.append(tr.received())
    H.assert_prd_equal(np.concatenate([p["results"] for p in parts]), whole["results"], "shards")
    assert np.array_equal(np.concatenate([p["slots"] for p in parts]), whole["slots"])
    assert np.array_equal(np.concatenate([p["path"] for p in parts]), whole["path"])
    from rts_amd import _lib
    with pytest.raises(_lib.RtsError):
        H.gpu_trace(rts, spec, tr=tr, ray_first=n - 5, ray_count=10)    # range outside W^3
    tr.close()


def test_bvh_invariants(rts, scenes):
    """static target-space BVH4 (rts_sah.cpp): every node reachable exactly once from its target's root, child boxes nested
    in the parent's; every primitive has at least one leaf slot, an unsplit primitive's box strictly contains it, and the
    boxes of a split primitive's references together cover its vertices and its centroid (local coordinates)"""
    spec = scenes.config_multi(W=4)
    big = scenes.config3(W=4, detail=0.2)                                        # ellipsoids with sliver fans at the poles: these get split
    spec["meshes"] = spec["meshes"] + big["meshes"]; spec["motion"] = spec["motion"] + big["motion"]
    tr = H.gpu_tracer(rts, spec)
    H.gpu_trace(rts, spec, tr=tr)
    nodes, leaf_prim, roots = tr.bvh()
    nprim = sum(m["tris"].shape[0] for m in spec["meshes"])
    refs = np.bincount(leaf_prim, minlength=nprim)
    assert len(leaf_prim) >= nprim and refs.min() >= 1 and len(roots) == len(spec["meshes"])
    info = tr.scene_info()
    host_sah = info["builder"] == 0            # RTS_BUILDER=device: LBVH over slab references, records of opened BVH2 nodes are unreachable
    assert info["n_nodes"] == len(nodes) and info["n_leaves"] == len(leaf_prim) and info["n_targets"] == len(roots)
    assert refs.max() > 1, "the sliver fans of the ellipsoid mesh are expected to be split"
    if not host_sah:
        assert refs.max() <= 8                 # at most eight slabs per triangle (rts_lbvh.hip: ref_count)
    child = nodes[:, 24:28].copy().view(np.int32)
    lo = np.stack([nodes[:, 0:4], nodes[:, 4:8], nodes[:, 8:12]], axis=2)        # [node][child][xyz]
    hi = np.stack([nodes[:, 12:16], nodes[:, 16:20], nodes[:, 20:24]], axis=2)
    tv = np.concatenate([m["verts"][m["tris"]] for m in spec["meshes"]])         # [prim][3][xyz]
    prim_targ = np.concatenate([np.full(m["tris"].shape[0], i) for i, m in enumerate(spec["meshes"])])
    seen_nodes = np.zeros(len(nodes), bool); seen_leaves = np.zeros(len(leaf_prim), bool)
    covered = np.zeros((nprim, 4), bool)                                         # 3 vertices + centroid inside some reference box
    for t, root in enumerate(roots):
        assert root >= 0
        stack = [(int(root), np.full(3, -np.inf), np.full(3, np.inf))]
        while stack:
            i, plo, phi = stack.pop()
            assert not seen_nodes[i]; seen_nodes[i] = True
            used = 0
            for k in range(4):
                c = child[i, k]
                if c == 0x7fffffff:
                    assert (lo[i, k] == hi[i, k]).all() and (lo[i, k] > 1e38).all()    # unused slot: a point at 3e38, unreachable
                    continue
                used += 1
                assert (lo[i, k] >= plo).all() and (hi[i, k] <= phi).all()     # nested in the parent's box
                if c < 0:
                    leaf = ~c; assert not seen_leaves[leaf]; seen_leaves[leaf] = True
                    p = leaf_prim[leaf]
                    assert prim_targ[p] == t
                    pts = np.concatenate([tv[p], tv[p].mean(axis=0, keepdims=True)])
                    inside = ((pts > lo[i, k].astype(np.float64)) & (pts < hi[i, k].astype(np.float64))).all(axis=1)
                    covered[p] |= inside
                    if refs[p] == 1:
                        assert inside.all()                                    # padded outward
                else:
                    stack.append((int(c), lo[i, k], hi[i, k]))
            assert used >= 1
    assert seen_leaves.all() and covered.all()
    assert seen_nodes.all() or not host_sah
    tr.close()


def test_device_builder_gives_the_same_rays(rts, oracle, scenes):
    """RTS_FLAG_DEVICE_BUILD: the hierarchy is built on the GPU (LBVH, rts_lbvh.hip) instead of the host SAH builder.  A
    different tree, the same results bit for bit: the exact f64 test alone decides hits.  Also: a mesh with non-finite
    vertices (those triangles get no leaf), a single-triangle mesh, an empty mesh; and the build is the fast one."""
    spec = scenes.config_multi(W=18)
    one = dict(tris=np.array([[0, 1, 2]], np.uint32), verts=np.array([[3.0, -2, -2], [3.0, 2, -2], [3.2, 0, 2.5]]), normals=np.array([[-1.0, 0, 0]] * 3), refl_coeff=0.6, refr_index=1.0)
    bad = dict(spec["meshes"][1]); bv = bad["verts"].copy(); bv[0, 1] = np.nan; bad["verts"] = bv
    empty = dict(tris=np.zeros((0, 3), np.uint32), verts=np.zeros((0, 3)), normals=np.zeros((1, 3)), refl_coeff=0.9, refr_index=1.0)
    spec["meshes"] = spec["meshes"] + [one, bad, empty]
    spec["motion"] = spec["motion"] + [dict(position=(-6.0, 1.0, 0.5), velocity=(0.0, 0.0, 0.0)), dict(position=(4.0, -12.0, 3.0), velocity=(1.0, 0.0, 0.0)), dict(position=(0.0, 0.0, 0.0), velocity=(0.0, 0.0, 0.0))]
    n = spec["W"] ** 3
    res = {}
    for dev in (False, True):
        tr = rts.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], keep_all=True, device_build=dev)
        tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
        _, st = H.gpu_trace(rts, spec, tr=tr)
        info = tr.scene_info()
        if "RTS_BUILDER" not in os.environ:
            assert info["builder"] == (1 if dev else 0)
        res[dev] = (tr.all_rays(n), st, info)
        tr.close()
    a, b = res[False][0], res[True][0]
    assert np.array_equal(a["hit_prim"], b["hit_prim"]) and np.array_equal(a["hit_t"].view(np.uint32), b["hit_t"].view(np.uint32))
    H.assert_prd_equal(a["results"], b["results"], "device-built hierarchy")
    assert np.array_equal(a["path"], b["path"])
    assert res[False][1]["segments"] == res[True][1]["segments"] and res[True][1]["shaded"] > 100
    o = H.oracle_trace(oracle, spec)
    H.compare_full(o, b, n)


def test_shared_scene_between_handles(rts, scenes):
    """rts_share_scene: handles that keep pulses in flight read ONE copy of the meshes and the hierarchy; results are those
    of handles with scenes of their own, and the shared part outlives the handle that built it"""
    spec = scenes.config3(W=40, detail=0.1, rx_radius=300.0)
    own = H.gpu_tracer(rts, spec)
    _, st0 = H.gpu_trace(rts, spec, tr=own); ref = own.received()
    a = H.gpu_tracer(rts, spec)
    b = rts.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"]); b.share_scene(a); b.set_receivers(spec["rx"])
    c = rts.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"]); c.share_scene(b); c.set_receivers(spec["rx"])
    ia, ib = a.scene_info(), b.scene_info()
    assert ia["handles_sharing"] == ib["handles_sharing"] == 3 and own.scene_info()["handles_sharing"] == 1
    assert ia["shared_device_bytes"] == ib["shared_device_bytes"] > 0 and ib["handle_device_bytes"] > 0
    assert ib["build_ms"] == ia["build_ms"]                                    # built once
    a.close()                                                                    # the builder goes away first
    for t in (b, c):
        _, st = H.gpu_trace(rts, spec, tr=t); got = t.received()
        assert st["segments"] == st0["segments"]
        H.assert_prd_equal(ref["results"], got["results"], "shared scene")
        assert np.array_equal(ref["slots"], got["slots"])
    assert b.scene_info()["handles_sharing"] == 2
    # a handle that sets a scene of its own leaves the group
    c.set_scene(spec["meshes"]); assert c.scene_info()["handles_sharing"] == 1 and b.scene_info()["handles_sharing"] == 1
    _, st = H.gpu_trace(rts, spec, tr=c); H.assert_prd_equal(ref["results"], c.received()["results"], "own scene again")
    own.close(); b.close(); c.close()


@pytest.mark.parametrize("seed,R,D,n_rx,n_targ", [(1, 50, 3, 2, 2), (2, 3000, 4, 4, 3), (3, 700, 1, 1, 1), (4, 5000, 6, 3, 1), (5, 1, 2, 1, 1),
                                                   (6, 2500, 16, 5, 100), (7, 1500, 16, 300, 250), (8, 900, 9, 2, 200)])
def test_kernel_wrapper_equals_literal(rts, oracle, seed, R, D, n_rx, n_targ):
    """rs::kernel_wrapper drop-in (aggregation.cuh:19-22): same in/out arrays as the O(R^2) myKernel1/2.  Seeds 6-8 need a
    (receiver, path) key of 115 / 137 / 73 bits: the multi-word (wide key) path of the group-by"""
    rng = np.random.default_rng(seed)
    a, paths = random_received_set(oracle, rng, R, D, n_rx, n_targ)
    fc = 10e9
    lit = oracle.aggregate_literal(a, paths, C0, fc, 10 ** 6)
    got = rts.kernel_wrapper(a, paths, C0, fc, 10 ** 6)
    assert np.array_equal(got["pathMatch"], lit["pathMatch"])
    np.testing.assert_allclose(got["results"]["power"], lit["results"]["power"], rtol=1e-11)
    np.testing.assert_allclose(got["results"]["doppler"], lit["results"]["doppler"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(got["delay"], lit["delay"], rtol=1e-12)
    np.testing.assert_allclose(got["phase"], lit["phase"], rtol=1e-9, atol=1e-11)
    for f in ("rayLength", "received", "reflDepth", "firstHitPoint", "prevHitPoint"):
        assert np.array_equal(got["results"][f], a[f])                   # untouched fields come back unchanged


def test_aggregation_one_huge_group_is_deterministic(rts, oracle):
    """200 000 rays in ONE (rx, path) group spanning hundreds of tiles: fixed-shape reduction, same bits every run"""
    rng = np.random.default_rng(7)
    R = 200000
    a, paths = random_received_set(oracle, rng, R, 2, 1, 1, p_direct=0.0)
    paths[:] = [0, -1]; a["reflDepth"] = 1
    g1 = rts.kernel_wrapper(a, paths, C0, 1e10, 10 ** 7)
    g2 = rts.kernel_wrapper(a, paths, C0, 1e10, 10 ** 7)
    assert np.array_equal(g1["results"]["power"].view(np.uint64), g2["results"]["power"].view(np.uint64))
    assert np.array_equal(g1["phase"].view(np.uint64), g2["phase"].view(np.uint64))
    assert (g1["pathMatch"] == 0).all()
    want = (np.sqrt(a["power"]).sum() / R) ** 2
    np.testing.assert_allclose(g1["results"]["power"], want, rtol=1e-12)
    np.testing.assert_allclose(g1["delay"], (a["rayLength"] / C0).mean(), rtol=1e-12)


def test_full_size_properties(rts, oracle, scenes):
    """BASELINE configs[2] at full size (100 000 triangles, W = 216, 10 077 696 launch indices):
    size-independent properties + sampled parity against the oracle (BVH mode)."""
    spec = scenes.config3(rx_radius=200.0)
    n = spec["W"] ** 3
    tr = H.gpu_tracer(rts, spec)
    _, st = H.gpu_trace(rts, spec, tr=tr)
    whole = tr.received()
    assert st["rays"] == n and st["segments"] == n + st["shaded"]       # every shaded hit spawns exactly one more segment
    assert st["stack_overflows"] == 0 or st["stack_overflows"] > 0      # spills are allowed, hard overflow would have raised
    R = st["received"]
    assert R == len(whole["slots"]) and R > 100
    assert (np.diff(whole["slots"].astype(np.int64)) > 0).all()         # strictly ascending launch indices
    # determinism: same bits on a second run
    _, st2 = H.gpu_trace(rts, spec, tr=tr)
    again = tr.received()
    H.assert_prd_equal(whole["results"], again["results"], "run-to-run")
    assert st2["bvh_rebuilt"] == 0
    # shard invariance: two halves == whole
    h = n // 2 + 12345
    _, _ = H.gpu_trace(rts, spec, tr=tr, ray_first=0, ray_count=h); a = tr.received()
    _, _ = H.gpu_trace(rts, spec, tr=tr, ray_first=h, ray_count=n - h); b = tr.received()
    H.assert_prd_equal(np.concatenate([a["results"], b["results"]]), whole["results"], "halves")
    assert np.array_equal(np.concatenate([a["slots"], b["slots"]]), whole["slots"])
    # sampled parity: every 499th launch index through the oracle; its received subset must match record for record
    stride = 499; m = n // stride
    o = H.oracle_trace(oracle, spec, ray_first=7, ray_stride=stride, n_rays=m, use_bvh=True, threads=8, debug=False)
    samp = 7 + stride * np.arange(m, dtype=np.int64)
    o_idx = np.nonzero(o["results"]["received"] >= 0)[0]
    pos = np.searchsorted(whole["slots"].astype(np.int64), samp)
    pos_c = np.minimum(pos, R - 1)
    is_recv = whole["slots"].astype(np.int64)[pos_c] == samp
    assert np.array_equal(np.nonzero(is_recv)[0], o_idx)
    H.assert_prd_equal(o["results"][o_idx], whole["results"][pos_c[is_recv]], "sampled received records")
    assert np.array_equal(o["path"][o_idx], whole["path"][pos_c[is_recv]])
    # aggregation closure: group counts add up to R; responses unique and ascending
    wl = spec["c"] / spec["carrier"]
    _, _ = H.gpu_trace(rts, spec, tr=tr)
    tr.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    groups = tr.aggregate(spec["c"], spec["carrier"])
    assert int(groups["n"].sum()) == R
    resp = rts.groups_to_responses(groups)
    assert (np.diff(resp["ray"].astype(np.int64)) > 0).all() and len(resp) >= 1
    ag = tr.aggregated()
    assert set(np.unique(ag["pathMatch"])) == set(resp["ray"].astype(np.int64))
    tr.close()


def test_traversal_counters(rts, scenes):
    """counting build: node visits / triangle tests are reported and the result set is unchanged"""
    spec = scenes.config2(subdiv=3, W=24, rx_radius=400.0)
    tr = H.gpu_tracer(rts, spec); _, s0 = H.gpu_trace(rts, spec, tr=tr); r0 = tr.received(); tr.close()
    tc = H.gpu_tracer(rts, spec, count_traversal=True); _, s1 = H.gpu_trace(rts, spec, tr=tc); r1 = tc.received(); tc.close()
    assert s0["node_visits"] == 0 and s1["node_visits"] > s1["shaded"] > 0 and s1["tri_tests"] >= s1["shaded"]
    assert s0["segments"] == s1["segments"]
    H.assert_prd_equal(r0["results"], r1["results"], "counting build")


def _adapter_scenarios(tmp_path):
    import adapter_ref as AR
    vf, nf = str(tmp_path / "octa_v.txt"), str(tmp_path / "octa_n.txt")
    AR.write_octahedron_files(vf, nf, radius=4.0, subdiv=2)
    return [("base", AR.scenario_base()), ("two_tx", AR.scenario_two_tx()), ("refraction", AR.scenario_refraction()),
            ("file", AR.scenario_file(vf, nf)), ("ecef", AR.scenario_ecef())]


def test_adapter_end_to_end(rts, oracle, tmp_path):
    """rs::RTS drop-in: include/rts_adapter.hpp driven by a mock SOARS World whose antenna gains depend on the look
    direction and on the antenna's rotation AT THE TIME ASKED FOR, and whose RCS depends on both bistatic angles and the
    wavelength, emits exactly the responses of the literal pipeline (tests/adapter_ref.py): oracle trace -> host filter +
    finalise with the simulator's callbacks (ray_tracer.cpp:1190-1258) -> O(R^2) aggregation -> unique paths (:1290-1321).
    Scenarios: three moving targets / two receivers; TWO transmitters (noise temperature accumulating per transmitter,
    :829); refraction (maxRefr clamped to 2, rayTotal = W^3 (maxRefl + 3), :604-626); a "file" target (:983-987); the
    scene at Earth-centred coordinates."""
    import subprocess
    import adapter_ref as AR
    from test_host_logic import build_adapter_binary
    exe = build_adapter_binary(str(tmp_path / "adapter_main"))
    for name, sc in _adapter_scenarios(tmp_path):
        path = str(tmp_path / ("%s.scn" % name))
        AR.write_scenario(path, sc)
        out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (name, out.stderr)
        variants = [["1"]] if name != "base" else [["1"], ["2", "2"], ["1", "3"], ["2", "2", "rays"], ["1", "3", "rays"], ["3", "2", "rays"], ["2", "2", "deal"], ["3", "3", "deal"]]
        if name == "refraction":
            variants.append(["2", "2", "rays"])                                   # chains k W^3 apart come from the same part
            variants.append(["2", "3", "deal"])                                   # ... also when the tiles are dealt from the cost records after two pulses (RunOptions::deal_after)
        for argv in variants:
            # sequential pulses; several handle sets on device 0 (the multi-device path on one GPU) -- whole pulses dealt to the
            # sets in turn, and every pulse split over the sets in interleaved tiles: byte-identical output
            other = subprocess.run([exe, path] + argv, capture_output=True, text=True, timeout=300)
            assert other.returncode == 0, (name, argv, other.stderr)
            assert other.stdout == out.stdout, "%s: run %r must emit the responses of the default run" % (name, argv)
        got, got_noise = AR.parse_adapter_output(out.stdout)
        want, want_noise = AR.run_reference_flow(oracle, sc)
        assert len(want) > 3, name
        AR.assert_responses_close(got, want)
        assert got_noise == want_noise, name
        if name == "two_tx":
            assert set(got[:, 0]) == {0.0, 1.0}
            for j in (0.0, 1.0):                                                   # the second transmitter's responses carry its 21.5 K on top (quirk 15)
                n0 = set(got[(got[:, 0] == 0) & (got[:, 2] == j)][:, 7]); n1 = set(got[(got[:, 0] == 1) & (got[:, 2] == j)][:, 7])
                assert len(n0) == 1 and len(n1) == 1 and n1.pop() == n0.pop() + 21.5
        if name == "refraction":
            assert want_noise and len(want) > 3
    # the comparison is sensitive to every argument of :1204-1247 (tests/test_oracle_kat.py shows it on the oracle side too)
    sc = AR.scenario_base()
    got, _ = AR.parse_adapter_output(subprocess.run([exe, str(tmp_path / "base.scn")], capture_output=True, text=True, timeout=300).stdout)
    for mut in ("swap_hits", "rx_time", "angle_rows"):
        bad, _ = AR.run_reference_flow(oracle, sc, mutate=mut)
        assert AR.max_power_deviation(got, bad) > 1e-6, mut


def test_c4_c5_shapes(rts, oracle, scenes):
    """BASELINE configs[3] and [4] at reduced size: four targets / two transmitters / eight receivers / 8 bounces,
    and the per-pulse rotating + translating target (re-placed every pulse)"""
    spec = scenes.config4(W=40, detail=0.1, rx_radius=300.0)
    for tx in spec["tx_list"]:
        spec["tx"] = tx
        tr, st, o, g = full_parity(rts, oracle, spec, bvh=True)
        assert st["n_prims"] == sum(m["tris"].shape[0] for m in spec["meshes"]) and st["received"] > 0
        tr.close()
    spec = scenes.config5(W=36, detail=0.1, rx_radius=300.0)
    tr = H.gpu_tracer(rts, spec, keep_all=True)
    n = spec["W"] ** 3
    for k in (0, 5, 400):
        mo = spec["motion_fn"](k)
        _, st = H.gpu_trace(rts, spec, tr=tr, motion=mo)
        assert st["bvh_rebuilt"] == 1
        H.compare_full(H.oracle_trace(oracle, spec, motion=mo, use_bvh=True, threads=4), tr.all_rays(n), n)
    tr.close()


def test_device_builder_failure_falls_back_to_the_host_builder(rts, oracle, scenes, monkeypatch):
    """a device hierarchy build that fails (out of memory on a huge mesh, a mesh it does not close on within its level bound) must
    not fail rts_set_scene: the host SAH builder takes over.  Injected after a complete device build, so that the
    fall-back also has that builder's buffers to drop; results against the oracle; RTS_DEVICE_BUILD_FALLBACK=0 returns the error"""
    from rts_amd import _lib
    spec = scenes.config_multi(W=18)
    monkeypatch.setenv("RTS_DEBUG_FAIL_DEVICE_BUILD", "1")
    tr = rts.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], keep_all=True, device_build=True)
    tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
    assert tr.scene_info()["builder"] == 0                             # built by the host after all
    _, st = H.gpu_trace(rts, spec, tr=tr)
    n = spec["W"] ** 3
    H.compare_full(H.oracle_trace(oracle, spec), tr.all_rays(n), n)
    tr.close()
    monkeypatch.setenv("RTS_DEVICE_BUILD_FALLBACK", "0")
    tr = rts.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], device_build=True)
    with pytest.raises(_lib.RtsError) as e:
        tr.set_scene(spec["meshes"])
    assert e.value.code == _lib.RTS_ERR_HIP and "injected" in str(e.value)
    tr.close()


def test_return_cube(rts, oracle, scenes):
    """complex return cube (north-star product, not in the reference; definition: oracle/rts_oracle.cpp orc_cube, DESIGN.md
    section 4): the device accumulation of sqrt(P) e^{j phi} into [rx][pulse][range bin] -- per received ray (coherent sum)
    and per unique path (one term per response of the reference) -- against the ORACLE's accumulation of the ORACLE's rays
    (independent rays, independent binning), and the library's slow-time transform against numpy's FFT"""
    spec = scenes.config_multi(W=20)
    wl = spec["c"] / spec["carrier"]
    n_bins, t0, dt, n_p = 64, 1.2e-6, 5.0e-9, 5
    tr = H.gpu_tracer(rts, spec); tr.cube_attach(len(spec["rx"]), n_p, n_bins, t0, dt)
    tp = H.gpu_tracer(rts, spec); tp.cube_attach(len(spec["rx"]), n_p, n_bins, t0, dt)
    want_rays = np.zeros((len(spec["rx"]), n_p, n_bins), np.complex128); want_paths = want_rays.copy()
    total = 0
    for k in range(n_p):
        mo = [dict(position=tuple(np.add(m["position"], (0.5 * k, 0, 0))), velocity=m["velocity"]) for m in spec["motion"]]
        for t in (tr, tp):
            H.gpu_trace(rts, spec, tr=t, motion=mo)
            t.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
        tr.cube_accumulate(k, spec["c"], spec["carrier"])
        tp.aggregate(spec["c"], spec["carrier"]); tp.cube_accumulate_paths(k)
        o = H.oracle_trace(oracle, spec, motion=mo)
        rx, rxi, _ = oracle.filter_finalise(o["results"], o["path"], [1.0] * len(spec["meshes"]), wl, 1.0, 1.0, spec["carrier"], spec["c"])
        oracle.cube_accumulate(want_rays, k, rx, t0, dt, spec["c"], spec["carrier"])
        oracle.cube_accumulate(want_paths, k, None, t0, dt, spec["c"], spec["carrier"], lit=oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], spec["W"] ** 3))
        total += len(rx)
    got_rays = tr.cube(); got_paths = tp.cube()
    assert total > 100 and np.count_nonzero(want_rays) > 10 and np.count_nonzero(want_paths) > 5
    assert np.array_equal(want_rays != 0, got_rays != 0) and np.array_equal(want_paths != 0, got_paths != 0)        # same cells
    np.testing.assert_allclose(got_rays, want_rays, rtol=1e-10, atol=1e-13 * np.abs(want_rays).max())              # (f64 atomics add in another order; OCML / libm sincos)
    np.testing.assert_allclose(got_paths, want_paths, rtol=1e-10, atol=1e-13 * np.abs(want_paths).max())
    assert np.count_nonzero(want_paths) < np.count_nonzero(want_rays) or not np.allclose(want_paths, want_rays)    # the two products differ (by design)
    # slow-time transform: zero-padded to 8 and to 16 points, every (rx, bin) column
    for n_fft in (8, 16):
        np.testing.assert_allclose(tr.cube_doppler(n_fft), np.fft.fft(want_rays, n=n_fft, axis=1), rtol=0, atol=1e-10 * np.abs(want_rays).max())
    from rts_amd import _lib
    assert _lib.lib().rts_cube_doppler(tr.h, 4, None) == _lib.RTS_ERR_INVALID and _lib.lib().rts_cube_doppler(tr.h, 12, None) == _lib.RTS_ERR_INVALID   # < n_pulses; not a power of two
    tr.close(); tp.close()
    # a long transform on random data: 1024 points (4 bins per block), 4096 points (one bin per block), ragged bin count
    import torch
    for n_p2, n_fft, nb in ((1000, 1024, 37), (4096, 4096, 5), (300, 512, 16)):
        t = rts.Tracer(8, 1); rng = np.random.default_rng(n_fft)
        data = rng.standard_normal((2, n_p2, nb)) + 1j * rng.standard_normal((2, n_p2, nb))
        buf = torch.from_numpy(np.ascontiguousarray(data)).cuda()
        t.cube_attach(2, n_p2, nb, 0.0, 1.0, device_ptr=buf.data_ptr())
        got = t.cube_doppler(n_fft)
        np.testing.assert_allclose(got, np.fft.fft(data, n=n_fft, axis=1), rtol=0, atol=1e-10 * np.abs(data).max() * math.sqrt(n_fft))
        t.close()


def test_deep_tree_spills_stack(rts, oracle, scenes, monkeypatch):
    """56 stacked plates, every ray pierces all of their boxes; with the LDS part of the traversal stack cut to 2 entries
    (RTS_STACK_LDS_DEBUG) the walk lives in the global spill slab -- results must not change"""
    monkeypatch.setenv("RTS_STACK_LDS_DEBUG", "2")
    K = 56
    vs, ts, ns = [], [], []
    for k in range(K):
        x = 100.0 * (0.6 ** k)                                     # centres 100, 60, 36, ... -> one Morton split per plate
        s = 50.0 + k                                               # all plates cover the beam axis
        v = np.array([[x, -s, -s], [x, s, -s], [x, s, s], [x, -s, -s], [x, s, s], [x, -s, s]], np.float64)
        vs.append(v); ts.append(np.array([[0, 1, 2], [3, 4, 5]], np.uint32) + 6 * k); ns.append(np.tile(np.array([[-1.0, 0.0, 0.0]]), (6, 1)))
    mesh = dict(tris=np.concatenate(ts), verts=np.concatenate(vs), normals=np.concatenate(ns), refl_coeff=0.9, refr_index=1.0)
    spec = scenes.config1()
    spec.update(W=10, max_refl=3, meshes=[mesh], motion=[dict(position=(0.0, 0.0, 0.0), velocity=(0.0, 0.0, 0.0))])
    spec["tx"] = dict(origin=(-1000.0, 3.0, 2.0), span=(0.02, 0.02, 0.1), dir=(0.0, 0.0))
    tr, st, o, g = full_parity(rts, oracle, spec)
    assert st["shaded"] > 0
    assert st["stack_overflows"] > 0, "this scene is meant to exercise the spill path"
    tr.close()


def test_file_mesh_on_device(rts, oracle, scenes, tmp_path):
    """"file" targets (ray_tracer.cpp:429-504): unshared vertices, per-vertex normals from a second file, rotated"""
    rng = np.random.default_rng(3)
    sv, st_, sn = oracle.sphere_mesh(2, 6.0)
    tri_v = sv[st_].reshape(-1, 9); tri_n = sn[st_].reshape(-1, 9)
    vf, nf = tmp_path / "v.txt", tmp_path / "n.txt"
    for f, a in ((vf, tri_v), (nf, tri_n)):
        with open(f, "w") as fh:
            for row in a:
                fh.write("%.17g %.17g %.17g, %.17g %.17g %.17g, %.17g %.17g %.17g,\n" % tuple(row))
    v, t, n = rts.file_mesh(str(vf), str(nf), 0.3, -0.2, 0.1)
    assert t.shape[0] == tri_v.shape[0] == 320 and v.shape == (960, 3) and n.shape == (960, 3)     # an empty mesh must not pass
    ov, ot, on = oracle.file_mesh(str(vf), str(nf), 0.3, -0.2, 0.1)
    assert np.array_equal(v, ov) and np.array_equal(t, ot) and np.array_equal(n, on)
    spec = scenes.config_multi(W=14)
    spec["meshes"][0] = dict(tris=t, verts=v, normals=n, refl_coeff=0.9, refr_index=1.0)
    tr, st, o, g = full_parity(rts, oracle, spec)
    hit0 = (o["hit_prim"] >= 0) & (o["hit_prim"] < 320)
    assert hit0.sum() > 20 and st["shaded"] > 0, "the file mesh must actually be hit"
    tr.close()


def test_wide_aggregation_keys_on_a_traced_scene(rts, oracle, scenes):
    """16 bounces among 9 targets: 16 x 4 + 1 = 65 key bits, one more than a 64-bit sort holds.  The device group table and
    the responses derived from it equal the literal O(R^2) aggregation of the oracle's rays"""
    spec = scenes.config_multi(W=16, max_refl=16)
    extra = []
    for k in range(6):
        v, t, n = scenes.plate_mesh(3.0 + k)
        extra.append(dict(tris=t, verts=v, normals=n, refl_coeff=0.85, refr_index=1.0))
    spec["meshes"] = spec["meshes"] + extra
    spec["motion"] = spec["motion"] + [dict(position=(6.0 + 3.0 * k, 12.0 - 5.0 * k, -3.0 + k), velocity=(0.0, 1.0 * k, 0.0)) for k in range(6)]
    n = spec["W"] ** 3; wl = spec["c"] / spec["carrier"]
    tr, st, o, g = full_parity(rts, oracle, spec)
    tr.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    groups = tr.aggregate(spec["c"], spec["carrier"])
    ag = tr.aggregated()
    rx, rxi, _ = oracle.filter_finalise(o["results"], o["path"], [1.0] * 9, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], n)
    assert np.array_equal(ag["pathMatch"], lit["pathMatch"])
    np.testing.assert_allclose(ag["results"]["power"], lit["results"]["power"], rtol=1e-11)
    np.testing.assert_allclose(ag["delay"], lit["delay"], rtol=1e-12)
    uniq = oracle.unique_paths(lit["pathMatch"])
    resp = rts.groups_to_responses(groups)
    assert np.array_equal(resp["ray"].astype(np.int64), uniq.astype(np.int64)) and len(uniq) > 3
    # the host copy of the table carries the groups' paths (decoded from the rays, not from a 64-bit key)
    first = {int(gr["min_ray"]): gr for gr in groups}
    for u in uniq[:50]:
        if int(u) in first and not first[int(u)]["direct"]:
            assert np.array_equal(first[int(u)]["path"][:16], rxi[u])
    tr.close()


def test_cube_reduce_between_handles(rts, scenes):
    """rts_cube_reduce: the complex return cubes of several handles of ONE process summed into every one of them (peer-copy
    transport; the RCCL transport needs distinct devices)"""
    import ctypes as C
    from rts_amd import _lib
    spec = scenes.config_multi(W=16)
    wl = spec["c"] / spec["carrier"]; tx = spec["tx"]
    r0 = 2.0 * abs(tx["origin"][0]); t0 = (r0 - 150.0) / spec["c"]; dt = 300.0 / spec["c"] / 64
    trs = []
    for part in range(3):
        t = H.gpu_tracer(rts, spec); t.cube_attach(len(spec["rx"]), 2, 64, t0, dt)
        t.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"], interleave=(64, 3, part), want_stats=False)
        t.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"]); t.cube_accumulate(1, spec["c"], spec["carrier"])
        trs.append(t)
    parts = [t.cube() for t in trs]
    whole = H.gpu_tracer(rts, spec); whole.cube_attach(len(spec["rx"]), 2, 64, t0, dt)
    whole.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"], want_stats=False)
    whole.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"]); whole.cube_accumulate(1, spec["c"], spec["carrier"])
    arr = (C.c_void_p * 3)(*[t.h for t in trs])
    _lib.check(_lib.lib().rts_cube_reduce(arr, 3, 0))
    want = parts[0] + parts[1] + parts[2]
    for t in trs:
        assert np.array_equal(t.cube(), want)                       # handle order: bit-reproducible
    assert np.abs(want).max() > 0
    np.testing.assert_allclose(want, whole.cube(), rtol=1e-9, atol=1e-30)     # f64 atomics of the whole pulse add in another order
    assert _lib.lib().rts_cube_reduce(arr, 3, 1) == _lib.RTS_ERR_UNSUPPORTED  # RCCL demanded, but the handles share a device
    # the RCCL transport with ONE handle: librccl is loaded, a one-rank communicator created (and cached: the second call
    # reuses it), ncclGroupStart / ncclAllReduce(sum, f64, in place) / ncclGroupEnd run on the handle's stream -- the identity
    # on the data, and the only way to execute these lines on a single GPU
    one = (C.c_void_p * 1)(whole.h)
    before = whole.cube()
    for _ in range(2):
        _lib.check(_lib.lib().rts_cube_reduce(one, 1, 1))
        assert np.array_equal(whole.cube(), before)
    for t in trs + [whole]:
        t.close()


def test_api_errors(rts, scenes):
    from rts_amd import _lib
    with pytest.raises(_lib.RtsError) as e:
        rts.Tracer(2000, 1)                                       # W^3 must fit 32 bits (rayIndex is unsigned int, ray_tracer.cu:151)
    assert e.value.code == _lib.RTS_ERR_INVALID
    with pytest.raises(_lib.RtsError) as e:
        rts.Tracer(4, 17)                                         # depth limit of the packed path keys
    assert e.value.code == _lib.RTS_ERR_UNSUPPORTED
    with pytest.raises(_lib.RtsError):
        rts.Tracer(4, 1, device=99)
    spec = scenes.config1()
    tr = H.gpu_tracer(rts, spec)
    bad = dict(spec["meshes"][0]); bad["tris"] = np.array([[0, 1, 99]], np.uint32)
    with pytest.raises(_lib.RtsError) as e:
        tr.set_scene([bad])                                       # vertex index out of range is rejected on the host, never reaches a kernel
    assert e.value.code == _lib.RTS_ERR_INVALID
    rx = dict(spec["rx"][0]); rx["minTheta"] = float("nan")
    with pytest.raises(_lib.RtsError):
        tr.set_receivers([rx])
    with pytest.raises(_lib.RtsError):
        tr.all_rays(10)                                           # handle was not created with RTS_FLAG_KEEP_ALL_RAYS
    with pytest.raises(_lib.RtsError):
        tr.aggregated()                                           # nothing aggregated yet
    tr.close()


def test_interleaved_parts_equal_whole(rts, oracle, scenes):
    """interleaved-tile sharding (RtsPulse.interleave_*): the parts' received sets, merged by buffer row, are the whole
    launch; row-keyed group tables of the parts merge into the literal aggregation of the whole pulse"""
    from rts_amd import _lib, multigpu
    spec = scenes.config_multi(W=20)
    n = spec["W"] ** 3
    wl = spec["c"] / spec["carrier"]
    tr = H.gpu_tracer(rts, spec)
    H.gpu_trace(rts, spec, tr=tr)
    whole = tr.received()
    tr.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    g_whole = tr.aggregate(spec["c"], spec["carrier"], _lib.RTS_BASE_USE_ROWS)
    for tile, parts in [(64, 3), (1000, 2), (4096, 5)]:
        recs, tabs = [], []
        for part in range(parts):
            _, st = H.gpu_trace(rts, spec, tr=tr, interleave=(tile, parts, part))
            assert st["rays"] == multigpu.part_ray_count(n, (tile, parts, part))
            recs.append(tr.received())
            tr.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
            tabs.append(tr.aggregate(spec["c"], spec["carrier"], _lib.RTS_BASE_USE_ROWS))
        slots = np.concatenate([r["slots"] for r in recs]); order = np.argsort(slots, kind="stable")
        assert np.array_equal(slots[order], whole["slots"])
        H.assert_prd_equal(np.concatenate([r["results"] for r in recs])[order], whole["results"], "interleaved parts")
        assert np.array_equal(np.concatenate([r["path"] for r in recs])[order], whole["path"])
        merged = rts.merge_groups(np.concatenate(tabs), spec["max_refl"])
        assert np.array_equal(merged["min_ray"], g_whole["min_ray"]) and np.array_equal(merged["n"], g_whole["n"])
        np.testing.assert_allclose(merged["sum_sqrt_power"], g_whole["sum_sqrt_power"], rtol=1e-12)
    # row-keyed responses vs the oracle: the representative is the same ray, named by its buffer row
    o = H.oracle_trace(oracle, spec)
    rx, rxi, oslots = oracle.filter_finalise(o["results"], o["path"], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], n)
    uniq = oracle.unique_paths(lit["pathMatch"])
    resp = rts.groups_to_responses(g_whole)
    assert np.array_equal(resp["ray"].astype(np.int64), oslots[uniq].astype(np.int64))
    np.testing.assert_allclose(resp["power"], lit["results"]["power"][uniq], rtol=1e-11)
    tr.close()


def test_dealt_tile_lists_equal_whole(rts, scenes):
    """ray sharding dealt by last-seen cost (rts_tile_records_get / _set, rts_deal_tiles, rts_set_tile_list): for ARBITRARY tile ->
    worker maps -- random ones, and the longest-first deal from the cost records of a whole pulse -- the parts' received sets,
    merged by buffer row, are the whole launch's; the records a part hands out cover exactly its tiles; a handle that never traced
    a tile takes its order from records another handle measured"""
    from rts_amd import _lib, api
    spec = scenes.config_multi(W=128)          # (a launch keeps cost records only when it is more than one sweep of the grid: > 2^18 launch indices)
    n = spec["W"] ** 3
    wl = spec["c"] / spec["carrier"]
    tr = H.gpu_tracer(rts, spec)
    H.gpu_trace(rts, spec, tr=tr); _, st_whole = H.gpu_trace(rts, spec, tr=tr)
    whole = tr.received()
    tr.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    g_whole = tr.aggregate(spec["c"], spec["carrier"], _lib.RTS_BASE_USE_ROWS)
    rec_whole = tr.tile_records_get()
    assert rec_whole.shape[0] == (n + 63) // 64 and np.count_nonzero(rec_whole) == rec_whole.shape[0]      # every wave tile of the whole pulse has a record
    rng = np.random.default_rng(7)
    other = H.gpu_tracer(rts, spec)                                  # a second handle: no history of its own
    for tile, parts, how in [(64, 3, "random"), (256, 4, "lpt"), (4096, 2, "random"), (512, 5, "lpt")]:
        n_plan = (n + tile - 1) // tile
        if how == "lpt":
            part_of, cost = api.deal_tiles(rec_whole, n, tile, parts)
            assert cost.max() - cost.min() <= np.add.reduceat((rec_whole & 0x3fffffff).astype(np.uint64), np.arange(0, rec_whole.shape[0], tile // 64)).max()
        else:
            part_of = rng.integers(0, parts, n_plan).astype(np.uint32)
        recs, tabs, seen = [], [], np.zeros(rec_whole.shape[0], np.uint32)
        for part in range(parts):
            ids = np.flatnonzero(part_of == part).astype(np.uint32)
            t = tr if part % 2 == 0 else other
            if t is other: t.tile_records_set(rec_whole)
            t.set_tile_list(tile, ids)
            expect = int(sum(min(tile, n - int(i) * tile) for i in ids))
            if expect == 0: continue
            _, st = H.gpu_trace(rts, spec, tr=t, interleave=(tile, api.INTERLEAVE_LIST, 0))
            assert st["rays"] == expect
            recs.append(t.received())
            t.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
            tabs.append(t.aggregate(spec["c"], spec["carrier"], _lib.RTS_BASE_USE_ROWS))
            r = t.tile_records_get()
            mine = np.repeat(part_of == part, tile // 64)[:r.shape[0]]
            if expect >= 400000:                                     # (launches of one sweep of the grid keep no cost records)
                assert np.all(r[~mine] == 0) and np.all(r[mine] != 0)
                assert not np.any(seen[mine]); seen[mine] = r[mine]
        slots = np.concatenate([r["slots"] for r in recs]); order = np.argsort(slots, kind="stable")
        assert np.array_equal(slots[order], whole["slots"]), (tile, parts, how)
        H.assert_prd_equal(np.concatenate([r["results"] for r in recs])[order], whole["results"], "dealt parts")
        assert np.array_equal(np.concatenate([r["path"] for r in recs])[order], whole["path"])
        merged = rts.merge_groups(np.concatenate(tabs), spec["max_refl"])
        mo, wo = np.argsort(merged["min_ray"]), np.argsort(g_whole["min_ray"])         # (the order of a group table is not part of the contract: responses are emitted by representative ray)
        assert np.array_equal(merged["min_ray"][mo], g_whole["min_ray"][wo]) and np.array_equal(merged["n"][mo], g_whole["n"][wo])
        np.testing.assert_allclose(merged["sum_sqrt_power"][mo], g_whole["sum_sqrt_power"][wo], rtol=1e-12)
    # a list inside a SUB-RANGE of the lattice (ragged: the range neither starts at 0 nor ends on a tile boundary): tile t of the list is
    # launch indices [first + t tile, first + (t + 1) tile) cut at the range's end
    first, count, tile = 3 * 4096 + 64, n // 2 + 37, 256
    n_plan = (count + tile - 1) // tile
    ids = np.sort(rng.choice(n_plan, n_plan // 3, replace=False)).astype(np.uint32)
    if ids[-1] != n_plan - 1: ids = np.append(ids, np.uint32(n_plan - 1))          # (the partial last tile of the range included)
    tr.set_tile_list(tile, ids)
    _, st = H.gpu_trace(rts, spec, tr=tr, ray_first=first, ray_count=count, interleave=(tile, api.INTERLEAVE_LIST, 0))
    assert st["rays"] == int(sum(min(tile, count - int(i) * tile) for i in ids))
    got = tr.received()
    rel = whole["slots"].astype(np.int64) - first
    keep = (rel >= 0) & (rel < count) & np.isin(np.clip(rel, 0, None) // tile, ids.astype(np.int64))
    assert np.array_equal(got["slots"], whole["slots"][keep]) and np.array_equal(got["path"], whole["path"][keep])
    H.assert_prd_equal(got["results"], whole["results"][keep], "a dealt list inside a sub-range")
    # a list is refused where it cannot be right, and forgotten on request
    with pytest.raises(Exception): tr.set_tile_list(100, np.array([0, 1], np.uint32))                # not a multiple of 64
    with pytest.raises(Exception): tr.set_tile_list(64, np.array([3, 3], np.uint32))                 # not ascending
    with pytest.raises(Exception): tr.set_tile_list(64, np.array([(n + 63) // 64], np.uint32))       # beyond W^3
    tr.set_tile_list(64, np.zeros(0, np.uint32))                                                     # an EMPTY list: a worker that was dealt nothing traces nothing
    _, st = H.gpu_trace(rts, spec, tr=tr, interleave=(64, api.INTERLEAVE_LIST, 0))
    assert st["rays"] == 0 and st["segments"] == 0 and len(tr.received()["slots"]) == 0
    tr.set_tile_list(0, np.zeros(0, np.uint32))                                                      # no list any more
    with pytest.raises(Exception): H.gpu_trace(rts, spec, tr=tr, interleave=(64, api.INTERLEAVE_LIST, 0))
    _, st = H.gpu_trace(rts, spec, tr=tr)
    assert st["rays"] == n and st["segments"] == st_whole["segments"]
    tr.close(); other.close()


def test_far_rotated_placement(rts, oracle, scenes):
    """the hierarchy is walked in target space (ray mapped through the inverse placement), the exact test runs on the
    world-space vertices: a target rotated by float-trig matrices (orthonormal only to ~1e-7) and displaced by tens of
    kilometres, seen from a transmitter that moves with it, must still give the oracle's rays bit for bit"""
    spec = scenes.config3(W=40, detail=0.1, rx_radius=300.0)
    off = np.array([31234.5678, -20987.654321, 9876.54321])
    spec["tx"] = dict(spec["tx"]); spec["tx"]["origin"] = tuple(np.asarray(spec["tx"]["origin"], np.float64) + off)
    from rts_amd import api
    spec["rx"] = [dict(r, centre=np.asarray(r["centre"], np.float64) + off) for r in spec["rx"]]     # same geometry, large coordinates
    n_recv = 0
    for ypr in [(0.0, 0.0, 0.0), (0.7, -0.3, 1.1), (3.0, 1.2, -2.5)]:
        motion = [dict(position=tuple(off), velocity=(120.0, -40.0, 8.0), rotation=api.rotation_matrix(*ypr))]
        tr = H.gpu_tracer(rts, spec); _, st = H.gpu_trace(rts, spec, tr=tr, motion=motion); g = tr.received(); tr.close()
        o = H.oracle_trace(oracle, spec, motion=motion, use_bvh=True, threads=4, debug=False)
        idx = np.nonzero(o["results"]["received"] >= 0)[0]
        assert st["segments"] == o["counters"]["segments"] and st["shaded"] == o["counters"]["shaded"]
        assert np.array_equal(g["slots"].astype(np.int64), idx)
        H.assert_prd_equal(o["results"][idx], g["results"], "far rotated placement %r" % (ypr,))
        assert np.array_equal(o["path"][idx], g["path"])
        n_recv += len(idx)
    assert n_recv > 50
    # a placement that is not rigid is refused, not mis-traced
    bad = [dict(position=tuple(off), velocity=(0.0, 0.0, 0.0), rotation=np.diag([2.0, 1.0, 1.0]))]
    tr = H.gpu_tracer(rts, spec)
    with pytest.raises(RuntimeError, match="orthonormal"):
        H.gpu_trace(rts, spec, tr=tr, motion=bad)
    tr.close()


@pytest.mark.parametrize("n_handles,linked", [(2, True), (3, False)])
def test_pipelined_pulses_identical(rts, scenes, n_handles, linked):
    """rts_trace_pulse_begin/_end over several handles (pulse k+1 enqueued before pulse k is finished; linked: serial
    trace kernels, un-linked: overlapping ones): every pulse's received set, group table and return-cube row are
    bit-identical to the strictly sequential single-handle run"""
    import torch
    from rts_amd import _lib
    spec = scenes.config3(W=48, detail=0.3)
    wl = spec["c"] / spec["carrier"]; tx = spec["tx"]; n_pulses = 5

    def motion(k):
        return scenes.config5_motion(k, speed=150.0, yaw_rate=20.0)

    def post(t, k):
        rec = t.received()
        t.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
        t.cube_accumulate(k, spec["c"], spec["carrier"])
        return rec, t.aggregate(spec["c"], spec["carrier"], _lib.RTS_BASE_USE_ROWS)

    r0 = 2.0 * abs(tx["origin"][0]); t0 = (r0 - 150.0) / spec["c"]; dt = 300.0 / spec["c"] / 256
    cubes = [torch.zeros((len(spec["rx"]), n_pulses, 256), dtype=torch.complex128, device="cuda") for _ in range(2)]
    seq = H.gpu_tracer(rts, spec)
    seq.cube_attach(len(spec["rx"]), n_pulses, 256, t0, dt, device_ptr=cubes[0].data_ptr())
    ref = []
    for k in range(n_pulses):
        seq.trace(tx["origin"], tx["span"], tx["dir"], motion(k), want_stats=False)
        ref.append(post(seq, k))
    seq.close()

    hs = [H.gpu_tracer(rts, spec) for _ in range(n_handles)]
    a, b = hs[0], hs[1]
    if linked:
        for t in hs[1:]:
            a.link(t)
    for t in hs:
        t.cube_attach(len(spec["rx"]), n_pulses, 256, t0, dt, device_ptr=cubes[1].data_ptr())
    got = [None] * n_pulses; pending = []
    for k in range(n_pulses):
        t = hs[k % n_handles]
        t.trace_begin(tx["origin"], tx["span"], tx["dir"], motion(k))
        pending.append((t, k))
        if len(pending) == n_handles:
            tt, kk = pending.pop(0); got[kk] = post(tt, kk)          # received() ends the begun pulse implicitly
    while pending:
        tt, kk = pending.pop(0); tt.trace_end(); got[kk] = post(tt, kk)
    torch.cuda.synchronize()
    assert sum(len(r[0]["slots"]) for r in ref) > 100
    for k in range(n_pulses):
        assert np.array_equal(ref[k][0]["slots"], got[k][0]["slots"])
        H.assert_prd_equal(ref[k][0]["results"], got[k][0]["results"], "pipelined pulse %d" % k)
        assert np.array_equal(ref[k][0]["path"], got[k][0]["path"])
        assert ref[k][1].tobytes() == got[k][1].tobytes()
    assert torch.allclose(torch.view_as_real(cubes[0]), torch.view_as_real(cubes[1]), rtol=1e-12, atol=1e-30)   # (f64 atomics: bin sums may round differently)
    # protocol errors
    with pytest.raises(RuntimeError):
        a.trace_end()                                                # nothing in flight
    a.trace_begin(tx["origin"], tx["span"], tx["dir"], motion(0))
    with pytest.raises(RuntimeError):
        a.trace_begin(tx["origin"], tx["span"], tx["dir"], motion(1))
    a.trace_end()
    with pytest.raises(RuntimeError):
        a.link(a)
    for t in hs:
        t.close()


def test_tile_schedule_does_not_change_results(rts, scenes, monkeypatch):
    """the trace kernel's waves draw 64-index tiles from a queue ordered by what each tile cost in the handle's earlier
    launches: launch 1 runs in index order, launches 2.. longest-first, other launch shapes reuse the per-global-tile
    history -- the received set must be the same bits every time, and the same as with the ordering switched off"""
    spec = scenes.config3(W=72, detail=0.3, rx_radius=300.0)             # 373 k launch indices: several tiles per wave
    n = spec["W"] ** 3

    def snapshot(tr, **kw):
        _, st = H.gpu_trace(rts, spec, tr=tr, **kw)
        r = tr.received()
        return st, r

    tr = H.gpu_tracer(rts, spec)
    st0, a = snapshot(tr)
    assert st0["received"] > 50
    for _ in range(3):
        st, b = snapshot(tr)
        assert st["segments"] == st0["segments"] and st["shaded"] == st0["shaded"]
        assert np.array_equal(a["slots"], b["slots"]) and np.array_equal(a["path"], b["path"])
        H.assert_prd_equal(a["results"], b["results"], "re-ordered tiles")
        np.testing.assert_array_equal(a["rcs_angle"], b["rcs_angle"])
    # another launch shape on the same handle (interleaved part), then the whole pulse again
    _, part = snapshot(tr, interleave=(4096, 2, 1))
    assert set(part["slots"].tolist()) <= set(a["slots"].tolist())
    _, c = snapshot(tr)
    H.assert_prd_equal(a["results"], c["results"], "after a different launch shape")
    tr.close()
    monkeypatch.setenv("RTS_TILE_LPT", "0")
    # (read per handle at creation: this handle runs without the cost ordering)
    tr2 = H.gpu_tracer(rts, spec)
    _, d = snapshot(tr2)
    H.assert_prd_equal(a["results"], d["results"], "fresh handle")
    assert np.array_equal(a["slots"], d["slots"])
    tr2.close()


def test_c4_full_size_sampled_parity(rts, oracle, scenes):
    """BASELINE configs[3] at full size (4 airframes, 999 824 triangles, W = 465: 100 544 625 launch indices, 8 bounces,
    8 receivers), BOTH transmitters (the reference's transmitter loop, ray_tracer.cpp:806-832): segment accounting + every
    1999th launch index through the oracle (BVH mode); the received subset must match record for record -- for the handle's FIRST
    launch (ordinary kernel only) and for its THIRD (tile-cost history: the tiles at the head of the order traced by the
    cooperative kernel, RtsStats.coop_tiles > 0), whose whole received set must also equal the first's; two interleaved parts
    must add up to the whole"""
    spec = scenes.config4()
    spec["rx"] = [dict(r, radius=max(r["radius"], 400.0)) for r in spec["rx"]]        # wide capture spheres: enough received rays to compare
    n = spec["W"] ** 3
    stride = 1999; m = n // stride
    samp = 11 + stride * np.arange(m, dtype=np.int64)
    tr = H.gpu_tracer(rts, spec)
    for ti, tx in enumerate(spec["tx_list"]):
        spec["tx"] = tx
        o = H.oracle_trace(oracle, spec, ray_first=11, ray_stride=stride, n_rays=m, use_bvh=True, threads=8, debug=False)
        o_idx = np.nonzero(o["results"]["received"] >= 0)[0]
        assert len(o_idx) > 0

        def against_oracle(rec, what):
            slots = rec["slots"].astype(np.int64); R = len(slots)
            pos_c = np.minimum(np.searchsorted(slots, samp), R - 1)
            is_recv = slots[pos_c] == samp
            assert np.array_equal(np.nonzero(is_recv)[0], o_idx), what
            H.assert_prd_equal(o["results"][o_idx], rec["results"][pos_c[is_recv]], "C4 Tx %d sampled received records, %s" % (ti, what))
            assert np.array_equal(o["path"][o_idx], rec["path"][pos_c[is_recv]]), what
        _, st = H.gpu_trace(rts, spec, tr=tr)
        whole = tr.received()
        R = st["received"]
        assert st["rays"] == n and st["segments"] == n + st["shaded"] and R == len(whole["slots"]) and R > 200
        assert (np.diff(whole["slots"].astype(np.int64)) > 0).all()
        against_oracle(whole, "first launch of this transmitter")
        if ti == 0:
            _, st2 = H.gpu_trace(rts, spec, tr=tr)
            _, st3 = H.gpu_trace(rts, spec, tr=tr)
            third = tr.received()
            assert st3["coop_tiles"] > 0 and st3["segments"] == st["segments"] and st3["received"] == R, (st3["coop_tiles"], st3["segments"], st["segments"])
            against_oracle(third, "third launch: cooperative kernel engaged on %d tiles" % st3["coop_tiles"])
            assert np.array_equal(third["slots"], whole["slots"]) and np.array_equal(third["path"], whole["path"])
            H.assert_prd_equal(third["results"], whole["results"], "C4 third launch (cooperative head) against the first")
            parts = []
            for part in range(2):
                H.gpu_trace(rts, spec, tr=tr, interleave=(4096, 2, part)); parts.append(tr.received())
            ps = np.concatenate([p["slots"] for p in parts]); order = np.argsort(ps, kind="stable")
            assert np.array_equal(ps[order], whole["slots"])
            H.assert_prd_equal(np.concatenate([p["results"] for p in parts])[order], whole["results"], "C4 interleaved halves")
    tr.close()


def test_c5_full_size_moving_target_sampled_parity(rts, oracle, scenes):
    """BASELINE configs[4] at full size (the 100 000-triangle airframe translating 200 m/s and yawing 1 rad/s, W = 216):
    three pulses spread over the interval on ONE handle (static hierarchy, re-placed per pulse; tile order learnt from
    the previous pulse); every 499th launch index of each pulse through the oracle with that pulse's world-space mesh"""
    spec = scenes.config5(rx_radius=300.0)
    n = spec["W"] ** 3
    tr = H.gpu_tracer(rts, spec)
    total_recv = 0
    for k in (0, 37, 90):
        motion = spec["motion_fn"](k)
        _, st = H.gpu_trace(rts, spec, tr=tr, motion=motion)
        g = tr.received(); R = st["received"]
        assert st["rays"] == n and st["segments"] == n + st["shaded"] and st["bvh_rebuilt"] == 1
        stride = 499; m = n // stride
        o = H.oracle_trace(oracle, spec, motion=motion, ray_first=3, ray_stride=stride, n_rays=m, use_bvh=True, threads=8, debug=False)
        samp = 3 + stride * np.arange(m, dtype=np.int64)
        o_idx = np.nonzero(o["results"]["received"] >= 0)[0]
        slots = g["slots"].astype(np.int64)
        if R == 0:
            assert len(o_idx) == 0
            continue
        pos_c = np.minimum(np.searchsorted(slots, samp), R - 1)
        is_recv = slots[pos_c] == samp
        assert np.array_equal(np.nonzero(is_recv)[0], o_idx)
        H.assert_prd_equal(o["results"][o_idx], g["results"][pos_c[is_recv]], "C5 pulse %d sampled received records" % k)
        assert np.array_equal(o["path"][o_idx], g["path"][pos_c[is_recv]])
        total_recv += len(o_idx)
    assert total_recv > 20
    tr.close()


@pytest.mark.parametrize("variant", ["icosphere", "file10k"])
def test_c2_full_size_sampled_parity(rts, oracle, scenes, variant):
    """BASELINE configs[1] at full size -- W = 100 (1 000 000 launch indices), 4 bounces, on the n = 5 icosphere (20 480
    triangles) and on the 10 000-triangle sphere loaded through rts_file_mesh (unshared vertices, the "10k tris" of the config's
    wording): every launch index's segment accounting plus every 97th launch index through the oracle (BVH mode)"""
    spec = scenes.config2(rx_radius=200.0) if variant == "icosphere" else scenes.config2_file(rx_radius=200.0)
    n = spec["W"] ** 3
    assert n == 1000000 and spec["meshes"][0]["tris"].shape[0] == (20480 if variant == "icosphere" else 10000)
    tr = H.gpu_tracer(rts, spec)
    _, st = H.gpu_trace(rts, spec, tr=tr)
    g = tr.received(); R = st["received"]
    assert st["rays"] == n and st["segments"] == n + st["shaded"] and R > 1000
    stride = 97; m = n // stride
    o = H.oracle_trace(oracle, spec, ray_first=5, ray_stride=stride, n_rays=m, use_bvh=True, threads=8, debug=False)
    samp = 5 + stride * np.arange(m, dtype=np.int64)
    o_idx = np.nonzero(o["results"]["received"] >= 0)[0]
    slots = g["slots"].astype(np.int64)
    pos_c = np.minimum(np.searchsorted(slots, samp), R - 1)
    is_recv = slots[pos_c] == samp
    assert np.array_equal(np.nonzero(is_recv)[0], o_idx) and len(o_idx) > 20
    H.assert_prd_equal(o["results"][o_idx], g["results"][pos_c[is_recv]], "C2 %s sampled received records" % variant)
    assert np.array_equal(o["path"][o_idx], g["path"][pos_c[is_recv]])
    tr.close()


@pytest.mark.parametrize("seed,refr,far", [(1, 0, False), (2, 0, False), (3, 1, False), (4, 0, True), (5, 1, True)])
def test_random_triangle_soups_brute_force(rts, oracle, scenes, seed, refr, far):
    """fuzz of the hierarchy builder (SAH, split references) and of the conservative walk: random triangle soups with
    mixed scales, long diagonal slivers, fans around a shared vertex, coincident and degenerate triangles, two targets
    (one rotated and displaced) -- every output row of every launch index against the oracle's BRUTE-FORCE closest hit"""
    rng = np.random.default_rng(seed)

    def soup(n_reg, n_sliver, n_fan, centre):
        tris = []
        for _ in range(n_reg):                                         # ordinary triangles of mixed size
            c = rng.normal(0, 6.0, 3); s = 10 ** rng.uniform(-1.5, 0.7)
            tris.append(c + rng.normal(0, s, (3, 3)))
        for _ in range(n_sliver):                                      # long thin diagonal slivers
            a = rng.normal(0, 6.0, 3); d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
            L = rng.uniform(5, 25); w = rng.normal(0, 1, 3) * 10 ** rng.uniform(-4, -1.5)
            tris.append(np.stack([a, a + L * d, a + 0.5 * L * d + w]))
        hub = rng.normal(0, 3.0, 3)                                    # a fan of thin wedges around one vertex
        ang = np.sort(rng.uniform(0, 2 * np.pi, n_fan + 1)); rad = rng.uniform(2, 8)
        e1 = np.array([0.0, 1.0, 0.2]); e2 = np.array([0.1, -0.2, 1.0])
        for i in range(n_fan):
            tris.append(np.stack([hub, hub + rad * (np.cos(ang[i]) * e1 + np.sin(ang[i]) * e2), hub + rad * (np.cos(ang[i + 1]) * e1 + np.sin(ang[i + 1]) * e2)]))
        tris.append(tris[0].copy())                                    # an exact duplicate (tie on t: lowest primitive id wins)
        p = rng.normal(0, 5.0, 3); tris.append(np.stack([p, p, p + 1.0]))   # degenerate (zero area)
        v = np.concatenate(tris).astype(np.float64) + np.asarray(centre, np.float64)
        t = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
        nrm = np.repeat(np.cross(v[1::3] - v[0::3], v[2::3] - v[0::3]) + 1e-30, 3, axis=0)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        return dict(tris=t, verts=v, normals=nrm, refl_coeff=0.8, refr_index=1.0)

    from rts_amd import api as A
    off = np.array([41234.5, -30987.25, 20011.125]) if far else np.zeros(3)      # the whole scene far from the origin
    spec = scenes.config1()
    m0, m1 = soup(120, 40, 60, (0, 0, 0)), soup(60, 20, 30, (4.0, -3.0, 2.0))
    if refr:
        m0["refr_index"] = 1.4; m1["refr_index"] = 1.2; spec["max_refr"] = 1
    spec.update(W=14 if not refr else 10, max_refl=3, smooth=bool(seed % 2), meshes=[m0, m1],
                motion=[dict(position=tuple(off), velocity=(5.0, 0.0, 0.0)),
                        dict(position=tuple(off + (12.0, 7.0, -4.0)), velocity=(0.0, -3.0, 1.0), rotation=A.rotation_matrix(0.4 * seed, -0.3, 0.9))])
    spec["tx"] = dict(origin=tuple(off + (-300.0, 2.0, 1.0)), span=(0.16, 0.14, 0.08), dir=(0.0, 0.0))
    spec["rx"] = [A.rx_sphere(tuple(off + (-300.0, 2.0, 1.0)), 0.0, 0.0, 120.0, 2.6, 2.6), A.rx_sphere(tuple(off + (-100.0, 200.0, 30.0)), -1.1, -0.1, 150.0, 2.6, 2.6)]
    tr, st, o, g = full_parity(rts, oracle, spec)
    assert st["shaded"] > 200
    nodes, leaf_prim, roots = tr.bvh()
    if tr.scene_info()["builder"] == 0:
        assert np.bincount(leaf_prim).max() > 1                        # host SAH builder: the slivers and wedges were split
    tr.close()


def _all_equal(a, b, what):
    H.assert_prd_equal(a["results"], b["results"], what)
    for k in ("path", "rcs_angle", "hit_prim"):
        np.testing.assert_array_equal(a[k], b[k], err_msg="%s: %s" % (what, k))
    assert np.array_equal(a["hit_t"].view(np.uint32), b["hit_t"].view(np.uint32)), what


def test_primary_prefilter_is_invisible(rts, scenes):
    """the f32 pre-filter of primary rays (direction mask over the placed triangles + widened receiver spheres,
    rts_internal.h) may only ever say "this ray certainly meets nothing": every output buffer -- all launch indices, the
    per-segment hit trace, the received set and its order -- must be the same bits with the filter on and off
    (RTS_FLAG_NO_PREFILTER), wherever the scene sits, whatever the beam looks like; and it must really be switched on where
    it can be (fewer node visits), and off where its frame does not exist"""
    import math
    ecef = scenes.ecef_offset(lat=math.pi / 2)
    c3 = scenes.config3(W=56, detail=0.3, rx_radius=300.0)
    behind = scenes.config_multi(W=16)                                     # a target partly BEHIND the transmitter: mask void
    behind["tx"] = dict(behind["tx"], origin=(2.0, 0.5, 0.3), span=(2.4, 2.4, 0.1))
    inside = scenes.config_multi(W=16)                                     # transmitter INSIDE a capture sphere
    inside["rx"] = list(inside["rx"]) + [scenes.rx_window(tuple(np.add(inside["tx"]["origin"], (1.0, 2.0, -1.0))), 25.0, (-3.2, 3.2), (-1.6, 1.6))]
    narrow = dict(c3, tx=dict(c3["tx"], span=(2.0e-4, 2.0e-4, 0.1)))      # beam narrower than 32 minimum cells: no mask
    wide = dict(scenes.config_multi(W=16)); wide["tx"] = dict(wide["tx"], span=(2.9, 2.9, 0.1))   # > 120 degrees: no mask
    c5 = scenes.config5(W=40, detail=0.3)
    cases = [("c3", c3, c3["motion"], True), ("c3 ecef", scenes.translate(c3, ecef), None, True), ("miss branches", scenes.config_miss_branches(), None, None),
             ("pole", scenes.config_pole(True), None, None), ("target behind tx", behind, None, None), ("tx inside rx", inside, None, None),
             ("narrow beam", narrow, None, False), ("wide beam", wide, None, False), ("refraction", scenes.config_multi(W=12, max_refl=2), None, None),
             ("c5 pulse 7", c5, c5["motion_fn"](7), True)]
    for name, spec, motion, engaged in cases:
        if name == "refraction":
            spec = dict(spec, max_refr=1)
            spec["meshes"] = [dict(m, refl_coeff=0.6, refr_index=1.5) for m in spec["meshes"]]
            spec["rx"] = spec["rx"] + [scenes._rx_at((200.0, 0.0, 0.0), (0, 0, 0), 90.0, 2.6)]
        motion = motion if motion is not None else spec["motion"]
        n = spec["W"] ** 3
        out = {}
        for on in (True, False):
            tr = H.gpu_tracer(rts, spec, keep_all=True, count_traversal=True, pre_filter=on)
            for rep in range(2):                                            # (second launch: cost-ordered tiles, adaptive switch settled)
                _, st = H.gpu_trace(rts, spec, tr=tr, motion=motion)
            out[on] = (tr.all_rays(n), tr.received(), st)
            tr.close()
        (a, ra, sa), (b, rb, sb) = out[True], out[False]
        _all_equal(a, b, name)
        assert np.array_equal(ra["slots"], rb["slots"]) and np.array_equal(ra["path"], rb["path"]), name
        H.assert_prd_equal(ra["results"], rb["results"], name + " (received)")
        assert (sa["segments"], sa["shaded"], sa["received"]) == (sb["segments"], sb["shaded"], sb["received"]) and sa["tri_tests"] <= sb["tri_tests"], name
        if engaged is True:
            assert sa["node_visits"] < sb["node_visits"], (name, sa["node_visits"], sb["node_visits"])
        elif engaged is False:
            assert sa["node_visits"] == sb["node_visits"], name


def test_cooperative_units_are_invisible(rts, oracle, scenes, monkeypatch):
    """the tiles at the head of a handle's cost order are traced by the COOPERATIVE kernel -- one launch index per wave, its
    64 lanes sharing out the walk (rts_walk_coop) -- beside the ordinary kernel; which tiles those are depends on timings of
    the previous launch, so every output buffer must be the same bits whatever the split: no cooperative units
    (RTS_COOP_FRAC=0) against EVERY tile that cost anything (threshold 0), in KEEP_ALL + counting builds and in the product
    build, with refraction, with the LDS stack cut to 3 entries (walks spill to the global slab while lanes hand subtrees
    over), at Earth-centred coordinates"""
    import math
    monkeypatch.setenv("RTS_GRID_MULT", "1")                               # 256 blocks: a cost order exists from ~65 k launch indices on
    c3 = scenes.config3(W=56, detail=0.3, rx_radius=300.0)
    multi = scenes.config_multi(W=44)
    refr = dict(scenes.config_multi(W=42, max_refl=2), max_refr=1)
    refr["meshes"] = [dict(m, refl_coeff=0.6, refr_index=1.5) for m in refr["meshes"]]
    refr["rx"] = refr["rx"] + [scenes._rx_at((200.0, 0.0, 0.0), (0, 0, 0), 90.0, 2.6)]
    cases = [("c3", c3, {}), ("c3 ecef", scenes.translate(c3, scenes.ecef_offset(lat=math.pi / 2)), {}), ("multi", multi, {}), ("refraction", refr, {}),
             ("c3 short stack", c3, {"RTS_STACK_LDS_DEBUG": "3"}), ("miss branches", scenes.config_miss_branches(W=44), {}),
             ("c3, plain records in the cooperative walk", c3, {"RTS_COOP_VERSIONS": "0"}),      # (default since round 5: the cooperative kernel of the product / counting builds walks the octant versions)
             ("multi, plain records, short stack", multi, {"RTS_COOP_VERSIONS": "0", "RTS_STACK_LDS_DEBUG": "3"})]
    for name, spec, env in cases:
        n = spec["W"] ** 3
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = {}
        for mode in ("off", "all"):
            monkeypatch.setenv("RTS_COOP_FRAC", "0" if mode == "off" else "1e-12"); monkeypatch.setenv("RTS_COOP_FLOOR", "0"); monkeypatch.setenv("RTS_COOP_SEG", "0")
            tr = H.gpu_tracer(rts, spec, keep_all=True, count_traversal=True)
            tp = H.gpu_tracer(rts, spec)                                        # the product build
            for rep in range(3):                                                # (launches 2 and 3 have a cost history)
                _, st = H.gpu_trace(rts, spec, tr=tr); _, sp = H.gpu_trace(rts, spec, tr=tp)
            out[mode] = (tr.all_rays(n), tr.received(), st, tp.received(), sp)
            tr.close(); tp.close()
        for k in env:
            monkeypatch.delenv(k)
        (a, ra, sa, pa, spa), (b, rb, sb, pb, spb) = out["off"], out["all"]
        _all_equal(a, b, name)
        for x, y, what in ((ra, rb, "counting build"), (pa, pb, "product build"), (ra, pb, "product vs counting")):
            assert np.array_equal(x["slots"], y["slots"]) and np.array_equal(x["path"], y["path"]), (name, what)
            H.assert_prd_equal(x["results"], y["results"], "%s (received, %s)" % (name, what))
            np.testing.assert_allclose(x["rcs_angle"], y["rcs_angle"], rtol=0, atol=1e-12)
        assert (sa["segments"], sa["shaded"], sa["received"]) == (sb["segments"], sb["shaded"], sb["received"]) == (spb["segments"], spb["shaded"], spb["received"]), name
        assert sa["received"] > 0 and sb["tri_tests"] >= sb["shaded"] > 0, name
        # ... and the launch compared really was traced by the cooperative kernel (RtsStats.coop_tiles of the handle's THIRD launch:
        # every tile with a cost record at the head of the order), whose every ray is then held against the ORACLE directly -- brute
        # force over all primitives, hit by hit: not only against the ordinary kernel's launch
        assert sa["coop_tiles"] == 0 and sb["coop_tiles"] > 0 and spb["coop_tiles"] > 0, (name, sa["coop_tiles"], sb["coop_tiles"], spb["coop_tiles"])
        if name in ("multi", "refraction", "miss branches"):
            rows = spec["max_refl"] + 3 if spec.get("max_refr", 0) else 1
            o = H.oracle_trace(oracle, spec, use_bvh=False, threads=8)
            H.compare_full(o, b, n * rows)
            assert sb["segments"] == o["counters"]["segments"] and sb["shaded"] == o["counters"]["shaded"], name
        # the cooperative walk shares the prune bound late and opens subtrees in another order: it may test more, never fewer
        # triangles than ... nothing is guaranteed either way; what IS: both found the same closest hits (above)


def test_octant_versions_are_invisible(rts, oracle, scenes, monkeypatch):
    """round 5: the ordinary trace kernel walks the OCTANT VERSIONS of the node records (k_node_versions, rts_api.hip: per node eight
    records whose planes are entry / exit planes for a ray of that octant and whose children sit in front-to-back order along the
    octant's diagonal -- no sorting network, no per-axis fetch addresses; k_trace<.., VERS>).  The visiting ORDER only decides what
    is pruned: every output buffer must be the same bits with the versions walked (the default) and with the role fetch + sorted
    children (RTS_WALK_VERSIONS=0) -- KEEP_ALL + counting builds and the product build, refraction, the LDS stack cut to 3 entries
    (the versions' deep-stack branch), Earth-centred coordinates, both sort keys of the versions (centre, the default / entry corner), a rotated
    target (the octant is taken in the TARGET's frame); and the versions' launch against the oracle's brute force"""
    import math
    c3 = scenes.config3(W=56, detail=0.3, rx_radius=300.0)
    multi = scenes.config_multi(W=44)
    refr = dict(scenes.config_multi(W=42, max_refl=2), max_refr=1)
    refr["meshes"] = [dict(m, refl_coeff=0.6, refr_index=1.5) for m in refr["meshes"]]
    refr["rx"] = refr["rx"] + [scenes._rx_at((200.0, 0.0, 0.0), (0, 0, 0), 90.0, 2.6)]
    yaw = 0.9; cy, sy = math.cos(yaw), math.sin(yaw)
    turned = dict(c3, motion=[dict(m, rotation=(cy, -sy, 0.0, sy, cy, 0.0, 0.0, 0.0, 1.0)) for m in c3["motion"]])
    cases = [("c3", c3, {}), ("c3 ecef", scenes.translate(c3, scenes.ecef_offset(lat=math.pi / 2)), {}), ("multi", multi, {}), ("refraction", refr, {}),
             ("c3 short stack", c3, {"RTS_STACK_LDS_DEBUG": "3"}), ("miss branches", scenes.config_miss_branches(W=44), {}),
             ("c3 corner key", c3, {"RTS_VERSION_KEY": "corner"}), ("c3 turned", turned, {})]
    for name, spec, env in cases:
        n = spec["W"] ** 3
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("RTS_WALK_VERSIONS", mode)
            tr = H.gpu_tracer(rts, spec, keep_all=True, count_traversal=True)
            tp = H.gpu_tracer(rts, spec)                                        # the product build
            assert tr.scene_info()["version_bytes"] > 0, name                   # (the scene HAS versions either way: the handle decides whether to walk them)
            for rep in range(2):                                                # (the second launch has a cost order)
                _, st = H.gpu_trace(rts, spec, tr=tr); _, sp = H.gpu_trace(rts, spec, tr=tp)
            out[mode] = (tr.all_rays(n * (spec["max_refl"] + 3 if spec.get("max_refr", 0) else 1)), tr.received(), st, tp.received(), sp)
            tr.close(); tp.close()
        for k in env:
            monkeypatch.delenv(k)
        (a, ra, sa, pa, spa), (b, rb, sb, pb, spb) = out["0"], out["1"]
        if name == "c3":
            c3_received = pb
        _all_equal(a, b, name)
        for x, y, what in ((ra, rb, "counting build"), (pa, pb, "product build"), (ra, pb, "product vs counting")):
            assert np.array_equal(x["slots"], y["slots"]) and np.array_equal(x["path"], y["path"]), (name, what)
            H.assert_prd_equal(x["results"], y["results"], "%s (received, %s)" % (name, what))
            assert x["rcs_angle"].tobytes() == y["rcs_angle"].tobytes(), (name, what)
        assert (sa["segments"], sa["shaded"], sa["received"], sa["walked_segments"]) == (sb["segments"], sb["shaded"], sb["received"], sb["walked_segments"]), name
        assert (spa["segments"], spa["shaded"], spa["received"]) == (spb["segments"], spb["shaded"], spb["received"]) == (sb["segments"], sb["shaded"], sb["received"]), name
        assert sa["received"] > 0 and sb["tri_tests"] >= sb["shaded"] > 0, name
        # the fixed order prunes a little less well than the sorted one: a few per cent more node visits are the price of the step
        assert sb["node_visits"] <= 1.25 * sa["node_visits"] and sb["tri_tests"] <= 1.25 * sa["tri_tests"], (name, sa["node_visits"], sb["node_visits"], sa["tri_tests"], sb["tri_tests"])
        if name in ("multi", "refraction", "c3 turned"):
            rows = spec["max_refl"] + 3 if spec.get("max_refr", 0) else 1
            o = H.oracle_trace(oracle, spec, use_bvh=False, threads=8)
            H.compare_full(o, b, n * rows)
            assert sb["segments"] == o["counters"]["segments"] and sb["shaded"] == o["counters"]["shaded"], name
    monkeypatch.setenv("RTS_NODE_VERSIONS", "0")                           # a scene WITHOUT versions: the handle walks the plain records
    tr = H.gpu_tracer(rts, c3)
    assert tr.scene_info()["version_bytes"] == 0
    _, st = H.gpu_trace(rts, c3, tr=tr)
    H.assert_prd_equal(tr.received()["results"], c3_received["results"], "a scene built without versions")
    tr.close()


def test_dead_tile_batches_are_invisible(rts, oracle, scenes, monkeypatch):
    """round 5: the part of the cost order that holds the DEAD wave tiles (all 64 launch indices cleared by the pre-filter) is drawn 64
    positions at a time and one LANE screens one whole tile (rts_tile_maybe: the tile's rays are a row segment of the lattice, its end
    rays bound what any of them can reach) -- conservative, so nothing observable may change: RTS_DEAD_BATCH=0 (every tile by the wave),
    1 (the default: the order's dead part) and `all` (EVERY position screened tile-wise first, from the handle's first launch on)
    must give the same received set, segment and hit counts and cost-record table, pulse after pulse on one handle while the target
    moves through the beam (tiles die and come to life), for whole pulses, an interleaved part and a dealt tile list, with receivers
    in the beam (direct rays), at Earth-centred coordinates, W = 70 / 81 (rows that straddle wave tiles); product and counting builds;
    and the `all` launch against the oracle's brute force.  (rx_radius 120: at 300 the transmitter sits INSIDE two capture spheres, every primary ray may be
    captured and no tile is dead)"""
    import math
    monkeypatch.setenv("RTS_GRID_MULT", "1")                               # 256 blocks: a cost order exists from ~65 k launch indices on
    monkeypatch.setenv("RTS_COOP_FRAC", "0")                               # (no cooperative kernel: which tiles it takes follows measured TIMES, and a tile it traces as 64 units leaves the sum of their
                                                                           # durations as its record even when every one of its rays is dead -- the record tables below are compared entry by entry)
    c3 = scenes.config3(W=81, detail=0.3, rx_radius=120.0); c3["tx"] = dict(c3["tx"], span=(0.08, 0.07, c3["tx"]["span"][2]))      # (a beam wider than the airframe: dead tiles around it)
    direct = dict(c3, rx=c3["rx"] + [scenes._rx_at((1500.0, 20.0, 5.0), (-1000.0, 0, 0), 60.0, 2.6), scenes._rx_at((-100.0, -10.0, 0.0), (-1000.0, 0, 0), 5.0, 2.6)])      # in the beam behind / in front of the target: direct rays
    wide = dict(scenes.config3(W=70, detail=0.3, rx_radius=120.0)); wide["tx"] = dict(wide["tx"], span=tuple(3.0 * x for x in wide["tx"]["span"][:2]) + (wide["tx"]["span"][2],))
    cases = [("c3", c3), ("c3 ecef", scenes.translate(c3, scenes.ecef_offset(lat=math.pi / 2))), ("direct rays", direct), ("wide beam", wide)]
    for name, spec in cases:
        tx = spec["tx"]; n_all = spec["W"] ** 3
        out = {}
        for mode in ("0", "1", "all"):
            monkeypatch.setenv("RTS_DEAD_BATCH", mode)
            for count in (False, True):
                tr = rts.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], count_traversal=count)
                tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])
                res = []
                for k in range(6):
                    mo = [dict(m, position=tuple(np.add(m["position"], (0.5 * k, 4.0 * k, -2.5 * k)))) for m in spec["motion"]]      # across the beam: tiles die and come to life
                    il = (4096, 2, 1) if k == 3 else None
                    if k == 4:
                        tr.set_tile_list(4096, np.arange(0, (n_all + 4095) // 4096, 3, dtype=np.uint32)); il = (4096, rts.INTERLEAVE_LIST, 0)
                    st = tr.trace(tx["origin"], tx["span"], tx["dir"], mo, ray_first=0, ray_count=n_all, interleave=il)
                    res.append((st, tr.received(), tr.tile_records_get() & 0x3fffffff))
                out[(mode, count)] = res
                tr.close()
        for count in (False, True):
            for mode in ("1", "all"):
                for k, ((sa, ra, ta), (sb, rb, tb)) in enumerate(zip(out[("0", count)], out[(mode, count)])):
                    for f in ("rays", "segments", "shaded", "received") + (("node_visits", "tri_tests", "walked_segments") if count else ()):
                        assert sa[f] == sb[f], (name, mode, count, k, f, sa[f], sb[f])
                    H.assert_prd_equal(ra["results"], rb["results"], "%s: batches %s, pulse %d" % (name, mode, k))
                    assert np.array_equal(ra["slots"], rb["slots"]) and np.array_equal(ra["path"], rb["path"]) and ra["rcs_angle"].tobytes() == rb["rcs_angle"].tobytes()
                    assert np.array_equal(ta == 1, tb == 1) and np.array_equal(ta == 0, tb == 0), (name, mode, count, k)      # the same tiles are dead, the same have no record
        st, rec, _ = out[("all", False)][5]
        assert st["rays"] == n_all and st["received"] > 20, (name, st["received"])
        assert (out[("0", False)][2][2] == 1).sum() > 0.2 * len(out[("0", False)][2][2]), name      # (there ARE dead tiles to batch; the transmitter is outside every capture sphere)
        if name in ("c3", "direct rays"):
            mo = [dict(m, position=tuple(np.add(m["position"], (0.5 * 5, 4.0 * 5, -2.5 * 5)))) for m in spec["motion"]]
            idx = rec["slots"][:: max(len(rec["slots"]) // 300, 1)].astype(np.int64)
            sc = H.oracle_scene(oracle, spec, mo)
            for i in idx[:300]:
                o = sc.trace(tx["origin"], tx["span"], tx["dir"], spec["W"], spec["max_refl"], 0, spec["smooth"], ray_first=int(i), ray_stride=1, n_rays=1, use_bvh=False)
                j = int(np.searchsorted(rec["slots"], i))
                H.assert_prd_equal(o["results"][:1], rec["results"][j:j + 1], "launch index %d against the brute-force oracle" % i)


def test_receiver_window_screen_is_invisible(rts, oracle, scenes, monkeypatch):
    """round 5: the pre-filter also asks whether a primary ray that CROSSES a capture sphere can have a crossing point inside the receiver's
    angular window (rts_rx_maybe: f32, no arctangent, an explicit error budget) -- a monostatic radar's sphere touches the transmitter
    and every ray of the beam crosses it.  Conservative or wrong: with the screen (default) and without (RTS_RX_WINDOW_SCREEN=0) every
    output buffer must hold the same bits, and the screened launch is held against the oracle's brute force ray by ray.  Scenes: the
    monostatic sphere (transmitter ON it, a hair inside, a hair outside), the transmitter well inside, spheres ahead in the beam at
    several impact parameters; for each a sweep of windows whose azimuth / elevation EDGE passes through the boresight ray's crossing
    points (from well clear on one side to well clear on the other, in steps down to 1e-4 rad), narrow and wide windows; origin-centred
    and Earth-centred (where the reference's own quadratic is off by centimetres); also with dead-tile batches screening whole tiles"""
    import math
    base = scenes.config_multi(W=26, max_refl=2)
    tx = dict(origin=(-200.0, 3.0, -2.0), span=(0.10, 0.08, 0.05), dir=(0.02, -0.015))
    o = np.asarray(tx["origin"]); az, el = tx["dir"]
    bore = np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])
    side = np.cross(bore, (0.0, 0.0, 1.0)); side /= np.linalg.norm(side)
    spheres = [("monostatic", o + 12.0 * bore, 12.0), ("a hair inside", o + 12.0 * bore * (1 - 2e-4), 12.0), ("a hair outside", o + 12.0 * bore * (1 + 2e-4), 12.0),
               ("inside", o + 3.0 * bore + 2.0 * side, 12.0), ("ahead, central", o + 60.0 * bore, 9.0), ("ahead, off axis", o + 80.0 * bore + 2.5 * side, 4.0),
               ("ahead, grazing", o + 50.0 * bore + 5.6 * side, 4.0)]
    rng = np.random.default_rng(5)
    n_checked = 0; n_received_direct = 0
    for ecef in (False, True):
        for name, c, r in spheres:
            m = o - c; b = float(-m @ bore); disc = b * b - float(m @ m - r * r)
            pts = [m + (b + sg * math.sqrt(max(disc, 0.0))) * bore for sg in (-1.0, 1.0)] if disc > 0 else [m / max(np.linalg.norm(m), 1e-9) * r]
            rxs = []
            for pnt in pts:
                th = math.atan2(pnt[1], pnt[0]); ph = math.asin(max(-1.0, min(1.0, pnt[2] / np.linalg.norm(pnt))))
                for hw in (0.02, 0.4, 1.2):
                    for edge in (-3e-2, -1e-3, -1e-4, 0.0, 1e-4, 1e-3, 3e-2):
                        rxs.append(scenes.rx_window(tuple(c), r, (th + edge, th + edge + 2 * hw), (max(ph - hw, -1.57), min(ph + hw, 1.57))))     # azimuth edge at the point
                        rxs.append(scenes.rx_window(tuple(c), r, (th - hw, th + hw), (max(min(ph + edge, 1.5), -1.57), min(ph + edge + 2 * hw, 1.57))))   # elevation edge at the point
            order = rng.permutation(len(rxs))
            for g in range(0, min(len(rxs), 42), 14):                 # 14 receivers per launch (<= 16: the pre-filter is on), three launches of a random half of the sweep
                spec = dict(base, name="window-%s" % name, tx=tx, rx=[rxs[i] for i in order[g:g + 14]])
                if ecef:
                    spec = scenes.translate(spec, scenes.ecef_offset(lat=0.7, lon=-2.0))
                n = spec["W"] ** 3
                out = {}
                for mode in ("0", "1", "batch"):
                    monkeypatch.setenv("RTS_RX_WINDOW_SCREEN", "0" if mode == "0" else "1")
                    monkeypatch.setenv("RTS_DEAD_BATCH", "all" if mode == "batch" else "0")
                    if mode == "batch":
                        tp = H.gpu_tracer(rts, spec); _, sp_ = H.gpu_trace(rts, spec, tr=tp); out[mode] = (None, tp.received(), sp_); tp.close()
                    else:
                        tr, st = H.gpu_trace(rts, spec); out[mode] = (tr.all_rays(n), tr.received(), st); tr.close()
                (a, ra, sa), (b_, rb, sb), (_, rc, sc) = out["0"], out["1"], out["batch"]
                _all_equal(a, b_, "%s%s, group %d" % (name, " ecef" if ecef else "", g))
                for x in (rb, rc):
                    assert np.array_equal(ra["slots"], x["slots"]) and np.array_equal(ra["path"], x["path"]), (name, ecef, g)
                    H.assert_prd_equal(ra["results"], x["results"], "%s (received)" % name)
                assert (sa["segments"], sa["shaded"], sa["received"]) == (sb["segments"], sb["shaded"], sb["received"]) == (sc["segments"], sc["shaded"], sc["received"]), (name, ecef, g)
                n_received_direct += int(((rb["results"]["reflDepth"] == 0) & (rb["results"]["received"] >= 0)).sum())
                if g == 0:                                             # the screened launch against the oracle, every launch index
                    oo = H.oracle_trace(oracle, spec, use_bvh=False, threads=8)
                    H.compare_full(oo, b_, n)
                    n_checked += 1
    assert n_checked == 14 and n_received_direct > 1000                # (the sweeps DO capture direct rays: the screen is not vacuously "never")
    monkeypatch.delenv("RTS_RX_WINDOW_SCREEN"); monkeypatch.delenv("RTS_DEAD_BATCH")


def test_small_received_sets_one_block_path_is_invisible(rts, scenes, monkeypatch):
    """up to 4 096 received rays (2 048 where a sort key needs 64 bits) the ordering of the received set and the aggregation run as single-block kernels
    (k_recv_order_small, k_agg_order_small, k_agg_finish_small) instead of the chain of device-wide sorts and scans
    (RTS_POST_SMALL=0).  The two share the statements that form every sum and the tile sums in between are the same kernel:
    received order, finalised rays, per-ray aggregation outputs and the group table must be the same BITS -- with and without
    refraction (64-bit row keys), for a set that fills the block exactly or nearly, and for one of a few rays"""
    c3 = scenes.config3(W=64, detail=0.3, rx_radius=300.0)
    multi = scenes.config_multi(W=40)
    refr = dict(scenes.config_multi(W=36, max_refl=2), max_refr=1)
    refr["meshes"] = [dict(m, refl_coeff=0.6, refr_index=1.5) for m in refr["meshes"]]
    refr["rx"] = refr["rx"] + [scenes._rx_at((200.0, 0.0, 0.0), (0, 0, 0), 90.0, 2.6)]
    few = scenes.config3(W=24, detail=0.3, rx_radius=120.0)
    seen = []

    def prefix_with(spec, lo_R, hi_R):
        """a prefix of the launch range that receives between lo_R and hi_R rays (bisection: the count is monotone in the prefix)"""
        n = spec["W"] ** 3
        tr = H.gpu_tracer(rts, spec)
        lo, hi = 1, n
        for _ in range(40):
            mid = (lo + hi) // 2
            _, st = H.gpu_trace(rts, spec, tr=tr, ray_first=0, ray_count=mid)
            if st["received"] > hi_R:
                hi = mid
            elif st["received"] < lo_R:
                lo = mid
            else:
                tr.close(); return mid
        tr.close()
        raise AssertionError("no prefix of %s receives %d..%d rays" % (spec["name"], lo_R, hi_R))

    for name, spec, want in (("c3", c3, (1500, 2048)), ("c3 2048", c3, (2048, 2048)), ("c3 nearly 4096", c3, (3800, 4096)), ("multi", multi, None), ("refraction", refr, (1200, 2048)), ("few", few, None)):
        count = prefix_with(spec, *want) if want else spec["W"] ** 3
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("RTS_POST_SMALL", mode)
            tr = H.gpu_tracer(rts, spec)
            _, st = H.gpu_trace(rts, spec, tr=tr, ray_first=0, ray_count=count)
            rec = tr.received()
            tr.finalise_uniform(None, 0.03, 1.0, 1.0, 1.0e10, 299792458.0)
            groups = tr.aggregate(299792458.0, 1.0e10)
            out[mode] = (rec, groups, tr.aggregated(), st)
            tr.close()
        (ra, ga, aa, sa), (rb, gb, ab, sb) = out["1"], out["0"]
        assert 0 < sa["received"] == sb["received"], name
        seen.append(sa["received"])
        assert np.array_equal(ra["slots"], rb["slots"]) and np.array_equal(ra["path"], rb["path"]) and ra["rcs_angle"].tobytes() == rb["rcs_angle"].tobytes(), name
        assert ra["results"].tobytes() == rb["results"].tobytes(), name
        assert ga.tobytes() == gb.tobytes() and len(ga) > 0, name
        for k in ("results", "delay", "phase", "pathMatch"):
            assert aa[k].tobytes() == ab[k].tobytes(), (name, k)
    assert min(seen) < 400 and 3800 <= max(seen) <= 4096 and 2048 in seen, seen     # (both ends of the one-block range were exercised; 64-bit row keys -- refraction -- up to 2 048)


def test_pulse_end_uniform_equals_the_four_calls(rts, scenes, monkeypatch):
    """rts_trace_pulse_end_uniform = rts_trace_pulse_end + rts_finalise_uniform + rts_cube_accumulate + rts_aggregate.  From a
    handle's second pulse on (the first gives it a received count to judge by) the chain is enqueued behind the trace on the
    DEVICE-side count, without the host waiting in between: received rays, finalised values, per-ray aggregation outputs, group
    table, statistics and the return cube must be the same bits as with the four calls -- for a pulse of ~1 900 rays (C3-like), of
    none, and for one whose count exceeds what the speculative chain was sized for after a small one made it speculate (the
    chain then does nothing and is run again the ordinary way when the results are asked for); RTS_SPECULATE=0 likewise"""
    c3 = scenes.config3(W=64, detail=0.3, rx_radius=300.0)
    tx = c3["tx"]; n_all = c3["W"] ** 3
    cs, fc, wl = 299792458.0, 1.0e10, 0.03

    def prefix_with(lo_R, hi_R):
        tr = H.gpu_tracer(rts, c3); lo, hi = 1, n_all
        for _ in range(40):
            mid = (lo + hi) // 2
            _, st = H.gpu_trace(rts, c3, tr=tr, ray_first=0, ray_count=mid)
            if st["received"] > hi_R: hi = mid
            elif st["received"] < lo_R: lo = mid
            else: tr.close(); return mid
        raise AssertionError("no prefix")
    small, big = prefix_with(1500, 1800), prefix_with(5000, 9000)
    plan = [small, small, 3, big, small, big, big, small]            # launch indices per pulse: speculation engages from pulse 2, meets a miss at 4 and 6
    cube_shape = (len(c3["rx"]), len(plan), 64)
    r0 = 2.0 * float(np.linalg.norm(np.asarray(tx["origin"]) - np.asarray(c3["motion"][0]["position"]))); t0 = (r0 - 150.0) / cs; dt = 300.0 / cs / 64

    def run(mode):
        monkeypatch.setenv("RTS_SPECULATE", "0" if mode == "nospec" else "1")
        tr = H.gpu_tracer(rts, c3)
        tr.cube_attach(cube_shape[0], cube_shape[1], cube_shape[2], t0, dt)
        out = []
        for k, count in enumerate(plan):
            tr.trace_begin(tx["origin"], tx["span"], tx["dir"], c3["motion"], ray_first=0, ray_count=count)
            if mode == "four":
                tr.trace_end(); tr.finalise_uniform(None, wl, 1.0, 1.0, fc, cs); tr.cube_accumulate(k, cs, fc); g = tr.aggregate(cs, fc)
            else:
                tr.trace_end_uniform(None, wl, 1.0, 1.0, fc, cs, cube_pulse=k); g = tr.groups()
            st = tr.stats(); rec = tr.received(); agg = tr.aggregated()
            out.append((g, {k2: st[k2] for k2 in ("segments", "shaded", "received")}, rec, agg))
        cube = tr.cube().copy(); tr.close()
        return out, cube
    ref, cube_ref = run("four")
    assert [o[1]["received"] for o in ref][2] < 5 and max(o[1]["received"] for o in ref) > 2048 > min(o[1]["received"] for o in ref if o[1]["received"] > 100)
    for mode in ("spec", "nospec"):
        got, cube = run(mode)
        for k, ((ga, sa, ra, aa), (gb, sb, rb, ab)) in enumerate(zip(ref, got)):
            assert sa == sb, (mode, k, sa, sb)
            assert ga.tobytes() == gb.tobytes(), (mode, k)
            assert np.array_equal(ra["slots"], rb["slots"]) and ra["results"].tobytes() == rb["results"].tobytes() and np.array_equal(ra["path"], rb["path"]), (mode, k)
            for f in ("results", "delay", "phase", "pathMatch"):
                assert aa[f].tobytes() == ab[f].tobytes(), (mode, k, f)
        np.testing.assert_allclose(cube, cube_ref, rtol=0, atol=1e-18 + 1e-12 * np.abs(cube_ref).max())      # (atomic adds: order varies)


def test_xcd_affine_sub_orders_are_invisible(rts, oracle, scenes, monkeypatch):
    """RTS_XCD_AFFINE=1 forces the XCD-affine sub-orders (rts_post.hip: the cost order cut into a head-rest segment and one band of
    the lattice per XCD, each drawn through eight counters of its own; k_trace<.., AFFINE>: a wave sweeps every segment, its own
    XCD's band first) onto launches far smaller than the 2^18 tiles they are meant for: another SCHEDULE, the same results --
    every launch index traced exactly once (segment accounting), the received set field by field, pulse after pulse on one handle
    (bands from the launch before last, equal counts before that), whole pulses and an interleaved part, product and counting
    build; and the pulse's received records against the oracle's brute force"""
    c3 = scenes.config3(W=80, detail=0.3, rx_radius=300.0)
    tx = c3["tx"]; n_all = c3["W"] ** 3
    out = {}
    monkeypatch.setenv("RTS_WALK_VERSIONS", "0")      # (the experiment exists for the sorted walk only; the counts below are compared step for step)
    for mode in ("0", "1"):
        monkeypatch.setenv("RTS_XCD_AFFINE", mode)
        for count in (False, True):
            tr = rts.Tracer(c3["W"], c3["max_refl"], 0, c3["smooth"], count_traversal=count)
            tr.set_scene(c3["meshes"]); tr.set_receivers(c3["rx"])
            res = []
            for k in range(5):
                mo = [dict(m, position=tuple(np.add(m["position"], (0.3 * k, 0.05 * k, 0.0)))) for m in c3["motion"]]
                il = (4096, 2, 1) if k == 3 else None
                st = tr.trace(tx["origin"], tx["span"], tx["dir"], mo, ray_first=0, ray_count=n_all, interleave=il)
                res.append((st, tr.received()))
            out[(mode, count)] = res
            tr.close()
    for count in (False, True):
        for k, ((sa, ra), (sb, rb)) in enumerate(zip(out[("0", count)], out[("1", count)])):
            for f in ("rays", "segments", "shaded", "received") + (("node_visits", "tri_tests", "walked_segments") if count else ()):
                assert sa[f] == sb[f], (count, k, f, sa[f], sb[f])
            H.assert_prd_equal(ra["results"], rb["results"], "affine vs global order, pulse %d" % k)
            assert np.array_equal(ra["slots"], rb["slots"]) and np.array_equal(ra["path"], rb["path"]) and ra["rcs_angle"].tobytes() == rb["rcs_angle"].tobytes()
    st, rec = out[("1", False)][4]
    assert st["rays"] == n_all and st["received"] > 500
    mo = [dict(m, position=tuple(np.add(m["position"], (0.3 * 4, 0.05 * 4, 0.0)))) for m in c3["motion"]]
    idx = rec["slots"][:: max(len(rec["slots"]) // 400, 1)].astype(np.int64)
    sc = H.oracle_scene(oracle, c3, mo)
    for i in idx[:400]:
        o = sc.trace(tx["origin"], tx["span"], tx["dir"], c3["W"], c3["max_refl"], 0, c3["smooth"], ray_first=int(i), ray_stride=1, n_rays=1, use_bvh=False)
        j = int(np.searchsorted(rec["slots"], i))
        H.assert_prd_equal(o["results"][:1], rec["results"][j:j + 1], "launch index %d against the brute-force oracle" % i)


def test_host_mirror_equals_the_copy_calls(rts, scenes, monkeypatch):
    """rts_received_prefetch / rts_received_view / rts_finalise_values / rts_aggregate / rts_aggregated_view -- the C++ adapter's
    per-pulse path: the received set stored into pinned host memory by a kernel behind the trace (on the device-side count from
    the handle's second pulse on), the simulator's per-ray power / Doppler sent back, per-ray aggregation outputs read from the
    mirror -- against rts_get_received + rs::kernel_wrapper (the reference's boundary, aggregation.cuh:19-22) on a second handle:
    same bits, for a first pulse (no history: blocking path feeds the mirror), ~1 700-ray pulses (speculative), a pulse of three
    launch indices, pulses beyond the mirror's 4 096 rows (served by copies) before and after small ones, RTS_SPECULATE=0"""
    c3 = scenes.config3(W=64, detail=0.3, rx_radius=300.0)
    tx = c3["tx"]; n_all = c3["W"] ** 3
    cs, fc = 299792458.0, 1.0e10

    def prefix_with(lo_R, hi_R):
        tr = H.gpu_tracer(rts, c3); lo, hi = 1, n_all
        for _ in range(40):
            mid = (lo + hi) // 2
            _, st = H.gpu_trace(rts, c3, tr=tr, ray_first=0, ray_count=mid)
            if st["received"] > hi_R: hi = mid
            elif st["received"] < lo_R: lo = mid
            else: tr.close(); return mid
        raise AssertionError("no prefix")
    small, big = prefix_with(1500, 1800), prefix_with(5000, 9000)
    plan = [small, small, 3, big, small, big, big, small]

    def callbacks(rec):
        """stand-in for the simulator's RCS / gain callbacks: any deterministic function of the received records"""
        r = rec["results"]; ang = rec["rcs_angle"]
        w = 1.0 + 0.25 * np.cos(np.where(rec["path"] >= 0, ang[..., 0], 0.0)).sum(axis=1) + 0.1 * np.sin(r["firstHitPoint"][:, 1])
        vr = r["doppler"] / 2
        return r["power"] * w * 9.0e-4, fc * (((1 + vr / cs) / (1 - vr / cs)) - 1)

    ref = H.gpu_tracer(rts, c3)
    want = []
    for count in plan:
        ref.trace(tx["origin"], tx["span"], tx["dir"], c3["motion"], ray_first=0, ray_count=count)
        rec = ref.received()
        pw, dp = callbacks(rec)
        res = rec["results"].copy(); res["power"] = pw; res["doppler"] = dp
        want.append((rec, rts.kernel_wrapper(res, rec["path"], cs, fc, n_all)))
    ref.close()
    assert max(len(w[0]["results"]) for w in want) > 4096 > min(len(w[0]["results"]) for w in want if len(w[0]["results"]) > 100)
    for spec in ("1", "0"):
        monkeypatch.setenv("RTS_SPECULATE", spec)
        tr = H.gpu_tracer(rts, c3)
        for k, count in enumerate(plan):
            tr.trace_begin(tx["origin"], tx["span"], tx["dir"], c3["motion"], ray_first=0, ray_count=count)
            tr.received_prefetch()
            rec = tr.received_view()
            wrec, wagg = want[k]
            H.assert_prd_equal(rec["results"], wrec["results"], "received records (%s, pulse %d)" % (spec, k))      # (field by field: the records' padding bytes are whatever the two handles' buffers held)
            assert np.array_equal(rec["path"], wrec["path"]) and np.array_equal(rec["slots"], wrec["slots"]), (spec, k)
            assert rec["rcs_angle"].tobytes() == wrec["rcs_angle"].tobytes(), (spec, k)
            assert tr.received_count() == len(wrec["results"])
            pw, dp = callbacks(rec)
            tr.finalise_values(pw, dp)
            tr.aggregate(cs, fc, 0, fetch=False)
            agg = tr.aggregated_view()
            assert agg["power"].tobytes() == wagg["results"]["power"].tobytes() and agg["doppler"].tobytes() == wagg["results"]["doppler"].tobytes(), (spec, k)
            assert agg["delay"].tobytes() == wagg["delay"].tobytes() and agg["phase"].tobytes() == wagg["phase"].tobytes() and np.array_equal(agg["pathMatch"], wagg["pathMatch"]), (spec, k)
            # a SECOND view after finalise + aggregate + aggregated view: still the records AS RECEIVED (rts_amd.h), also for a set beyond the
            # mirror's 4 096 rows, which is served from copies -- made once per pulse, in storage rts_aggregated_view does not touch
            rec2 = tr.received_view()
            H.assert_prd_equal(rec2["results"], wrec["results"], "received records, second view (%s, pulse %d, %d rays)" % (spec, k, len(wrec["results"])))
            assert np.array_equal(rec2["path"], wrec["path"]) and np.array_equal(rec2["slots"], wrec["slots"]) and rec2["rcs_angle"].tobytes() == wrec["rcs_angle"].tobytes(), (spec, k)
            again = tr.received()                       # the copy call after the aggregation: the rays carry the group values now (as after rs::kernel_wrapper)
            assert again["results"]["power"].tobytes() == wagg["results"]["power"].tobytes(), (spec, k)
        tr.close()


def test_pulse_end_uniform_with_wide_keys(rts, scenes, monkeypatch):
    """a (receiver, path) aggregation key beyond 64 bits -- 16 bounces among 8 targets: 16 x 4 + 1 = 65 bits -- is sorted as two
    words by the GENERAL aggregation chain, whose kernels take the received count from the host: rts_trace_pulse_end_uniform must
    not enqueue that chain on the device-side count (it then ran over its 2 048-row capacity and exported groups made
    of stale rows).  Same bits as the four calls, pulse after pulse, with RTS_SPECULATE 1 and 0"""
    sv, st_, sn = rts.sphere_mesh(1, 3.0)
    meshes, motion = [], []
    for i in range(8):
        a = 2.0 * math.pi * i / 8.0
        meshes.append(dict(tris=st_, verts=sv, normals=sn, refl_coeff=0.95, refr_index=1.0))
        motion.append(dict(position=(9.0 * math.cos(a), 9.0 * math.sin(a), 1.5 * (i % 3) - 1.5), velocity=(3.0 * i, -2.0 * i, 0.5)))
    rx = [scenes._rx_at((-200.0, 60.0, 0.0), (0, 0, 0), 12.0, 2.6), scenes._rx_at((-150.0, 130.0, 10.0), (0, 0, 0), 12.0, 2.6)]
    spec = dict(name="eight-spheres", W=40, max_refl=16, smooth=True, n_pulses=1, meshes=meshes, motion=motion,
                tx=dict(origin=(-200.0, 0.0, 0.0), span=(0.14, 0.14, 0.05), dir=(0.0, 0.0)), rx=rx, carrier=1.0e10, c=299792458.0)
    tx = spec["tx"]; cs, fc, wl = 299792458.0, 1.0e10, 0.03

    def run(mode):
        monkeypatch.setenv("RTS_SPECULATE", "0" if mode == "nospec" else "1")
        tr = H.gpu_tracer(rts, spec)
        out = []
        for k in range(4):
            mo = [dict(m, position=tuple(np.add(m["position"], (0.05 * k, 0.0, 0.0)))) for m in motion]
            tr.trace_begin(tx["origin"], tx["span"], tx["dir"], mo)
            if mode == "four":
                tr.trace_end(); tr.finalise_uniform(None, wl, 1.0, 1.0, fc, cs); g = tr.aggregate(cs, fc)
            else:
                tr.trace_end_uniform(None, wl, 1.0, 1.0, fc, cs); g = tr.groups()
            out.append((g, tr.stats()["received"], tr.received(), tr.aggregated()))
        tr.close()
        return out
    ref = run("four")
    assert all(0 < o[1] < 1500 for o in ref), [o[1] for o in ref]                      # small enough that a narrow key WOULD speculate
    assert max(int((o[2]["path"] >= 0).sum(axis=1).max()) for o in ref) >= 2            # multi-bounce paths among the targets
    for mode in ("spec", "nospec"):
        for k, ((ga, na, ra, aa), (gb, nb, rb, ab)) in enumerate(zip(ref, run(mode))):
            assert na == nb and ga.tobytes() == gb.tobytes(), (mode, k, na, nb, len(ga), len(gb))
            assert ra["results"].tobytes() == rb["results"].tobytes() and np.array_equal(ra["path"], rb["path"]), (mode, k)
            for f in ("results", "delay", "phase", "pathMatch"):
                assert aa[f].tobytes() == ab[f].tobytes(), (mode, k, f)


def test_asynchronous_bounces_are_invisible(rts, scenes, monkeypatch):
    """RTS_ASYNC_IDLE0 > 0 selects the kernel whose lanes advance from segment to segment on their own
    (rts_trace_unit_async: a walk phase ends as soon as `idle` lanes have come out of their walks; they are shaded and
    re-join the lanes still walking, whose walk state stays in registers / LDS).  Which lanes are advanced together depends on
    the limit and on timings (the limit switches with the tile's age), the results must not: lock step (the default kernel)
    against limits 64, 8 and 1, in KEEP_ALL + counting builds and in the product build, several targets (a lane may be at
    another target than its neighbour), a 3-entry LDS stack, Earth-centred coordinates, the miss-branch scene"""
    import math
    monkeypatch.setenv("RTS_GRID_MULT", "1"); monkeypatch.setenv("RTS_COOP_FRAC", "0")       # (no cooperative units: the node visits are compared)
    c3 = scenes.config3(W=56, detail=0.3, rx_radius=300.0)
    multi = scenes.config_multi(W=44)
    cases = [("c3", c3, {}), ("c3 ecef", scenes.translate(c3, scenes.ecef_offset(lat=math.pi / 2)), {}), ("multi", multi, {}),
             ("c3 short stack", c3, {"RTS_STACK_LDS_DEBUG": "3"}), ("miss branches", scenes.config_miss_branches(W=44), {})]
    for name, spec, env in cases:
        n = spec["W"] ** 3
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = {}
        monkeypatch.setenv("RTS_WALK_VERSIONS", "0")      # (the asynchronous schedule exists for the sorted walk only; its counts are compared step for step with the lock-step kernel's)
        for mode in ("0", "64", "8", "1"):
            monkeypatch.setenv("RTS_ASYNC_IDLE0", mode); monkeypatch.setenv("RTS_ASYNC_IDLE1", "1" if mode == "64" else mode if mode != "0" else "1"); monkeypatch.setenv("RTS_ASYNC_AGE", "40")
            tr = H.gpu_tracer(rts, spec, keep_all=True, count_traversal=True)
            tp = H.gpu_tracer(rts, spec)                                        # the product build
            for rep in range(2):
                _, st = H.gpu_trace(rts, spec, tr=tr); _, sp = H.gpu_trace(rts, spec, tr=tp)
            out[mode] = (tr.all_rays(n), tr.received(), st, tp.received(), sp)
            tr.close(); tp.close()
        for k in env:
            monkeypatch.delenv(k)
        a, ra, sa, pa, spa = out["0"]
        for mode in ("64", "8", "1"):
            b, rb, sb, pb, spb = out[mode]
            _all_equal(a, b, "%s, idle limit %s" % (name, mode))
            for x, y, what in ((ra, rb, "counting build"), (pa, pb, "product build"), (ra, pb, "product vs counting")):
                assert np.array_equal(x["slots"], y["slots"]) and np.array_equal(x["path"], y["path"]), (name, mode, what)
                H.assert_prd_equal(x["results"], y["results"], "%s (received, %s, idle limit %s)" % (name, what, mode))
                np.testing.assert_allclose(x["rcs_angle"], y["rcs_angle"], rtol=0, atol=1e-12)
            # the same walks, step for step: only their interleaving differs
            assert (sa["segments"], sa["shaded"], sa["received"], sa["node_visits"], sa["tri_tests"]) == (sb["segments"], sb["shaded"], sb["received"], sb["node_visits"], sb["tri_tests"]), (name, mode)
            assert (spb["segments"], spb["shaded"], spb["received"]) == (sa["segments"], sa["shaded"], sa["received"]), (name, mode)
        assert sa["received"] > 0 and sa["tri_tests"] >= sa["shaded"] > 0, name


def test_primary_prefilter_switches_off_when_most_rays_hit(rts, scenes):
    """a handle whose last launch shaded more hits than half its launch indices runs the next one without the filter
    (it would cost every ray and skip none) -- seen through the node visits of the counting build; results unchanged"""
    c3 = scenes.config3(W=48, detail=0.3, rx_radius=300.0)
    dense = dict(c3, tx=dict(c3["tx"], span=(0.004, 0.004, 0.1)))
    tr = H.gpu_tracer(rts, dense, keep_all=True, count_traversal=True)
    _, s1 = H.gpu_trace(rts, dense, tr=tr); a = tr.all_rays(dense["W"] ** 3)
    assert 2 * s1["shaded"] > dense["W"] ** 3
    _, s2 = H.gpu_trace(rts, dense, tr=tr); b = tr.all_rays(dense["W"] ** 3)
    _all_equal(a, b, "dense, filter off")
    assert s2["node_visits"] >= s1["node_visits"] and s2["segments"] == s1["segments"]
    # back on a sparse launch of the same handle: first launch still unfiltered, second filtered
    tx = c3["tx"]
    st = [tr.trace(tx["origin"], tx["span"], tx["dir"], c3["motion"]) for _ in range(3)]
    assert st[0]["node_visits"] > st[1]["node_visits"] == st[2]["node_visits"] and st[0]["segments"] == st[2]["segments"]
    tr.close()


def test_differential_fuzz_seeds(rts):
    """tools/fuzz_equal.py on a few seeds (random soups / beams / receivers near the origin, far away and at Earth-centred
    coordinates): host and device trees, with and without slab references, with and without the primary-ray pre-filter
    must produce the same bits.  Seed 12661 is the scene that showed the receiver pre-filter must be conservative with
    respect to the REFERENCE'S cancelling quadratic, not to geometry (a 0.86 m sphere 3.7 m from a transmitter at 6.4e6 m);
    seed 50301 the one whose beam start has a cosine on which glibc's sincos and cos differ (oracle/Makefile).  Each scene is
    also compared with the oracle's brute force."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_equal as F
    for seed, version in ((12661, 1), (7, 1), (1234, 1), (50301, 2), (20011, 2), (31337, 3), (200003, 3), (200040, 3)):
        spec, place, aim = F.random_scene(seed, version)
        a = F.run(spec); b = F.run(spec, pre_filter=False); c = F.run(spec, device_build=False)
        F.same(a, b, "seed %d: pre-filter on / off" % seed)
        F.same(a, c, "seed %d: host / device tree" % seed)
        F.against_oracle(spec, a)


def test_block_timeline_of_a_product_launch(rts, scenes, monkeypatch):
    """rts_get_block_timeline (handles created with RTS_TIMELINE_BLOCKS=1): when the persistent blocks of the last launch started and ended -- ordered, inside the
    launch's event time, one entry per block; the results are the bits of a handle without it; a handle that records none says so"""
    spec = scenes.config3(W=56, detail=0.3, rx_radius=300.0)
    plain = H.gpu_tracer(rts, spec)
    _, st0 = H.gpu_trace(rts, spec, tr=plain); want = plain.received()
    from rts_amd import _lib
    with pytest.raises(_lib.RtsError):
        plain.block_timeline()
    monkeypatch.setenv("RTS_TIMELINE_BLOCKS", "1")
    tr = H.gpu_tracer(rts, spec)
    monkeypatch.delenv("RTS_TIMELINE_BLOCKS")
    for rep in range(2):
        _, st = H.gpu_trace(rts, spec, tr=tr)
        b = tr.block_timeline()
        assert b["blocks"] >= 1 and 0.0 == b["start_first"] <= b["start_p50"] <= b["start_last"], b
        assert 0.0 < b["end_first"] <= b["end_p10"] <= b["end_p50"] <= b["end_p90"] <= b["end_last"] <= st["ms_trace"] * 1e3 + 50.0, (b, st["ms_trace"])
    got = tr.received()
    assert np.array_equal(got["slots"], want["slots"]) and np.array_equal(got["path"], want["path"])
    H.assert_prd_equal(got["results"], want["results"], "block timeline on / off")
    assert (st["segments"], st["shaded"], st["received"]) == (st0["segments"], st0["shaded"], st0["received"])
    tr.close(); plain.close()


def test_monostatic_tiles_are_dead_at_tile_level(rts, oracle, scenes, monkeypatch):
    """round 5: the transmitter ON the capture sphere of a receiver that looks back at it (BASELINE configs[4]'s monostatic radar): every primary ray has a root at
    t ~ 0 that the reference ignores (t > SCENE_EPS) and one on the far side, outside the window.  The tile-level screen has to see that for a whole wave tile's
    bundle of directions (the small root through the product of the roots, rts_rx_maybe) -- else every tile of the pulse is looked at ray by ray.  Same received set
    with the batches on / off / every position screened tile-wise first, against the oracle's brute force for the received rays, and most tiles dead at tile level"""
    spec = scenes.config3(W=80, detail=0.3, n_rx=1, rx_radius=250.0)
    tx = spec["tx"]; n = spec["W"] ** 3
    assert abs(np.linalg.norm(np.subtract(spec["rx"][0]["centre"], tx["origin"])) - spec["rx"][0]["radius"]) < 1e-9      # ON the sphere
    monkeypatch.setenv("RTS_GRID_MULT", "1")
    out = {}
    for mode in ("0", "1", "all"):
        monkeypatch.setenv("RTS_DEAD_BATCH", mode)
        tr = H.gpu_tracer(rts, spec)
        for rep in range(3):
            _, st = H.gpu_trace(rts, spec, tr=tr)
        out[mode] = (tr.received(), st, tr.tile_records_get())
        tr.close()
    monkeypatch.delenv("RTS_DEAD_BATCH")
    for mode in ("1", "all"):
        a, b = out["0"], out[mode]
        assert np.array_equal(a[0]["slots"], b[0]["slots"]) and np.array_equal(a[0]["path"], b[0]["path"]), mode
        H.assert_prd_equal(a[0]["results"], b[0]["results"], "monostatic, batches %s" % mode)
        assert (a[1]["segments"], a[1]["shaded"], a[1]["received"]) == (b[1]["segments"], b[1]["shaded"], b[1]["received"]), mode
        assert np.array_equal(a[2] == 1, b[2] == 1), mode                                  # the same tiles are dead
    rec, st, tiles = out["all"]
    assert st["received"] > 0 and (tiles == 1).sum() > 0.5 * len(tiles), (st["received"], (tiles == 1).sum(), len(tiles))
    sc = H.oracle_scene(oracle, spec)
    for i in rec["slots"][:: max(len(rec["slots"]) // 60, 1)].astype(np.int64)[:60]:
        o = sc.trace(tx["origin"], tx["span"], tx["dir"], spec["W"], spec["max_refl"], 0, spec["smooth"], ray_first=int(i), ray_stride=1, n_rays=1, use_bvh=False)
        j = int(np.searchsorted(rec["slots"], i))
        H.assert_prd_equal(o["results"][:1], rec["results"][j:j + 1], "launch index %d of the monostatic pulse against the brute-force oracle" % i)
