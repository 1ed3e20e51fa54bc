#!/usr/bin/env python3
"""Differential fuzz on the GPU: results must not depend on HOW a launch is traced.  For every random scene the full
per-launch-index buffers (KEEP_ALL: payload, paths, RCS angles, per-segment hit primitive and f32 t), the received set and
its order, and the segment / shaded / received counts are compared between
    host SAH tree + primary-ray pre-filter (the default)   |   host SAH tree, no pre-filter
    device LBVH with slab references + pre-filter          |   device LBVH, one reference per triangle, no pre-filter
(the hierarchy and the filter may only change how much work is done).  Scenes: triangle soups (ordinary triangles, slivers,
fans, duplicates, degenerate triangles) as one to three targets, placed near the origin, far from it or at Earth-centred
coordinates, some behind or around the transmitter; beams from 1e-4 to 3 rad wide, pointed at / beside / away from the
targets; 0-6 capture spheres, some containing the transmitter; W = 6..40; reflection depth 0..6; smooth / flat normals;
optionally the refraction branch.
   python tools/fuzz_equal.py [n_scenes] [seed0] [--oracle]  (about 0.1 s per scene on an MI355X; --oracle adds the comparison of
                                                              the default launch with the CPU restatement's brute force)
The counting builds of the same two host-tree launches say in how many scenes the pre-filter actually removed work."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rts_amd import api, scenes  # noqa: E402
import helpers as H  # noqa: E402


def soup(rng, n_reg, n_sliver, n_fan, scale):
    tris = []
    for _ in range(n_reg):
        c = rng.normal(0, 6.0 * scale, 3); s = scale * 10 ** rng.uniform(-1.5, 0.7)
        tris.append(c + rng.normal(0, s, (3, 3)))
    for _ in range(n_sliver):
        a = rng.normal(0, 6.0 * scale, 3); d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
        L = scale * rng.uniform(5, 25); w = rng.normal(0, 1, 3) * scale * 10 ** rng.uniform(-4, -1.5)
        tris.append(np.stack([a, a + L * d, a + 0.5 * L * d + w]))
    if n_fan:
        hub = rng.normal(0, 3.0 * scale, 3)
        ang = np.sort(rng.uniform(0, 2 * np.pi, n_fan + 1)); rad = scale * rng.uniform(2, 8)
        e1 = np.array([0.0, 1.0, 0.2]); e2 = np.array([0.1, -0.2, 1.0])
        for i in range(n_fan):
            tris.append(np.stack([hub, hub + rad * (np.cos(ang[i]) * e1 + np.sin(ang[i]) * e2), hub + rad * (np.cos(ang[i + 1]) * e1 + np.sin(ang[i + 1]) * e2)]))
    tris.append(tris[0].copy())
    p = rng.normal(0, 5.0 * scale, 3); tris.append(np.stack([p, p, p + 1.0]))
    v = np.concatenate(tris).astype(np.float64)
    t = np.arange(len(v), dtype=np.uint32).reshape(-1, 3)
    nrm = np.repeat(np.cross(v[1::3] - v[0::3], v[2::3] - v[0::3]) + 1e-30, 3, axis=0)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return dict(tris=t, verts=v, normals=nrm, refl_coeff=float(rng.choice([0.8, -0.6, 1.0, 0.3])), refr_index=float(rng.choice([1.0, 1.3, 2.0])))


def random_scene(seed, version=4, big=False):
    """versions 1, 2: the generator as it was when the regression seeds of tests/test_gpu_parity.py were found (1: as first
    written; 2: + the scenes biased towards an active pre-filter; 3: + many targets / receivers, NaN vertices, W = 1, deep chains;
    4 (round 5): + receivers the pre-filter's WINDOW screen has to judge -- capture spheres the beam crosses, the transmitter on or near their
    surface (a monostatic radar's), windows narrow enough for the screen to apply and aimed so that their EDGES pass near the crossing points)"""
    rng = np.random.default_rng(seed)
    v2 = version >= 3; friendly = version >= 2
    place = rng.choice(["origin", "far", "ecef"])
    off = {"origin": np.zeros(3), "far": rng.normal(0, 4.0e4, 3), "ecef": scenes.ecef_offset(lat=rng.uniform(-1.5, 1.5), lon=rng.uniform(-3, 3))}[place]
    n_t = int(rng.integers(1, 4)); scale = 10 ** rng.uniform(-0.5, 0.5)
    many_targets = v2 and rng.random() < 0.08
    if many_targets:
        n_t = int(rng.integers(8, 40))                                # the linear loop over the targets' hierarchies
    meshes, motion = [], []
    for k in range(n_t):
        if many_targets:
            meshes.append(soup(rng, int(rng.integers(1, 12)), int(rng.integers(0, 4)), int(rng.integers(0, 5)), scale))
        else:
            k_big = 12 if big else 1                                   # --big: soups of a few thousand triangles
            meshes.append(soup(rng, int(rng.integers(5, 120)) * k_big, int(rng.integers(0, 40)) * k_big, int(rng.integers(0, 50)) * k_big, scale))
        if v2 and rng.random() < 0.05:                                # a vertex that is not a number: its triangles can never be hit
            meshes[-1]["verts"][int(rng.integers(0, len(meshes[-1]["verts"])))] = np.nan
        pos = off + rng.normal(0, 15.0 * scale, 3)
        m = dict(position=tuple(pos), velocity=tuple(rng.normal(0, 5.0, 3)))
        if rng.random() < 0.6:
            m["rotation"] = api.rotation_matrix(*rng.uniform(-3, 3, 3))
        motion.append(m)
    dist = scale * 10 ** rng.uniform(0.3, 3.3)                        # 2 m .. 2 km (x scale): inside the soup up to far away
    dirv = rng.normal(0, 1, 3); dirv /= np.linalg.norm(dirv)
    origin = off - dist * dirv + rng.normal(0, 2.0 * scale, 3)
    az = math.atan2(dirv[1], dirv[0]); el = math.asin(max(-1.0, min(1.0, dirv[2])))
    aim = rng.choice(["at", "beside", "away"], p=[0.7, 0.2, 0.1])
    if aim == "beside":
        az += rng.uniform(-0.5, 0.5); el += rng.uniform(-0.3, 0.3)
    if aim == "away":
        az += math.pi
    span_w = 10 ** rng.uniform(-4, 0.45)
    if friendly and rng.random() < 0.5:                              # half of the scenes: where the pre-filter has work to do --
        dist = scale * 10 ** rng.uniform(2.0, 3.5)                    # targets well away (no triangle covers thousands of mask cells),
        origin = off - dist * dirv + rng.normal(0, 2.0 * scale, 3)
        span_w = min(2.0, (30.0 * scale / dist) * 10 ** rng.uniform(0.2, 1.2))    # the beam a few times wider than the targets
    tx = dict(origin=tuple(origin), span=(span_w, span_w * rng.uniform(0.3, 1.0), float(rng.uniform(0.0, 0.3))), dir=(az, el))
    rx = []
    for _ in range(int(rng.integers(17, 48)) if (v2 and rng.random() < 0.08) else int(rng.integers(0, 7))):      # (> 16 receivers: beyond the LDS copy, no pre-filter)
        kind = rng.random()
        if kind < 0.25:
            c = origin + rng.normal(0, 1.0, 3) * rng.uniform(0.1, 30.0); r = float(np.linalg.norm(c - origin) * rng.uniform(1.1, 3.0))     # contains the transmitter
        else:
            c = off + rng.normal(0, 1.0, 3) * dist * rng.uniform(0.2, 2.0); r = float(dist * 10 ** rng.uniform(-2, -0.3))
        th0 = rng.uniform(-3.2, 3.2); ph0 = rng.uniform(-1.6, 1.6)
        if version >= 4 and rng.random() < 0.6:
            # a sphere in the path of the boresight ray, and a window whose edge lies near one of that ray's crossing points
            bore = np.array([math.cos(az) * math.cos(el), math.sin(az) * math.cos(el), math.sin(el)])
            how = rng.random()
            r = float(10 ** rng.uniform(-0.5, 2.5) * scale)
            if how < 0.35:                                             # monostatic: the sphere touches the transmitter (ray_tracer.cpp:903-905)
                c = np.asarray(origin) + r * bore * (1.0 + (rng.normal(0, 1e-3) if rng.random() < 0.5 else 0.0)) + (rng.normal(0, 0.05 * r, 3) if rng.random() < 0.3 else 0.0)
            elif how < 0.55:                                           # the transmitter inside
                c = np.asarray(origin) + rng.normal(0, 1.0, 3) * r * rng.uniform(0.0, 0.6)
            else:                                                      # ahead, crossed by (part of) the beam at any impact parameter
                perp = np.cross(bore, rng.normal(0, 1, 3)); perp /= max(np.linalg.norm(perp), 1e-12)
                c = np.asarray(origin) + bore * (r + dist * rng.uniform(0.01, 1.5)) + perp * r * rng.uniform(0.0, 1.2)
            m = np.asarray(origin) - c; b = float(-m @ bore); disc = b * b - float(m @ m - r * r)
            if disc > 0:
                t = b + math.sqrt(disc) * (1.0 if (rng.random() < 0.5 or b - math.sqrt(disc) <= 0) else -1.0)
                pnt = m + t * bore
                th0 = math.atan2(pnt[1], pnt[0]); ph0 = math.asin(max(-1.0, min(1.0, pnt[2] / max(np.linalg.norm(pnt), 1e-300))))
            hw_t = 10 ** rng.uniform(-2.0, 0.15); hw_p = 10 ** rng.uniform(-2.0, -0.1)
            th0 += hw_t * rng.choice([0.0, 0.9, 1.0, 1.1, -0.9, -1.0, -1.1, 2.0]) + rng.normal(0, 0.02 * hw_t)      # the window's edge at / near the crossing point
            ph0 += hw_p * rng.choice([0.0, 0.9, 1.0, 1.1, -0.9, -1.0, -1.1, 2.0]) + rng.normal(0, 0.02 * hw_p)
            lo_p, hi_p = ph0 - hw_p, ph0 + hw_p
            if rng.random() < 0.8:
                lo_p, hi_p = max(lo_p, -math.pi / 2), min(hi_p, math.pi / 2)                               # (within the poles: the screen applies)
            if hi_p > lo_p:
                rx.append(scenes.rx_window(tuple(c), r, (th0 - hw_t, th0 + hw_t), (lo_p, hi_p)))
                continue
        rx.append(scenes.rx_window(tuple(c), r, (th0 - rng.uniform(0.1, 3.2), th0 + rng.uniform(0.1, 3.2)), (ph0 - rng.uniform(0.1, 1.7), ph0 + rng.uniform(0.1, 1.7))))
    refr = rng.random() < 0.2
    W = int(rng.integers(6, 24 if refr else 41))
    if big:
        W = int(rng.integers(20, 36 if refr else 65))
    if v2 and rng.random() < 0.03:
        W = 1                                                         # the single boresight ray (ray_tracer.cu:160-161)
    spec = dict(name="fuzz-%d" % seed, W=W, max_refl=int(rng.integers(0, 7)) if (not v2 or rng.random() < 0.9) else int(rng.integers(7, 15)), smooth=bool(rng.integers(0, 2)), n_pulses=1, meshes=meshes, motion=motion,
                tx=tx, rx=rx, carrier=scenes.FC, c=scenes.C0)
    if refr:
        spec["max_refr"] = 1
    return spec, place, aim


def run(spec, **kw):
    tr = H.gpu_tracer(api, spec, keep_all=True, **kw)
    _, st = H.gpu_trace(api, spec, tr=tr)
    out = (tr.all_rays(spec["W"] ** 3), tr.received(), st, tr.scene_info())
    tr.close()
    return out


def moved(spec, rng):
    """the same scene one pulse later: every target displaced and turned a little (or a lot)"""
    sp = dict(spec); sp["motion"] = []
    big = rng.random() < 0.3
    for m in spec["motion"]:
        mm = dict(m); mm["position"] = tuple(np.asarray(m["position"]) + rng.normal(0, 30.0 if big else 0.3, 3))
        if rng.random() < 0.7:
            mm["rotation"] = api.rotation_matrix(*rng.uniform(-3, 3, 3)) if big else api.rotation_matrix(*(rng.normal(0, 0.02, 3)))
        sp["motion"].append(mm)
    return sp


def pulses_on_one_handle(spec, seed):
    """three pulses on ONE handle (tile-cost history, adaptive pre-filter switch, placement buffers and mask carried from
    pulse to pulse) against fresh handles without the pre-filter"""
    rng = np.random.default_rng(seed + 7777777)
    tr = H.gpu_tracer(api, spec, keep_all=True)
    sp = spec
    for k in range(3):
        _, st = H.gpu_trace(api, sp, tr=tr, motion=sp["motion"])
        got = (tr.all_rays(spec["W"] ** 3), tr.received(), st, None)
        if k:
            same(got, run(sp, pre_filter=False), "seed %d: pulse %d on a used handle" % (seed, k))
        sp = moved(sp, rng)
    tr.close()


def cooperative_pulses(spec, seed):
    """round 5: three pulses on ONE handle of the PRODUCT build and of the counting build with every tile that cost anything handed to the cooperative kernel
    (which walks the octant versions in these two builds) -- the received set of pulses 1 and 2 against fresh launches without cooperative units"""
    rng = np.random.default_rng(seed + 31337)
    env = {"RTS_COOP_FRAC": "1e-12", "RTS_COOP_FLOOR": "0", "RTS_COOP_SEG": "0", "RTS_GRID_MULT": "1"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        trp = H.gpu_tracer(api, spec); trc = H.gpu_tracer(api, spec, count_traversal=True)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    sp = spec; coop = 0
    for k in range(3):
        outs = []
        for tr in (trp, trc):
            _, st = H.gpu_trace(api, sp, tr=tr, motion=sp["motion"]); outs.append((tr.received(), st)); coop += st["coop_tiles"]
        if k:
            ref = run(sp, pre_filter=False)
            for (rec, st), what in zip(outs, ("product", "counting")):
                assert np.array_equal(rec["slots"], ref[1]["slots"]) and np.array_equal(rec["path"], ref[1]["path"]), (seed, k, what, "cooperative pulses: rows / paths")
                H.assert_prd_equal(rec["results"], ref[1]["results"], "seed %d: pulse %d, cooperative kernel forced, %s build" % (seed, k, what))
                assert rec["rcs_angle"].tobytes() == ref[1]["rcs_angle"].tobytes(), (seed, k, what)
                assert (st["segments"], st["shaded"], st["received"]) == (ref[2]["segments"], ref[2]["shaded"], ref[2]["received"]), (seed, k, what, st, ref[2])
        sp = moved(sp, rng)
    trp.close(); trc.close()
    return coop


def dealt_parts(spec, seed, whole):
    """the launch split into 2-4 parts by a RANDOM map of plan tiles (rts_set_tile_list, RTS_INTERLEAVE_LIST): the parts' received sets,
    merged by buffer row, and their segment / shaded counts are the whole launch's"""
    rng = np.random.default_rng(seed + 424242)
    n = spec["W"] ** 3; tile = int(rng.choice([64, 128, 256, 4096])); parts = int(rng.integers(2, 5)); n_plan = (n + tile - 1) // tile
    part_of = rng.integers(0, parts, n_plan)
    tr = H.gpu_tracer(api, spec)
    recs = []; seg = sh = 0
    for p in range(parts):
        ids = np.flatnonzero(part_of == p).astype(np.uint32)
        if ids.shape[0] == 0: continue
        tr.set_tile_list(tile, ids)
        _, st = H.gpu_trace(api, spec, tr=tr, interleave=(tile, api.INTERLEAVE_LIST, 0))
        assert st["rays"] == sum(min(tile, n - int(i) * tile) for i in ids), (seed, "rays of a dealt part")
        recs.append(tr.received()); seg += st["segments"]; sh += st["shaded"]
    tr.close()
    ra = whole[1]
    slots = np.concatenate([r["slots"] for r in recs]); order = np.argsort(slots, kind="stable")
    assert np.array_equal(slots[order], ra["slots"]), (seed, "dealt parts: received rows")
    H.assert_prd_equal(np.concatenate([r["results"] for r in recs])[order], ra["results"], "seed %d: dealt parts (received)" % seed)
    assert np.array_equal(np.concatenate([r["path"] for r in recs])[order], ra["path"]), (seed, "dealt parts: paths")
    assert (seg, sh) == (whole[2]["segments"], whole[2]["shaded"]), (seed, "dealt parts: counts", seg, sh, whole[2])


def same(a, b, what):
    (ga, ra, sa, _), (gb, rb, sb, _) = a, b
    H.assert_prd_equal(ga["results"], gb["results"], what)
    for k in ("path", "rcs_angle", "hit_prim"):
        assert np.array_equal(ga[k], gb[k]), (what, k)
    assert np.array_equal(ga["hit_t"].view(np.uint32), gb["hit_t"].view(np.uint32)), (what, "hit_t")
    assert np.array_equal(ra["slots"], rb["slots"]) and np.array_equal(ra["path"], rb["path"]), (what, "received order")
    H.assert_prd_equal(ra["results"], rb["results"], what + " (received)")
    assert (sa["segments"], sa["shaded"], sa["received"]) == (sb["segments"], sb["shaded"], sb["received"]), (what, "counts")


def against_oracle(spec, a):
    """the default GPU launch against the CPU restatement in BRUTE-FORCE mode (every triangle tested for every segment)"""
    from oracle import oracle as O
    rows = spec["max_refl"] + 3 if spec.get("max_refr", 0) else 1
    o = H.oracle_trace(O, spec, use_bvh=False, threads=min(32, os.cpu_count() or 1))
    H.compare_full(o, a[0], spec["W"] ** 3 * rows)
    assert a[2]["segments"] == o["counters"]["segments"] and a[2]["shaded"] == o["counters"]["shaded"], (a[2], o["counters"])


def main():
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    big = "--big" in sys.argv                                        # thousands of triangles per target, W up to 64
    with_oracle = "--oracle" in sys.argv                             # also: the default launch against the oracle's brute force (CPU, slower)
    n = int(args[0]) if len(args) > 0 else 200
    seed0 = int(args[1]) if len(args) > 1 else 1
    tot = dict(filter_engaged=0, scenes=0, rays=0, segments=0, shaded=0, received=0, refs_host=0, refs_dev=0, prims=0)
    kinds = {}
    for seed in range(seed0, seed0 + n):
        spec, place, aim = random_scene(seed, big=big)
        try:
            os.environ.pop("RTS_SPLIT_BUDGET", None)
            a = run(spec)
            b = run(spec, pre_filter=False)
            c = run(spec, device_build=False)
            os.environ["RTS_SPLIT_BUDGET"] = "0"
            d = run(spec, device_build=False, pre_filter=False)
            os.environ.pop("RTS_SPLIT_BUDGET", None)
            same(a, b, "seed %d: pre-filter on / off" % seed)
            os.environ["RTS_RX_WINDOW_SCREEN"] = "0"                  # round 5: the pre-filter without its window screen (it then only asks whether a capture sphere is reached)
            h = run(spec)
            os.environ.pop("RTS_RX_WINDOW_SCREEN", None)
            same(a, h, "seed %d: receiver window screen on / off" % seed)
            os.environ["RTS_WALK_VERSIONS"] = "0"                    # round 5: the default walks the octant versions of the node records; here the role fetch + sorted children
            g = run(spec); g2 = run(spec, device_build=False, pre_filter=False) if seed % 2 else run(spec, count_traversal=True)
            os.environ.pop("RTS_WALK_VERSIONS", None)
            same(a, g, "seed %d: octant versions / sorted walk" % seed); same(a, g2, "seed %d: octant versions / sorted walk (second way)" % seed)
            e = run(spec, count_traversal=True); f = run(spec, count_traversal=True, pre_filter=False)     # counting builds: is the filter doing anything?
            same(a, e, "seed %d: counting build" % seed); same(a, f, "seed %d: counting build, no pre-filter" % seed)
            assert e[2]["tri_tests"] <= f[2]["tri_tests"] and e[2]["node_visits"] <= f[2]["node_visits"], (seed, e[2], f[2])       # (the filter only ever REMOVES visits)
            tot["filter_engaged"] += 1 if e[2]["node_visits"] < f[2]["node_visits"] else 0
            same(a, c, "seed %d: host / device tree" % seed)
            same(a, d, "seed %d: host / device tree without references" % seed)
            if with_oracle:
                against_oracle(spec, a)
            if seed % 4 == 0:
                pulses_on_one_handle(spec, seed)
            if seed % 3 == 0:
                dealt_parts(spec, seed, a)
            if seed % 2 == 1:
                tot["coop_tiles"] = tot.get("coop_tiles", 0) + cooperative_pulses(spec, seed)
        except Exception as e:
            print("FAILED seed %d (%s, aimed %s, W=%d, refl=%d, refr=%s): %r" % (seed, place, aim, spec["W"], spec["max_refl"], "max_refr" in spec, e), flush=True)
            raise
        st = a[2]
        tot["scenes"] += 1; tot["rays"] += spec["W"] ** 3; tot["segments"] += st["segments"]; tot["shaded"] += st["shaded"]; tot["received"] += st["received"]
        tot["refs_host"] += a[3]["n_leaves"]; tot["refs_dev"] += c[3]["n_leaves"]; tot["prims"] += a[3]["n_prims"]
        kinds[(place, aim)] = kinds.get((place, aim), 0) + 1
        if tot["scenes"] % 25 == 0:
            print("%d scenes ok: %s" % (tot["scenes"], tot), flush=True)
    print("ALL EQUAL: %s" % tot)
    print("placement x aim:", dict(sorted((("%s/%s" % k), v) for k, v in kinds.items())))


if __name__ == "__main__":
    main()
