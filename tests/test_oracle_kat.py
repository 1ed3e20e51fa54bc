"""Known-answer tests that pin the CPU oracle to the reference's formulas.

The reference ships no tests or golden vectors; these closed-form cases are derived from the
cited source lines and are what "pins" the oracle (parity status: unpinned by reference
fixtures, pinned by KATs).  All run on the CPU.
"""
import math

import numpy as np
import pytest

C0 = 299792458.0
FOUR_PI = 4 * math.pi


def _monostatic_rx(O, pos=(0.0, 0.0, 0.0), radius=10.0):
    return O.rx_sphere(pos, 0.0, 0.0, radius, math.pi / 2, math.pi / 2)


def test_prd_layout(oracle):
    # ray_tracer.h:13-28 compiled with hipcc: sizeof 144, offsets below
    d = oracle.PRD_DTYPE
    assert d.itemsize == 144
    assert [d.fields[n][1] for n in d.names] == [0, 16, 32, 36, 40, 48, 72, 96, 120, 128, 136, 140]


def test_kat_direct_path(oracle):
    """no targets, W = 1 aimed at a receiver: power = 1/((4 pi)^2 R^2), rayLength = t (ray_tracer.cu:410-417)"""
    sc = oracle.Scene()
    rx = oracle.rx_sphere((1000.0, 0.0, 0.0), math.pi, 0.0, 10.0, math.pi / 2, math.pi / 2)   # looks back at the tx
    sc.set_receivers([rx])
    r = sc.trace((0, 0, 0), (0.02, 0.02, 0.0), (0.0, 0.0), 1, 2)
    p = r["results"][0]
    assert p["received"] == 0 and p["reflDepth"] == 0
    assert p["rayLength"] == pytest.approx(1000.0, rel=1e-12)
    assert p["power"] == pytest.approx(1.0 / (FOUR_PI ** 2 * 1000.0 ** 2), rel=1e-12)
    assert p["doppler"] == 0.0
    assert (r["path"] == -1).all() and (r["rcs_angle"] == -1000000).all()


def test_kat_plate_normal_incidence_and_doppler(oracle):
    """plate at range R moving at v along the line of sight (normal_shader.cu:163,298,314; ray_tracer.cu:422)"""
    v, t, n = oracle.rect_mesh(0.1, 10.0, 10.0)
    sc = oracle.Scene()
    vel = 15.0
    sc.add_mesh(t, v + np.array([1000.0, 0, 0]), n, refl_coeff=0.9, vel=(vel, 0, 0))
    sc.set_receivers([_monostatic_rx(oracle)])
    r = sc.trace((0, 0, 0), (0.02, 0.02, 0.0), (0.0, 0.0), 1, 1)
    p = r["results"][0]
    R = 1000.0 - 0.05
    t1 = float(np.float32(R))                                    # hit distance is narrowed to f32 (normal_shader.cu:150-153)
    assert p["received"] == 0 and p["reflDepth"] == 1
    assert p["rayLength"] == pytest.approx(t1 + t1, rel=1e-12)   # back to the rx position on the far side of the sphere
    assert p["power"] == pytest.approx(0.9 / (FOUR_PI * t1 ** 2) / (FOUR_PI ** 2 * t1 ** 2), rel=1e-9)
    assert p["doppler"] == pytest.approx(-2.0 * vel, rel=1e-12)  # V.(k1 - k0) = -2 v
    assert r["path"][0, 0] == 0
    assert r["hit_t"][0, 0] == np.float32(R) and r["hit_prim"][0, 0] >= 0 and r["hit_prim"][0, 1] == -1
    # host Doppler conversion, ray_tracer.cpp:1252-1253
    fc = 10e9
    rx, rxi, slots = oracle.filter_finalise(r["results"], r["path"], [1.0], C0 / fc, 1.0, 1.0, fc, C0)
    Vr = -vel
    assert rx["doppler"][0] == pytest.approx(fc * ((1 + Vr / C0) / (1 - Vr / C0) - 1), rel=1e-12)
    assert rx["power"][0] == pytest.approx(p["power"] * (C0 / fc) ** 2, rel=1e-12)


def test_kat_absorbed_second_hit(oracle):
    """two facing plates, maxRefl = 1: the second geometry hit is gated out (normal_shader.cu:134), never received"""
    v, t, n = oracle.rect_mesh(0.1, 10.0, 10.0)
    sc = oracle.Scene()
    sc.add_mesh(t, v + np.array([1000.0, 0, 0]), n, refl_coeff=0.9)
    sc.add_mesh(t, v + np.array([-500.0, 0, 0]), n, refl_coeff=0.9)          # behind the tx: catches the reflected ray
    sc.set_receivers([_monostatic_rx(oracle)])
    r = sc.trace((0, 0, 0), (0.02, 0.02, 0.0), (0.0, 0.0), 1, 1)
    p = r["results"][0]
    assert p["received"] == -1 and p["reflDepth"] == 1
    assert r["hit_prim"][0, 0] >= 0 and r["hit_prim"][0, 1] >= 12            # second segment hit target 1, but was absorbed
    assert list(r["path"][0]) == [0]
    # with maxRefl = 2 the second plate reflects too and the path records both targets
    r2 = sc.trace((0, 0, 0), (0.02, 0.02, 0.0), (0.0, 0.0), 1, 2)
    assert list(r2["path"][0]) == [0, 1] and r2["results"][0]["reflDepth"] == 2


def _mk_rx(oracle, n, rx, refl, powers, lengths, dopplers):
    a = np.zeros(n, oracle.PRD_DTYPE)
    a["received"] = rx; a["reflDepth"] = refl; a["power"] = powers; a["rayLength"] = lengths; a["doppler"] = dopplers
    a["refrIndex"] = 1.0
    return a


def test_kat_aggregation_mean(oracle):
    """3 rays, same rx + path, powers 1, 4, 9 -> power ((1+2+3)/3)^2 = 4, arithmetic means (aggregation.cu:62-65,89-92)"""
    rx = _mk_rx(oracle, 3, 0, 1, [1.0, 4.0, 9.0], [300.0, 600.0, 900.0], [10.0, 20.0, 60.0])
    paths = np.zeros((3, 2), np.int32); paths[:, 1] = -1
    fc = 1e9
    out = oracle.aggregate_literal(rx, paths, C0, fc, 1000)
    assert np.allclose(out["results"]["power"], 4.0)
    assert np.allclose(out["results"]["doppler"], 30.0)
    d = np.array([300.0, 600.0, 900.0]) / C0
    assert np.allclose(out["delay"], d.mean())
    ph = -np.fmod(d * 2 * math.pi * fc, 2 * math.pi)
    assert np.allclose(out["phase"], ph.mean())                 # arithmetic mean of wrapped phases (quirk 10)
    assert list(out["pathMatch"]) == [0, 0, 0]
    assert list(oracle.unique_paths(out["pathMatch"])) == [0]


def test_kat_aggregation_direct_rule(oracle):
    """a direct ray matches EVERY same-rx ray (aggregation.cu:56); its pathMatch is the smallest same-rx index"""
    # order: reflected, direct, reflected -- all rx 0; plus one ray at rx 1
    rx = _mk_rx(oracle, 4, [0, 0, 0, 1], [1, 0, 1, 1], [4.0, 1.0, 16.0, 9.0], [10.0, 5.0, 10.0, 7.0], [1.0, 0.0, 3.0, 5.0])
    paths = np.array([[0], [-1], [0], [0]], np.int32)
    out = oracle.aggregate_literal(rx, paths, C0, 1e9, 100)
    assert list(out["npath"]) == [2, 3, 2, 1]                   # the direct ray counted all three rx-0 rays
    assert out["results"]["power"][1] == pytest.approx(((2 + 1 + 4) / 3.0) ** 2)
    assert out["results"]["power"][0] == pytest.approx(((2 + 4) / 2.0) ** 2)
    assert list(out["pathMatch"]) == [0, 0, 0, 3]               # the direct response collapses onto ray 0 (quirk 9)
    assert list(oracle.unique_paths(out["pathMatch"])) == [0, 3]
    # direct ray first: it keeps its own response
    rx2 = rx[[1, 0, 2, 3]]; p2 = paths[[1, 0, 2, 3]]
    out2 = oracle.aggregate_literal(rx2, p2, C0, 1e9, 100)
    assert list(out2["pathMatch"]) == [0, 1, 1, 3]


@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_kat_icosphere_size(oracle, n):
    v, t, nr = oracle.sphere_mesh(n, 2.5)
    assert t.shape[0] == 20 * 4 ** n and v.shape[0] == 10 * 4 ** n + 2          # ray_tracer.cpp:354-418
    assert np.allclose(np.linalg.norm(v, axis=1), 2.5, rtol=1e-12)
    assert np.allclose(np.linalg.norm(nr, axis=1), 1.0, rtol=1e-12)
    # std::set ordering: vertices ascending lexicographically, triangles ascending lexicographically
    key = [tuple(x) for x in (v / 2.5).round(15)]
    assert all(tuple(t[i]) < tuple(t[i + 1]) for i in range(len(t) - 1))
    assert t.max() == v.shape[0] - 1 and len(set(key)) == len(key)


def test_kat_aabb_rounding(oracle):
    """min = rd(f64 min), max = ru(f64 max) (triangle_mesh.cu:228-229); degenerate triangles invalidated (:222,232)"""
    v0, v1, v2 = np.array([0.1, 1.0, -3.3]), np.array([0.7, 1.0 + 1e-12, 2.2]), np.array([0.3, -5.5, 0.0])
    ok, b = oracle.bound(v0, v1, v2)
    assert ok
    lo = np.minimum(np.minimum(v0, v1), v2); hi = np.maximum(np.maximum(v0, v1), v2)
    assert (b[:3].astype(np.float64) <= lo).all() and (b[3:].astype(np.float64) >= hi).all()
    assert (np.nextafter(b[:3], np.float32(np.inf)).astype(np.float64) > lo).all()      # tight: one ulp up would cross
    assert (np.nextafter(b[3:], np.float32(-np.inf)).astype(np.float64) < hi).all()
    assert b[1 + 3] > np.float32(1.0)                                                   # 1 + 1e-12 rounds UP
    ok2, b2 = oracle.bound(v0, v0, v2)
    assert not ok2 and b2[0] > b2[3]


def test_vertex_rotation_uses_float_trig(oracle):
    """yaw/pitch/roll are float arguments: std::cos(float) (ray_tracer.cpp:156-161).  cosf(pi/2) != cos(pi/2)."""
    v = oracle.vertex_rotation(np.array([[1.0, 0.0, 0.0]]), np.float32(math.pi / 2), 0.0, 0.0)
    c = float(np.cos(np.float32(math.pi / 2))); s = float(np.sin(np.float32(math.pi / 2)))
    assert v[0, 0] == c and v[0, 1] == s and v[0, 0] != math.cos(math.pi / 2)


def test_rect_mesh(oracle):
    v, t, n = oracle.rect_mesh(2.0, 4.0, 6.0)
    assert v.shape == (8, 3) and t.shape == (12, 3) and n.shape == (12, 3)      # face normals in the normals slot
    assert sorted(set(np.abs(v[:, 0]))) == [1.0] and sorted(set(np.abs(v[:, 2]))) == [3.0]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0)
    assert np.array_equal(n[0], [1.0, 0.0, 0.0])                                 # tris[0] = (0,1,2) lies in x = +w/2


def test_file_mesh_roundtrip(oracle, tmp_path):
    rng = np.random.default_rng(5)
    tv = rng.normal(size=(7, 9)); tn = rng.normal(size=(7, 9))
    vf, nf = tmp_path / "v.txt", tmp_path / "n.txt"
    for f, a in ((vf, tv), (nf, tn)):
        with open(f, "w") as fh:
            for row in a:
                fh.write("%.17g %.17g %.17g, %.17g %.17g %.17g, %.17g %.17g %.17g,\n" % tuple(row))
    v, t, n = oracle.file_mesh(str(vf), str(nf))
    assert t.shape == (7, 3) and np.array_equal(t.reshape(-1), np.arange(21))    # unshared vertices (ray_tracer.cpp:445-451)
    assert np.array_equal(v.reshape(7, 9), tv) and np.array_equal(n.reshape(7, 9), tn)
    with pytest.raises(IOError):
        oracle.file_mesh(str(tmp_path / "missing"), str(nf))


def test_atan2f_within_one_ulp_of_libm(oracle):
    """[D1] the oracle's correctly rounded atan2f vs glibc atan2f (the reference calls CUDA's, <= 2 ulp)"""
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(20000):
        y, x = (rng.normal(size=2) * 10.0 ** rng.uniform(-4, 4)).astype(np.float32)
        a, b = oracle.atan2f(y, x), oracle.libm_atan2f(y, x)
        exact = math.atan2(float(y), float(x))
        ulp = abs(float(np.spacing(np.float32(exact))))
        assert abs(a - exact) <= 0.5000001 * ulp                                 # correctly rounded
        worst = max(worst, abs(a - b) / ulp)
    assert worst <= 1.0
    for y, x, want in [(0.0, 1.0, 0.0), (0.0, -1.0, math.pi), (1.0, 0.0, math.pi / 2), (-1.0, 0.0, -math.pi / 2),
                       (1.0, 1.0, math.pi / 4), (-1.0, -1.0, -3 * math.pi / 4)]:
        assert oracle.atan2f(y, x) == float(np.float32(want))


def test_oracle_bvh_equals_brute_force(oracle):
    """the oracle's own BVH (cpu_baseline leg, large sampled runs) must reproduce the brute-force definition"""
    from rts_amd import scenes
    import helpers as H
    for spec in (scenes.config2(subdiv=2, W=12, rx_radius=400.0), scenes.config3(W=12, detail=0.03, rx_radius=400.0)):
        a = H.oracle_trace(oracle, spec)
        b = H.oracle_trace(oracle, spec, use_bvh=True, threads=2)
        H.assert_prd_equal(a["results"], b["results"], spec["name"])
        assert np.array_equal(a["hit_prim"], b["hit_prim"]) and np.array_equal(a["path"], b["path"])
        assert np.array_equal(a["hit_t"].view(np.uint32), b["hit_t"].view(np.uint32))


def test_strided_sample_matches_full_launch(oracle):
    """[D3] tracing a subset of launch indices gives the same per-ray records as the full launch"""
    from rts_amd import scenes
    import helpers as H
    spec = scenes.config2(subdiv=1, W=10, rx_radius=400.0)
    full = H.oracle_trace(oracle, spec)
    sub = H.oracle_trace(oracle, spec, ray_first=3, ray_stride=7, n_rays=100)
    idx = 3 + 7 * np.arange(100)
    H.assert_prd_equal(full["results"][idx], sub["results"], "strided")
    assert np.array_equal(full["path"][idx], sub["path"])


def test_refraction_rows(oracle):
    """maxRefr > 0 is clamped to 2 and adds maxRefl + 2 child rows per launch index (ray_tracer.cpp:604-626)"""
    v, t, n = oracle.sphere_mesh(1, 5.0)
    sc = oracle.Scene()
    sc.add_mesh(t, v + np.array([100.0, 0, 0]), n, refl_coeff=0.5, refr_index=1.5)
    sc.set_receivers([_monostatic_rx(oracle, radius=30.0)])
    r = sc.trace((0, 0, 0), (0.05, 0.05, 0.0), (0.0, 0.0), 3, 2, max_refr=1)
    assert r["results"].shape[0] == 27 * (2 + 3) and r["path"].shape[1] == 4
    hit = r["results"]["reflDepth"][:27] > 0
    assert hit.any()
    child = r["results"][27:54]
    assert (child["refrDepth"][hit] >= 1).all()                  # first refraction child row = index + W^3 (normal_shader.cu:214)


def test_finalise_callbacks_closed_form(oracle):
    """ray_tracer.cpp:1204-1247 in the oracle (orc_filter_finalise_cb): the vectors and times handed to the simulator's
    callbacks, on a hand-made record set -- direct ray: transvec = origin - repos, recvvec = repos - origin (the Rx
    POSITION, not the end point); reflected ray: firstHitPoint - origin, prevHitPoint - repos; Gt at time_t, Gr at
    delay + time_t; RCS by (targ_k, rcs_angle.x, rcs_angle.y, Wl) for every depth with targ_k >= 0"""
    res = np.zeros(4, oracle.PRD_DTYPE)
    res["received"] = [-1, 1, 0, -1]
    res["reflDepth"] = [0, 0, 2, 1]
    res["rayLength"] = [0.0, 2997.92458, 5995.84916, 1.0]
    res["power"] = [0.0, 2.0, 3.0, 9.0]
    res["doppler"] = [0.0, 0.0, 40.0, 1.0]
    res["firstHitPoint"][2] = (10.0, 20.0, 30.0); res["prevHitPoint"][2] = (-5.0, 6.0, 7.0)
    path = np.array([[-1, -1], [-1, -1], [1, 0], [0, -1]], np.int32)
    ang = np.full((4, 2, 2), -1.0e6); ang[2] = [[0.1, 0.2], [0.3, 0.4]]
    origin = (1.0, 2.0, 3.0); rx_pos = np.array([[100.0, 0.0, 0.0], [0.0, 200.0, 50.0]])
    calls = []

    def get_rcs(targ, az, el, wl):
        calls.append(("rcs", targ, az, el, wl)); return 2.0 + targ

    def get_gain(is_rx, index, vec, t, wl):
        calls.append(("gain", is_rx, index, vec, t, wl)); return 5.0 if is_rx else 7.0
    c, fc, t0 = 299792458.0, 1.0e9, 0.25
    wl = c / fc
    rx, rxi, slots = oracle.filter_finalise_cb(res, path, ang, origin, rx_pos, 3, t0, wl, fc, c, get_rcs, get_gain)
    assert list(slots) == [1, 2] and np.array_equal(rxi, path[[1, 2]])
    assert calls[0] == ("gain", 0, 3, (1.0, 2.0 - 200.0, 3.0 - 50.0), t0, wl)                   # direct ray: origin - repos, time_t
    assert calls[1] == ("gain", 1, 1, (-1.0, 200.0 - 2.0, 50.0 - 3.0), 2997.92458 / c + t0, wl)  # repos - origin, delay + time_t
    assert calls[2] == ("rcs", 1, 0.1, 0.2, wl) and calls[3] == ("rcs", 0, 0.3, 0.4, wl)
    assert calls[4] == ("gain", 0, 3, (9.0, 18.0, 27.0), t0, wl)                                 # firstHitPoint - origin
    assert calls[5] == ("gain", 1, 0, (-105.0, 6.0, 7.0), 5995.84916 / c + t0, wl) and len(calls) == 6   # prevHitPoint - repos
    assert rx["power"][0] == 2.0 * (wl * wl * 7.0 * 5.0) and rx["power"][1] == 3.0 * 3.0 * 2.0 * (wl * wl * 7.0 * 5.0)
    vr = 20.0
    assert rx["doppler"][1] == fc * (((1 + vr / c) / (1 - vr / c)) - 1) and rx["doppler"][0] == 0.0


def test_reference_flow_scenarios_and_their_sensitivity(oracle, tmp_path):
    """tests/adapter_ref.py (rs::RTS's control flow over the oracle, the checker of the C++ adapter's GPU test): every
    scenario yields responses; the noise temperature accumulates once per transmitter (quirk 15); and each deliberate
    mistake a driver could make in ray_tracer.cpp:1204-1247 -- first / previous hit point swapped, Gr's rotation taken at
    time_t instead of delay + time_t, rcs_angle columns shifted -- moves some response's power by far more than the
    tolerance of the comparison (1e-9)"""
    import adapter_ref as AR
    vf, nf = str(tmp_path / "v.txt"), str(tmp_path / "n.txt")
    assert AR.write_octahedron_files(vf, nf, subdiv=2) == 128
    sc = AR.scenario_base()
    good, noise = AR.run_reference_flow(oracle, sc)
    assert len(good) > 6 and noise == [325.0, 185.0] and set(good[:, 2]) == {0.0, 1.0}
    assert (good[:, 3] > 0).all() and len(np.unique(good[:, 8])) == 3
    for mut in ("swap_hits", "rx_time", "angle_rows"):
        bad, _ = AR.run_reference_flow(oracle, sc, mutate=mut)
        assert AR.max_power_deviation(bad, good) > 1e-6, mut
    two, noise2 = AR.run_reference_flow(oracle, AR.scenario_two_tx())
    assert noise2 == [290.0 + 35.0 + 21.5, 150.0 + 35.0 + 21.5]
    assert set(two[two[:, 0] == 0][:, 7]) == {325.0, 185.0} and set(two[two[:, 0] == 1][:, 7]) <= {346.5, 206.5} and (two[:, 0] == 1).any()
    refr, _ = AR.run_reference_flow(oracle, AR.scenario_refraction())
    fil, _ = AR.run_reference_flow(oracle, AR.scenario_file(vf, nf))
    ecef, _ = AR.run_reference_flow(oracle, AR.scenario_ecef())
    assert len(refr) > 10 and set(refr[:, 2]) == {0.0, 1.0, 2.0} and len(fil) > 3 and len(ecef) > 3 and np.isfinite(refr).all()
    AR.write_scenario(str(tmp_path / "s.scn"), sc)
    lines = open(str(tmp_path / "s.scn")).read().splitlines()
    assert [l.split()[0] for l in lines] == ["params", "tx", "rx", "rx", "target", "target", "target"]


def test_cube_definition_closed_form(oracle):
    """orc_cube (the definition of the complex return cube): per ray A = sqrt(P) e^{j phi} with
    phi = -fmod(2 pi fc delay, 2 pi) in bin floor((delay - t0) / dt); per unique path one term per representative ray with the
    group's power / delay / phase; rays outside the window or with a receiver index outside the cube are dropped"""
    c, fc = 299792458.0, 1.0e9
    rx = np.zeros(4, oracle.PRD_DTYPE)
    rx["received"] = [0, 0, 1, 5]; rx["power"] = [4.0, 9.0, 16.0, 1.0]
    rx["rayLength"] = [c * 1.05e-6, c * 1.05e-6, c * 1.31e-6, c * 1.0e-6]; rx["reflDepth"] = 1
    cube = np.zeros((2, 3, 8), np.complex128)
    oracle.cube_accumulate(cube, 1, rx, 1.0e-6, 0.1e-6, c, fc)
    ph = lambda d: -math.fmod(d * 2 * math.pi * fc, 2 * math.pi)
    d0 = rx["rayLength"][0] / c; d2 = rx["rayLength"][2] / c
    want = np.zeros_like(cube)
    want[0, 1, 0] = (2.0 + 3.0) * complex(math.cos(ph(d0)), math.sin(ph(d0)))
    want[1, 1, 3] = 4.0 * complex(math.cos(ph(d2)), math.sin(ph(d2)))
    np.testing.assert_allclose(cube, want, rtol=1e-15, atol=0)
    # per unique path: rays 0 and 1 share receiver and path -> ONE term: sqrt(((2 + 3) / 2)^2) at the mean delay and phase
    paths = np.array([[0], [0], [0], [0]], np.int32)
    lit = oracle.aggregate_literal(rx[:3], paths[:3], c, fc, 100)
    cube2 = np.zeros((2, 3, 8), np.complex128)
    oracle.cube_accumulate(cube2, 2, None, 1.0e-6, 0.1e-6, c, fc, lit=lit)
    assert list(oracle.unique_paths(lit["pathMatch"])) == [0, 2]
    assert np.count_nonzero(cube2) == 2 and abs(abs(cube2[0, 2, 0]) - 2.5) < 1e-12 and abs(abs(cube2[1, 2, 3]) - 4.0) < 1e-12
    assert abs(np.angle(cube2[0, 2, 0]) - math.remainder(ph(d0), 2 * math.pi)) < 1e-9
