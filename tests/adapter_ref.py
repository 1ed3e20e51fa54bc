"""The oracle-side counterpart of include/rts_adapter.hpp: the control flow of rs::RTS (ray_tracer.cpp:806-1336 --
transmitter loop, noise-temperature side effect, pulse loop, receiver spheres, target placement, launch, host filter +
finalise with the simulator's RCS / gain callbacks, aggregation, one response per unique path) restated in Python over the
CPU oracle, for scenarios that are also written out as the text file tests/adapter/adapter_main.cpp reads.

The antenna and RCS patterns of tests/adapter/mock_soars.hpp are restated here independently (plain math, no shared code):
a driver that hands a wrong argument to any callback produces different responses on one side only.

TEST INFRASTRUCTURE: imports the oracle; nothing under rts_amd/ imports this module.
"""
import math

import numpy as np

C0 = 299792458.0


# ------------------------------------------------------------------------------- scenario description
def antenna(az=0.0, el=0.0, az_rate=0.0, el_rate=0.0, wob=0.0, wob_w=0.0, g0=1.0, gk=0.0, gw=0.0):
    return dict(az=az, el=el, az_rate=az_rate, el_rate=el_rate, wob=wob, wob_w=wob_w, g0=g0, gk=gk, gw=gw)


def write_scenario(path, sc):
    def kv(d, keys):
        return " ".join("%s=%s" % (k, (repr(float(d[k])) if isinstance(d[k], float) else d[k])) for k in keys if k in d)
    lines = ["params " + kv(sc["params"], ["W", "max_refl", "max_refr", "smooth", "c", "start", "rate"])]
    ant = ["az", "el", "az_rate", "el_rate", "wob", "wob_w", "g0", "gk", "gw"]
    for t in sc["txs"]:
        lines.append("tx " + kv(t, ["x", "y", "z", "span_az", "span_el", "span_r", "pulses", "pri", "t_first", "carrier", "temp"]) + " " + kv(t["ant"], ant))
    for r in sc["rxs"]:
        lines.append("rx " + kv(r, ["x", "y", "z", "radius", "span_theta", "span_phi", "noise"]) + " " + kv(r["ant"], ant))
    for g in sc["targets"]:
        lines.append("target " + kv(g, ["shape", "x", "y", "z", "vx", "vy", "vz", "yaw", "pitch", "roll", "yaw_rate", "pitch_rate", "roll_rate",
                                        "rotating", "w", "h", "d", "radius", "subdivs", "vfile", "nfile", "refl", "refr", "rcs", "ra", "rb", "rw"]))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


# ------------------------------------------------------------------------------- the mock's patterns, restated
def svec3(v):
    """SOARS SVec3(Vec3): length, azimuth = atan2(y, x), elevation = asin(z / length)"""
    x, y, z = v
    length = math.sqrt(x * x + y * y + z * z)
    return length, math.atan2(y, x), (math.asin(z / length) if length > 0 else 0.0)


def rotation(a, t):
    return a["az"] + a["az_rate"] * t + a["wob"] * math.sin(a["wob_w"] * t), a["el"] + a["el_rate"] * t


def gain(a, vec, rot_time, wl):
    _, az, el = svec3(vec)                                   # .length is overwritten with 1 by the driver (:1217-1218)
    raz, rel = rotation(a, rot_time)
    da, de = az - raz, el - rel
    return a["g0"] * (1 + a["gw"] * wl) / (1 + a["gk"] * (da * da + 2 * de * de)) * 1.0


def rcs(g, az, el, wl):
    return g.get("rcs", 1.0) * (1 + g.get("ra", 0.0) * math.cos(az) * math.sin(el) + g.get("rb", 0.0) * math.cos(2 * el) + g.get("rw", 0.0) * wl)


# ------------------------------------------------------------------------------- rs::RTS over the oracle
def _f32(x):
    return float(np.float32(x))


def target_mesh(O, g):
    """mesh of one target in its own frame, rotated by its t = 0 attitude (ray_tracer.cpp:955-987; float arguments)"""
    ypr = (_f32(g.get("yaw", 0.0)), _f32(g.get("pitch", 0.0)), _f32(g.get("roll", 0.0)))
    if g["shape"] == "rect":
        return O.rect_mesh(_f32(g["w"]), _f32(g["h"]), _f32(g["d"]), *ypr)
    if g["shape"] == "sphere":
        return O.sphere_mesh(int(g["subdivs"]), _f32(g["radius"]), *ypr)
    if g["shape"] == "file":
        return O.file_mesh(g["vfile"], g["nfile"], *ypr)
    raise ValueError(g["shape"])


def run_reference_flow(O, sc, mutate=None):
    """returns (responses [n][9]: tx, time, rx, power, delay, doppler, phase, noise, pulse-index; noise temperatures per rx)
    mutate: None | "swap_hits" | "rx_time" | "angle_rows" | "direct_endpoint" -- deliberate mistakes a driver could make, used
    by the tests to show that the comparison would catch them"""
    P = sc["params"]
    W, max_refl = int(P["W"]), int(P["max_refl"])
    max_refr = 2 if int(P.get("max_refr", 0)) > 0 else 0                                   # :604-605
    smooth = bool(int(P.get("smooth", 1)))
    c = float(P.get("c", C0)); start = float(P.get("start", 0.0)); sample_time = 1.0 / float(P.get("rate", 1000.0))
    D = max_refl + max_refr
    ray_total = W ** 3 * ((max_refl + 3) if max_refr else 1)                               # :608-626
    meshes = [target_mesh(O, g) for g in sc["targets"]]
    noise = [float(r.get("noise", 290.0)) for r in sc["rxs"]]
    rx_pos = np.array([[r["x"], r["y"], r["z"]] for r in sc["rxs"]], np.float64)
    out = []
    for tx_i, tx in enumerate(sc["txs"]):                                                  # :806
        carrier = float(tx.get("carrier", 10e9)); wl = c / carrier
        for j in range(len(noise)):                                                        # :829 (once per transmitter)
            noise[j] = float(tx.get("temp", 0.0)) + noise[j]
        origin = (float(tx["x"]), float(tx["y"]), float(tx["z"]))
        for k in range(int(tx["pulses"])):                                                 # :843
            t = float(tx.get("t_first", 0.0)) + k * float(tx.get("pri", 1e-3))
            taz, tel = rotation(tx["ant"], t)
            spheres = []
            for r in sc["rxs"]:                                                            # :894-918
                raz, rel = rotation(r["ant"], t)
                spheres.append(O.rx_sphere((r["x"], r["y"], r["z"]), raz, rel, r["radius"], r["span_theta"], r["span_phi"]))
            scn = O.Scene()
            for g, (v, tri, nrm) in zip(sc["targets"], meshes):                            # :936-1014, 1144-1145
                p0 = np.array([g["x"] + g.get("vx", 0.0) * t, g["y"] + g.get("vy", 0.0) * t, g["z"] + g.get("vz", 0.0) * t])
                t1 = t + sample_time
                p1 = np.array([g["x"] + g.get("vx", 0.0) * t1, g["y"] + g.get("vy", 0.0) * t1, g["z"] + g.get("vz", 0.0) * t1])
                vv, nn = v, nrm
                if int(g.get("rotating", 0)) and t > start:                                # :993-1007
                    ypr = (g.get("yaw", 0.0) + g.get("yaw_rate", 0.0) * t, g.get("pitch", 0.0) + g.get("pitch_rate", 0.0) * t, g.get("roll", 0.0) + g.get("roll_rate", 0.0) * t)
                    vv = O.vertex_rotation(v, *ypr); nn = O.vertex_rotation(nrm, *ypr)
                scn.add_mesh(tri, vv + p0, nn, g.get("refl", 0.9), g.get("refr", 1.0), (p1 - p0) / sample_time)
            scn.set_receivers(spheres)
            o = scn.trace(origin, (tx["span_az"], tx["span_el"], tx["span_r"]), (taz, tel), W, max_refl, max_refr, smooth, debug=False)
            res, path, ang = o["results"], o["path"], o["rcs_angle"]
            if mutate == "swap_hits":
                res = res.copy(); f = res["firstHitPoint"].copy(); res["firstHitPoint"] = res["prevHitPoint"]; res["prevHitPoint"] = f
            if mutate == "angle_rows":
                ang = np.roll(ang, 1, axis=1) if D > 1 else ang[:, :, ::-1]

            def get_rcs(targ, az, el, wl_):
                return rcs(sc["targets"][targ], az, el, wl_)

            def get_gain(is_rx, index, vec, rot_time, wl_):
                if is_rx and mutate == "rx_time":
                    rot_time = t
                return gain((sc["rxs"][index] if is_rx else sc["txs"][index])["ant"], vec, rot_time, wl_)
            rx, rxi, _ = O.filter_finalise_cb(res, path, ang, origin, rx_pos, tx_i, t, wl, carrier, c, get_rcs, get_gain)
            if len(rx) == 0:
                continue
            lit = O.aggregate_literal(rx, rxi, c, carrier, ray_total)                       # :1266-1285
            for u in O.unique_paths(lit["pathMatch"]):                                     # :1290-1321
                r = int(lit["results"]["received"][u]); delay = float(lit["delay"][u])
                out.append([tx_i, t + delay, r, float(lit["results"]["power"][u]), delay, float(lit["results"]["doppler"][u]), float(lit["phase"][u]), noise[r], k])
    return np.array(out, np.float64).reshape(-1, 9), noise


def parse_adapter_output(text):
    resp, noise = [], {}
    for line in text.strip().splitlines():
        f = line.split()
        if f[0] == "R":
            # R tx time-delay rx power delay doppler phase noise  ->  tx, time, rx, power, delay, doppler, phase, noise
            tx, tmd, rx, power, delay, dop, ph, nz = int(f[1]), float(f[2]), int(f[3]), float(f[4]), float(f[5]), float(f[6]), float(f[7]), float(f[8])
            resp.append([tx, tmd + delay, rx, power, delay, dop, ph, nz])
        elif f[0] == "N":
            noise[int(f[1])] = float(f[2])
    return np.array(resp, np.float64).reshape(-1, 8), [noise[j] for j in sorted(noise)]


def sort_responses(a):
    """order by (tx, pulse time rounded to the microsecond, rx, delay)"""
    return a[np.lexsort((a[:, 4], a[:, 2], np.round((a[:, 1] - a[:, 4]) * 1e6), a[:, 0]))]


def assert_responses_close(got, want, rtol=1e-9):
    got = sort_responses(got); want = sort_responses(want[:, :8])
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 2], want[:, 2])
    np.testing.assert_allclose(got[:, 1], want[:, 1], rtol=1e-12, atol=1e-15)      # time
    np.testing.assert_allclose(got[:, 3], want[:, 3], rtol=rtol)                   # power
    np.testing.assert_allclose(got[:, 4], want[:, 4], rtol=1e-12)                  # delay
    np.testing.assert_allclose(got[:, 5], want[:, 5], rtol=rtol, atol=1e-6)        # doppler [Hz]
    np.testing.assert_allclose(got[:, 6], want[:, 6], rtol=rtol, atol=1e-9)        # phase
    assert np.array_equal(got[:, 7], want[:, 7])                                   # noise temperature seen by the response


def max_power_deviation(a, b):
    a = sort_responses(a[:, :8]); b = sort_responses(b[:, :8])
    if a.shape != b.shape:
        return float("inf")
    return float(np.max(np.abs(a[:, 3] - b[:, 3]) / np.abs(b[:, 3])))


# ------------------------------------------------------------------------------- scenarios
def _rx_ant(pos, look_at=(0.0, 0.0, 0.0), **kw):
    dx, dy, dz = look_at[0] - pos[0], look_at[1] - pos[1], look_at[2] - pos[2]
    return antenna(az=math.atan2(dy, dx), el=math.atan2(dz, math.hypot(dx, dy)), **kw)


def scenario_base(offset=(0.0, 0.0, 0.0), W=16, pulses=3):
    """three moving targets (one rotating), one transmitter, two receivers; every pattern angle-, time- and wavelength-dependent"""
    ox, oy, oz = offset
    txp = (-200.0 + ox, 0.0 + oy, 0.0 + oz); r0p = txp; r1p = (-150.0 + ox, 130.0 + oy, 10.0 + oz)
    centre = (ox, oy, oz)
    return dict(
        params=dict(W=W, max_refl=4, max_refr=0, smooth=1, c=C0, start=0.0, rate=1000.0),
        txs=[dict(x=txp[0], y=txp[1], z=txp[2], span_az=0.16, span_el=0.12, span_r=0.05, pulses=pulses, pri=1e-3, t_first=0.0, carrier=10e9, temp=35.0,
                  ant=_rx_ant(txp, centre, az_rate=2.0, el_rate=-1.0, g0=3.0, gk=40.0, gw=2.0))],
        rxs=[dict(x=r0p[0], y=r0p[1], z=r0p[2], radius=90.0, span_theta=2.6, span_phi=2.6, noise=290.0,
                  ant=_rx_ant(r0p, centre, wob=0.02, wob_w=2.0e5, g0=2.0, gk=25.0, gw=1.0)),
             dict(x=r1p[0], y=r1p[1], z=r1p[2], radius=90.0, span_theta=2.6, span_phi=2.6, noise=150.0,
                  ant=_rx_ant(r1p, centre, az_rate=5.0, wob=0.015, wob_w=3.0e5, el_rate=2.0, g0=1.5, gk=60.0))],
        targets=[dict(shape="sphere", subdivs=2, radius=4.0, x=ox, y=oy, z=oz, vx=10.0, vy=0.0, vz=0.0, refl=0.9, refr=1.0, rcs=1.3, ra=0.3, rb=0.1, rw=0.5),
                 dict(shape="rect", w=5.0, h=5.0, d=5.0, yaw=0.5, pitch=0.2, roll=0.1, x=2.0 + ox, y=9.0 + oy, z=1.0 + oz, vx=0.0, vy=-5.0, vz=0.0, refl=0.8, refr=1.0,
                      rotating=1, yaw_rate=30.0, pitch_rate=0.0, roll_rate=0.0, rcs=0.7, ra=-0.4, rb=0.2),
                 dict(shape="rect", w=0.2, h=14.0, d=14.0, yaw=0.6, pitch=0.0, roll=0.0, x=9.0 + ox, y=-7.0 + oy, z=0.0 + oz, vx=0.0, vy=0.0, vz=3.0, refl=0.7, refr=1.0,
                      rcs=2.0, ra=0.2, rb=-0.3, rw=1.0)])


def scenario_two_tx():
    """BASELINE configs[3]'s shape: two transmitters (different sites, carriers, noise contributions, pulse counts) --
    SetNoiseTemperature accumulates once per transmitter (quirk 15), so the second transmitter's responses carry more noise"""
    sc = scenario_base(W=14, pulses=2)
    t2 = dict(sc["txs"][0]); t2.update(x=-180.0, y=-60.0, z=15.0, carrier=9.4e9, temp=21.5, pulses=3, pri=0.7e-3, t_first=0.2e-3, span_az=0.2, span_el=0.16)
    t2["ant"] = _rx_ant((t2["x"], t2["y"], t2["z"]), az_rate=-3.0, g0=2.5, gk=15.0, gw=4.0)
    sc["txs"].append(t2)
    return sc


def scenario_refraction():
    """maxRefr > 0 (clamped to 2, rayTotal = W^3 (maxRefl + 3), ray_tracer.cpp:604-626): partly transparent targets"""
    sc = scenario_base(W=16, pulses=2)
    sc["params"].update(max_refr=1, max_refl=3)
    sc["targets"][0].update(refl=0.6, refr=1.5)
    sc["targets"][1].update(refl=0.5, refr=1.3)
    sc["targets"][2].update(refl=1.0)                        # |reflCoeff| == 1: never refracts (normal_shader.cu:198)
    # a third receiver BEHIND the targets: it captures the direct rays and the rays refracted into and out of the sphere
    # (rows rayIndex + 2 W^3, normal_shader.cu:214-215)
    p = (220.0, 15.0, 5.0)
    sc["rxs"].append(dict(x=p[0], y=p[1], z=p[2], radius=120.0, span_theta=2.6, span_phi=2.6, noise=90.0, ant=_rx_ant(p, (0.0, 0.0, 0.0), g0=1.2, gk=10.0, az_rate=1.0)))
    return sc


def scenario_file(vfile, nfile):
    """a "file" target (ray_tracer.cpp:983-987) beside a rect; the test writes the two text files"""
    sc = scenario_base(W=14, pulses=2)
    sc["targets"][0] = dict(shape="file", vfile=vfile, nfile=nfile, yaw=0.3, pitch=-0.2, roll=0.1, x=0.0, y=0.0, z=0.0, vx=10.0, vy=0.0, vz=0.0, refl=0.9, refr=1.0,
                            rotating=1, yaw_rate=12.0, pitch_rate=3.0, roll_rate=-4.0, rcs=1.1, ra=0.25, rb=0.15)
    return sc


def scenario_ecef():
    """the base scene 10 km above the reference's Earth sphere, at an oblique point: |x| ~ 6.4e6 m"""
    re = 6378136.0 + 10000.0
    lat, lon = 0.6, -1.1
    off = (re * math.cos(lat) * math.cos(lon), re * math.cos(lat) * math.sin(lon), re * math.sin(lat))
    return scenario_base(offset=off, W=14, pulses=2)


def write_octahedron_files(vpath, npath, radius=4.0, subdiv=2):
    """a small closed mesh in the reference's file format ("x y z, x y z, x y z," per line; a second file with the vertex
    normals): an octahedron subdivided `subdiv` times and pushed out to a sphere"""
    v = [np.array(p, float) for p in [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]]
    faces = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    tris = [(v[a], v[b], v[c]) for a, b, c in faces]
    for _ in range(subdiv):
        nt = []
        for a, b, c in tris:
            ab, bc, ca = [(p + q) / np.linalg.norm(p + q) for p, q in ((a, b), (b, c), (c, a))]
            nt += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        tris = nt
    with open(vpath, "w") as fv, open(npath, "w") as fn:
        for tri in tris:
            fv.write(" ".join("%.17g %.17g %.17g," % tuple(radius * p) for p in tri) + "\n")
            fn.write(" ".join("%.17g %.17g %.17g," % tuple(p) for p in tri) + "\n")
    return len(tris)
