"""Synthetic scenes for the BASELINE.json configurations.

The reference ships no scenes; these are this repository's own seeded generators.  They only
build INPUT arrays (meshes in the target frame, per-pulse placements, receiver spheres, beam
parameters) through the library's host helpers (rts_rect_mesh / rts_sphere_mesh / rts_rx_sphere)
and numpy; tests feed the identical arrays to the product and to the oracle.

Conventions: Tx at (-range, 0, 0) looking along +x at targets near the origin (the Earth test of
ray_tracer.cu:438-476 is inert for received rays), isotropic antennas, RCS = 1,
c = 299 792 458 m/s, fc = 10 GHz, cw_sample_rate = 1 kHz.
"""
import math

import numpy as np

from . import api

C0 = 299792458.0
FC = 10.0e9
SAMPLE_TIME = 1.0e-3


def _ellipsoid(nu, nv, semi, centre=(0, 0, 0), yaw=0.0):
    """Closed ellipsoid with 2*nu*(nv-1) triangles (pole fans + quad bands), shared vertices,
    analytic unit normals; long axis first in `semi`."""
    a, b, c = semi
    lat = np.linspace(-math.pi / 2, math.pi / 2, nv + 1)[1:-1]          # nv-1 interior rings
    lon = np.linspace(0.0, 2 * math.pi, nu, endpoint=False)
    cl, sl = np.cos(lat)[:, None], np.sin(lat)[:, None]
    ring = np.stack([np.broadcast_to(sl, (nv - 1, nu)), cl * np.cos(lon)[None, :], cl * np.sin(lon)[None, :]], -1)
    unit = np.concatenate([np.array([[-1.0, 0, 0]]), ring.reshape(-1, 3), np.array([[1.0, 0, 0]])], 0)   # poles on the long axis
    verts = unit * np.array([a, b, c])
    nrm = unit / np.array([a, b, c])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tris = []
    south, north = 0, 1 + (nv - 1) * nu

    def vid(r, k):
        return 1 + r * nu + (k % nu)
    for k in range(nu):
        tris.append((south, vid(0, k + 1), vid(0, k)))
        tris.append((north, vid(nv - 2, k), vid(nv - 2, k + 1)))
    for r in range(nv - 2):
        for k in range(nu):
            tris.append((vid(r, k), vid(r, k + 1), vid(r + 1, k + 1)))
            tris.append((vid(r, k), vid(r + 1, k + 1), vid(r + 1, k)))
    tris = np.array(tris, np.uint32)
    if yaw:
        cy, sy = math.cos(yaw), math.sin(yaw)
        R = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
        verts = verts @ R.T; nrm = nrm @ R.T
    verts = verts + np.array(centre, np.float64)
    return verts, tris, nrm


def _merge(parts):
    vs, ts, ns, off = [], [], [], 0
    for v, t, n in parts:
        vs.append(v); ns.append(n); ts.append(t + off); off += v.shape[0]
    return np.concatenate(vs), np.concatenate(ts).astype(np.uint32), np.concatenate(ns)


def aircraft_mesh(scale=1.0, detail=1.0):
    """'Aircraft-like' union of four ellipsoids: fuselage, wing, tailplane, fin.
    detail = 1 gives exactly 100 000 triangles."""
    d = math.sqrt(detail)

    def n(x):
        return max(8, int(round(x * d)))
    parts = [
        _ellipsoid(n(250), n(120) + 1, (15.0 * scale, 1.8 * scale, 1.8 * scale)),
        _ellipsoid(n(200), n(50) + 1, (2.5 * scale, 16.0 * scale, 0.35 * scale), centre=(1.0 * scale, 0, -0.3 * scale)),
        _ellipsoid(n(100), n(50) + 1, (1.2 * scale, 5.5 * scale, 0.2 * scale), centre=(-13.0 * scale, 0, 0.4 * scale)),
        _ellipsoid(n(100), n(50) + 1, (1.8 * scale, 0.2 * scale, 3.2 * scale), centre=(-13.0 * scale, 0, 2.5 * scale)),
    ]
    return _merge(parts)


def _ico_ellipsoid(subdivs, semi, centre=(0, 0, 0)):
    """ellipsoid from the library's icosphere (20 * 4^subdivs triangles, no poles: every vertex has valence 5 or 6), analytic
    unit normals"""
    v, t, _ = api.sphere_mesh(subdivs, 1.0)
    abc = np.array(semi, np.float64)
    unit = v / np.linalg.norm(v, axis=1, keepdims=True)
    nrm = unit / abc
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return unit * abc + np.array(centre, np.float64), t.astype(np.uint32), nrm


def aircraft_mesh_ico(scale=1.0):
    """the same four ellipsoids as aircraft_mesh, tessellated from icospheres: 81 920 + 3 x 5 120 = 97 280 triangles, and no
    pole fans (the lat-long ellipsoids of aircraft_mesh end in fans of 100-250 slivers around one vertex, whose boxes all
    overlap there: the rays through a pole are the slowest tiles of a C3 launch)"""
    return _merge([
        _ico_ellipsoid(6, (15.0 * scale, 1.8 * scale, 1.8 * scale)),
        _ico_ellipsoid(4, (2.5 * scale, 16.0 * scale, 0.35 * scale), centre=(1.0 * scale, 0, -0.3 * scale)),
        _ico_ellipsoid(4, (1.2 * scale, 5.5 * scale, 0.2 * scale), centre=(-13.0 * scale, 0, 0.4 * scale)),
        _ico_ellipsoid(4, (1.8 * scale, 0.2 * scale, 3.2 * scale), centre=(-13.0 * scale, 0, 2.5 * scale)),
    ])


def plate_mesh(size):
    """Two-triangle square plate in the local y-z plane facing -x (file_mesh-style unshared vertices)."""
    h = size / 2.0
    v = np.array([[0, -h, -h], [0, h, -h], [0, h, h], [0, -h, -h], [0, h, h], [0, -h, h]], np.float64)
    t = np.array([[0, 1, 2], [3, 4, 5]], np.uint32)
    n = np.tile(np.array([[-1.0, 0, 0]]), (6, 1))
    return v, t, n


def _rx_at(pos, look_at, radius, span):
    d = np.asarray(look_at, np.float64) - np.asarray(pos, np.float64)
    az = math.atan2(d[1], d[0]); el = math.atan2(d[2], math.hypot(d[0], d[1]))
    return api.rx_sphere(pos, az, el, radius, span, span)


def _static_motion(n_targets, positions=None, velocities=None):
    out = []
    for i in range(n_targets):
        out.append(dict(position=tuple(positions[i]) if positions is not None else (0.0, 0.0, 0.0),
                        velocity=tuple(velocities[i]) if velocities is not None else (0.0, 0.0, 0.0)))
    return out


def config1():
    """C1: flat plate (2 triangles) at 1 km, W = 22, 1 bounce, monostatic rx sphere r = 10 m."""
    v, t, n = plate_mesh(10.0)
    return dict(name="C1-plate-2tri", W=22, max_refl=1, smooth=True, n_pulses=1,
                meshes=[dict(tris=t, verts=v, normals=n, refl_coeff=0.9, refr_index=1.0)],
                motion=_static_motion(1, [(0.0, 0.0, 0.0)], [(0.0, 0.0, 0.0)]),
                tx=dict(origin=(-1000.0, 0.0, 0.0), span=(0.02, 0.02, 0.0), dir=(0.0, 0.0)),
                rx=[_rx_at((-1000.0, 0.0, 0.0), (0, 0, 0), 10.0, math.pi / 2)], carrier=FC, c=C0)


def config2(subdiv=5, W=100, rx_radius=50.0):
    """C2: icosphere (n = 5: 20 480 triangles) radius 5 m at 1 km, W = 100, 4 bounces, 64 pulses, static."""
    v, t, n = api.sphere_mesh(subdiv, 5.0)
    return dict(name="C2-icosphere-%dtri" % t.shape[0], W=W, max_refl=4, smooth=True, n_pulses=64,
                meshes=[dict(tris=t, verts=v, normals=n, refl_coeff=0.9, refr_index=1.0)],
                motion=_static_motion(1, [(0.0, 0.0, 0.0)], [(30.0, 0.0, 0.0)]),
                tx=dict(origin=(-1000.0, 0.0, 0.0), span=(0.011, 0.011, 0.1), dir=(0.0, 0.0)),
                rx=[_rx_at((-1000.0, 0.0, 0.0), (0, 0, 0), rx_radius, math.pi / 2)], carrier=FC, c=C0)


def config3(W=216, detail=1.0, rx_radius=50.0, n_rx=4, ico=False):
    """C3: aircraft-like mesh, 100 000 triangles, 1 Tx / 4 Rx on a 2 km arc, W = 216, 6 bounces, 256 pulses.
    ico: the pole-free tessellation of the same shape (aircraft_mesh_ico, 97 280 triangles)."""
    v, t, n = aircraft_mesh_ico() if ico else aircraft_mesh(detail=detail)
    # broadside-ish aspect: ~15 % of the beam's launch indices hit the airframe (stated with every result)
    R = api.rotation_matrix(math.radians(70.0), math.radians(5.0), math.radians(35.0)).reshape(3, 3)
    v = v @ R.T; n = n @ R.T
    rx = []
    for k in range(n_rx):
        ang = math.radians(-30.0 + 60.0 * k / max(n_rx - 1, 1)) if n_rx > 1 else 0.0
        pos = (-1000.0 * math.cos(ang), 1000.0 * math.sin(ang), 0.0)
        rx.append(_rx_at(pos, (0, 0, 0), rx_radius, math.pi / 2))
    return dict(name="C3-aircraft-%dtri" % t.shape[0], W=W, max_refl=6, smooth=True, n_pulses=256,
                meshes=[dict(tris=t, verts=v, normals=n, refl_coeff=0.9, refr_index=1.0)],
                motion=_static_motion(1, [(0.0, 0.0, 0.0)], [(200.0, 20.0, 0.0)]),
                tx=dict(origin=(-1000.0, 0.0, 0.0), span=(0.032, 0.032, 0.1), dir=(0.0, 0.0)),
                rx=rx, carrier=FC, c=C0)


def config_multi(W=16, rx_radius=90.0, max_refl=4, smooth=True):
    """Small multi-target scene for the parity tests: icosphere + rotated "rect" box (per-face normal
    rule, triangle_mesh.cu:178-180) + file-style plate, two receivers, inter-target bounces."""
    sv, st, sn = api.sphere_mesh(2, 4.0)
    bv, bt, bn = api.rect_mesh(5.0, 5.0, 5.0, 0.5, 0.2, 0.1)
    pv, pt, pn = plate_mesh(14.0)
    Rp = api.rotation_matrix(math.radians(35.0), 0.0, 0.0).reshape(3, 3)
    pv = pv @ Rp.T; pn = pn @ Rp.T
    meshes = [dict(tris=st, verts=sv, normals=sn, refl_coeff=0.9, refr_index=1.0),
              dict(tris=bt, verts=bv, normals=bn, refl_coeff=0.8, refr_index=1.0),
              dict(tris=pt, verts=pv, normals=pn, refl_coeff=0.7, refr_index=1.0)]
    motion = _static_motion(3, [(0.0, 0.0, 0.0), (2.0, 9.0, 1.0), (9.0, -7.0, 0.0)], [(10.0, 0.0, 0.0), (0.0, -5.0, 0.0), (0.0, 0.0, 3.0)])
    rx = [_rx_at((-200.0, 0.0, 0.0), (0, 0, 0), rx_radius, 2.6), _rx_at((-150.0, 130.0, 10.0), (0, 0, 0), rx_radius, 2.6)]
    return dict(name="multi-3targets", W=W, max_refl=max_refl, smooth=smooth, n_pulses=1, meshes=meshes, motion=motion,
                tx=dict(origin=(-200.0, 0.0, 0.0), span=(0.16, 0.12, 0.05), dir=(0.0, 0.0)), rx=rx, carrier=FC, c=C0)


def config4(W=465, detail=2.5, rx_radius=80.0, n_rx=8, n_tx=2):
    """C4: four aircraft-like meshes of 250 000 triangles each (1 M triangles), 50 m apart, bistatic 2 Tx / 8 Rx,
    W = 465 (100 544 625 launch indices per pulse), 8 bounces.  Returns the spec for transmitter `tx_index`
    via spec["tx_list"]; spec["tx"] is transmitter 0."""
    v, t, n = aircraft_mesh(detail=detail)
    meshes, motion = [], []
    offs = [(-25.0, -25.0, 0.0), (25.0, -25.0, 5.0), (-25.0, 25.0, -5.0), (25.0, 25.0, 0.0)]
    for i, o in enumerate(offs):
        R = api.rotation_matrix(math.radians(70.0 - 25.0 * i), math.radians(5.0 * i), math.radians(35.0 - 10.0 * i)).reshape(3, 3)
        meshes.append(dict(tris=t, verts=v @ R.T, normals=n @ R.T, refl_coeff=0.9 - 0.05 * i, refr_index=1.0))
        motion.append(dict(position=o, velocity=(150.0 + 20.0 * i, 10.0 * i, 0.0)))
    rx = []
    for k in range(n_rx):
        ang = math.radians(-70.0 + 140.0 * k / max(n_rx - 1, 1))
        rx.append(_rx_at((-1500.0 * math.cos(ang), 1500.0 * math.sin(ang), 50.0 * (k % 3)), (0, 0, 0), rx_radius, math.pi / 2))
    txs = [dict(origin=(-1500.0, -200.0, 0.0), span=(0.07, 0.07, 0.1), dir=(math.atan2(200.0, 1500.0), 0.0)),
           dict(origin=(-1400.0, 600.0, 100.0), span=(0.07, 0.07, 0.1), dir=(math.atan2(-600.0, 1400.0), math.atan2(-100.0, math.hypot(1400.0, 600.0))))][:n_tx]
    return dict(name="C4-4aircraft-%dtri" % (4 * t.shape[0]), W=W, max_refl=8, smooth=True, n_pulses=512, meshes=meshes, motion=motion,
                tx=txs[0], tx_list=txs, rx=rx, carrier=FC, c=C0)


def config5(W=216, detail=1.0, rx_radius=50.0):
    """C5: the C3 airframe with a per-pulse rigid transform (v = 200 m/s along +y, 1 rad/s yaw), 1024-pulse CPI"""
    s = config3(W=W, detail=detail, rx_radius=rx_radius, n_rx=1)
    s["name"] = s["name"].replace("C3", "C5-moving"); s["n_pulses"] = 1024
    s["motion_fn"] = config5_motion
    s["tx_track"] = True          # 1024 pulses at 200 m/s are 205 m of flight across a 32 m beam: the transmitter's boresight follows the target
                                  # (Transmitter::GetRotation is read per pulse, ray_tracer.cpp:888); callers that honour it aim tx dir at motion[0].position
    return s


def config5_motion(pulse, speed=200.0, yaw_rate=1.0, prf=1000.0):
    """C5: per-pulse rigid transform of the C3 mesh (v = 200 m/s along +y, 1 rad/s yaw)."""
    tt = pulse / prf
    R = api.rotation_matrix(yaw_rate * tt, 0.0, 0.0)
    return [dict(position=(0.0, speed * tt, 0.0), velocity=(0.0, speed, 0.0), rotation=R)]


def config_sphere6(W=216, subdiv=6, max_refl=6):
    """The scene of the C++ boundary benchmark (tests/adapter/adapter_bench.cpp over the mock SOARS world), so that the ctypes
    bench and the C++ adapter are timed on ONE scene: an icosphere of 20 * 4^subdiv triangles (6: 81 920), r = 15 m, 2 km from
    the transmitter, moving at (200, 20, 0) m/s and yawing at 1 rad/s, four receivers on a 2 km arc, W = 216, 6 bounces."""
    v, t, n = api.sphere_mesh(subdiv, 15.0)
    rx = []
    for k in range(4):
        a = -0.6 + 0.4 * k
        pos = (-2000.0 * math.cos(a), 2000.0 * math.sin(a), 20.0 * k)
        az = math.atan2(-pos[1], -pos[0]); el = math.atan2(-pos[2], math.hypot(pos[0], pos[1]))
        rx.append(api.rx_sphere(pos, az, el, 50.0, 2.6, 2.6))
    s = dict(name="sphere6-icosphere-%dtri" % t.shape[0], W=W, max_refl=max_refl, smooth=True, n_pulses=256,
             meshes=[dict(tris=t, verts=v, normals=n, refl_coeff=0.9, refr_index=1.0)],
             motion=_static_motion(1, [(0.0, 0.0, 0.0)], [(200.0, 20.0, 0.0)]),
             tx=dict(origin=(-2000.0, 0.0, 0.0), span=(0.04, 0.04, 0.05), dir=(0.0, 0.0)), rx=rx, carrier=FC, c=C0)
    s["motion_fn"] = sphere6_motion
    return s


def sphere6_motion(pulse, prf=1000.0):
    """placement of the adapter benchmark's target at pulse `pulse`: p0 + v t, yaw = 1 rad/s x t (applied for t > 0, as
    ray_tracer.cpp:993 does)"""
    tt = pulse / prf
    m = dict(position=(200.0 * tt, 20.0 * tt, 0.0), velocity=(200.0, 20.0, 0.0))
    if tt > 0:
        m["rotation"] = api.rotation_matrix(1.0 * tt, 0.0, 0.0)
    return [m]


def world_vertices(mesh, motion):
    """World-space vertices/normals exactly as the device placement kernel computes them
    (R * v accumulated from zero in k order, then + position; ray_tracer.cpp:120-137, 1010-1014)."""
    v = np.asarray(mesh["verts"], np.float64); n = np.asarray(mesh["normals"], np.float64)
    rot = motion.get("rotation")
    if rot is not None:
        R = np.asarray(rot, np.float64).reshape(3, 3)

        def rotate(a):
            out = np.zeros_like(a)
            for i in range(3):
                s = np.zeros(a.shape[0])
                for k in range(3):
                    s = s + R[i, k] * a[:, k]
                out[:, i] = s
            return out
        v = rotate(v); n = rotate(n)
    p = np.asarray(motion["position"], np.float64)
    return v + p[None, :], n


# ------------------------------------------------------------------------------- receiver-capture branch scenes
def rx_window(centre, radius, theta, phi):
    """a capture sphere given directly by its buffers (ray_tracer.cu:33-38): centre, radius, (minTheta, maxTheta), (minPhi, maxPhi)"""
    return dict(centre=np.array(centre, np.float64), radius=float(radius), minTheta=float(theta[0]), maxTheta=float(theta[1]),
                minPhi=float(phi[0]), maxPhi=float(phi[1]))


def config_miss_branches(W=20, max_refl=4):
    """The three-target scene with six capture spheres chosen so that one launch drives the miss program
    (ray_tracer.cu:260-478) through the branches the BASELINE scenes never reach:
      rx0 / rx1  windows that cross phi = +pi/2 / -pi/2 (second (theta, phi) region, :354-368), above / below the targets;
      rx2 / rx3  two overlapping spheres with wide windows in the path of the direct AND the reflected rays: no `break`
                 in the receiver loop (:272), so a ray is captured twice, power multiplied twice, last receiver wins (quirk 4);
      rx4        a sphere beside the targets crossed along chords: both roots inside the window, nearest wins (:378-381);
      rx5        a sphere in the beam whose window admits the EXIT point only, and only through the second region."""
    s = config_multi(W=W, max_refl=max_refl)
    s["name"] = "miss-branches"
    s["tx"] = dict(origin=(-200.0, 0.0, 0.0), span=(0.12, 0.10, 0.05), dir=(0.0, 0.0))
    s["rx"] = [
        rx_window((-30.0, 0.0, 40.0), 30.0, (-1.5, 1.5), (0.5, 2.6)),
        rx_window((-30.0, 0.0, -40.0), 30.0, (-1.5, 1.5), (-2.6, -0.5)),
        rx_window((-70.0, 0.0, 0.0), 30.0, (-1.5, 1.5), (-1.5, 1.5)),
        rx_window((-90.0, 5.0, 5.0), 40.0, (-1.5, 1.5), (-1.5, 1.5)),
        rx_window((0.0, 70.0, 0.0), 45.0, (-math.pi / 2 - 1.5, -math.pi / 2 + 1.5), (-1.5, 1.5)),
        rx_window((-150.0, -20.0, -20.0), 30.0, (math.pi + 0.3, math.pi + 1.3), (0.5, 2.6)),
    ]
    return s


def config_pole(up=True):
    """Direct rays straight up (down) through a capture sphere exactly above (below) the transmitter: the end points sit
    on the sphere's poles, where atan2f returns the f32 nearest to pi/2 -- which is LARGER than the f64 pi/2 -- so the
    phi correction of ray_tracer.cu:332-340 fires (it is dead code anywhere else: phi comes from atan2f(z, +sqrt))."""
    sgn = 1.0 if up else -1.0
    return dict(name="pole-%s" % ("up" if up else "down"), W=3, max_refl=1, smooth=True, n_pulses=1,
                meshes=[dict(zip(("verts", "tris", "normals"), plate_mesh(2.0)), refl_coeff=0.9, refr_index=1.0)],
                motion=_static_motion(1, [(500.0, 0.0, 0.0)], [(0.0, 0.0, 0.0)]),
                tx=dict(origin=(10.0, 20.0, 30.0), span=(1.0e-9, 1.0e-9, 0.0), dir=(0.3, sgn * math.pi / 2)),
                rx=[rx_window((10.0, 20.0, 30.0 + sgn * 100.0), 20.0, (0.3 - 1.5, 0.3 + 1.5), (sgn * math.pi / 2 - 0.6, sgn * math.pi / 2 + 0.6)),
                    rx_window((10.0, 20.0, 30.0 + sgn * 200.0), 20.0, (0.3 + math.pi - 1.5, 0.3 + math.pi + 1.5), (-sgn * math.pi / 2 - 0.6, -sgn * math.pi / 2 + 0.6))],
                carrier=FC, c=C0)


# ------------------------------------------------------------------------------- Earth-centred placement
EARTH_RADIUS = 6378136.0          # ray_tracer.cu:447


def translate(spec, offset):
    """the same scene with transmitter, receivers and targets moved by `offset` (np.float64[3]).  The Earth sphere of
    ray_tracer.cu:447-476 is centred on the world origin, so a real SOARS scene lives at |x| >= 6.378e6 m: every scene
    above sits INSIDE that sphere (each non-received ray takes the t1 >= 0 root only); translated to Earth-centred
    coordinates rays pointing down take both roots, rays pointing up none."""
    off = np.asarray(offset, np.float64)
    s = dict(spec)
    s["name"] = spec["name"] + "@ecef"
    s["tx"] = dict(spec["tx"], origin=tuple(np.asarray(spec["tx"]["origin"], np.float64) + off))
    if "tx_list" in spec:
        s["tx_list"] = [dict(t, origin=tuple(np.asarray(t["origin"], np.float64) + off)) for t in spec["tx_list"]]
    s["rx"] = [dict(r, centre=np.asarray(r["centre"], np.float64) + off) for r in spec["rx"]]
    s["motion"] = [dict(m, position=tuple(np.asarray(m["position"], np.float64) + off)) for m in spec["motion"]]
    return s


def ecef_offset(height=10.0e3, lat=0.0, lon=0.0):
    """a point `height` metres above the reference's Earth sphere; lat = lon = 0 puts it on the +x axis; the default of the
    bench configuration is the north pole (0, 0, R + h), with the C3 beam (along +x) then grazing horizontally"""
    r = EARTH_RADIUS + height
    return np.array([r * math.cos(lat) * math.cos(lon), r * math.cos(lat) * math.sin(lon), r * math.sin(lat)], np.float64)


def config2_file(W=100, rx_radius=50.0, n_tris=10000):
    """C2 as BASELINE.json words it ("sphere mesh 10k tris"): an icosphere has 20 * 4^n triangles (5 120 or 20 480), so the
    10 000-triangle sphere is a lat-long tessellation written in the reference's mesh file format ("x y z, x y z, x y z," per
    triangle, one file of vertices and one of per-vertex normals, ray_tracer.cpp:429-504) and loaded through rts_file_mesh:
    unshared vertices, 3 * n_tris of them."""
    import os
    import tempfile
    nv = 51; nu = n_tris // (2 * (nv - 1))
    v, t, n = _ellipsoid(nu, nv, (5.0, 5.0, 5.0))
    assert t.shape[0] == n_tris, (t.shape, n_tris)
    d = tempfile.mkdtemp(prefix="rts_c2file_")
    vf, nf = os.path.join(d, "sphere_v.txt"), os.path.join(d, "sphere_n.txt")
    for path, arr in ((vf, v[t].reshape(-1, 9)), (nf, n[t].reshape(-1, 9))):
        with open(path, "w") as fh:
            for row in arr:
                fh.write("%.17g %.17g %.17g, %.17g %.17g %.17g, %.17g %.17g %.17g,\n" % tuple(row))
    fv, ft, fn = api.file_mesh(vf, nf, 0.0, 0.0, 0.0)
    s = config2(subdiv=1, W=W, rx_radius=rx_radius)
    s["name"] = "C2-filemesh-%dtri" % ft.shape[0]
    s["meshes"] = [dict(tris=ft, verts=fv, normals=fn, refl_coeff=0.9, refr_index=1.0)]
    return s
