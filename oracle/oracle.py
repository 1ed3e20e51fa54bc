"""ctypes front-end of the CPU oracle (oracle/rts_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under rts_amd/ imports this module.  Parity status of the oracle
itself: *parity unpinned* (see the header of rts_oracle.cpp).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTS_ORACLE_NATIVE=1 (set by bench.py's cpu_baseline leg before the first use): the -O3 -march=native build
_NATIVE = os.environ.get("RTS_ORACLE_NATIVE", "0") == "1"
# RTS_ORACLE_LIB: another build of the same source, made by the caller (the sanitizer build, oracle/Makefile: asan)
_LIB = os.environ.get("RTS_ORACLE_LIB") or os.path.join(_HERE, "librts_oracle_native.so" if _NATIVE else "librts_oracle.so")

# PerRayData, /root/reference/ray_tracer.h:13-28 (144 B, 16-aligned; offsets verified with hipcc)
PRD_DTYPE = np.dtype({
    "names": ["rayLength", "refrIndex", "reflDepth", "refrDepth", "maxRayIndex", "rayDirection",
              "firstHitPoint", "prevHitPoint", "power", "doppler", "received", "end"],
    "formats": ["<f8", ("<f8", 2), "<u4", "<u4", "<u4", ("<f8", 3), ("<f8", 3), ("<f8", 3), "<f8", "<f8", "<i4", "u1"],
    "offsets": [0, 16, 32, 36, 40, 48, 72, 96, 120, 128, 136, 140],
    "itemsize": 144,
})


def build(force=False):
    src = os.path.join(_HERE, "rts_oracle.cpp")
    if os.environ.get("RTS_ORACLE_LIB"):
        return _LIB
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        # a child of a profiled process must not inherit the profiler's preload (it would initialise the GPU in make / g++)
        env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB") and not k.startswith("ROCP")}
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["native"] if _NATIVE else []), env=env)
    return _LIB


class OPulse(C.Structure):
    _fields_ = [("rayOrigin", C.c_double * 3), ("txSpan", C.c_double * 3), ("txDir", C.c_double * 2),
                ("width", C.c_uint32), ("maxRefl", C.c_uint32), ("maxRefr", C.c_uint32),
                ("interpolate_smooth", C.c_uint32)]


RCS_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double)
GAIN_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_clear_meshes.argtypes = [C.c_void_p]
        L.orc_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                         C.c_uint32, C.c_double, C.c_double, C.c_void_p]
        L.orc_set_receivers.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
        L.orc_rows_per_ray.restype = C.c_uint32
        L.orc_rows_per_ray.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_trace.restype = C.c_int
        L.orc_trace.argtypes = [C.c_void_p, C.POINTER(OPulse), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bound.restype = C.c_int
        L.orc_bound.argtypes = [C.c_void_p] * 4
        L.orc_atan2f.restype = C.c_float
        L.orc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_libm_atan2f.restype = C.c_float
        L.orc_libm_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_rx_sphere.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_vertex_rotation.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float]
        L.orc_rect_mesh.argtypes = [C.c_float] * 6 + [C.c_void_p] * 3
        L.orc_sphere_mesh.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 5
        L.orc_file_mesh.restype = C.c_int
        L.orc_file_mesh.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_uint32]
        L.orc_filter_finalise.restype = C.c_uint64
        L.orc_filter_finalise.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
        L.orc_filter_finalise_cb.restype = C.c_uint64
        L.orc_filter_finalise_cb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int,
                                             C.c_double, C.c_double, C.c_double, C.c_double, RCS_FN, GAIN_FN, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_aggregate_literal.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_double, C.c_double] + \
                                           [C.c_void_p] * 6
        L.orc_cube.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_uint,
                               C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_unique_paths.restype = C.c_uint
        L.orc_unique_paths.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
        L.orc_coverage.restype = C.c_uint
        L.orc_coverage.argtypes = [C.c_void_p, C.c_int]
        L.orc_sizeof_prd.restype = C.c_uint32
        assert L.orc_sizeof_prd() == PRD_DTYPE.itemsize
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ------------------------------------------------------------------------------- meshes
def rect_mesh(w, h, d, yaw=0.0, pitch=0.0, roll=0.0):
    v = np.zeros((8, 3)); t = np.zeros((12, 3), np.uint32); n = np.zeros((12, 3))
    lib().orc_rect_mesh(w, h, d, yaw, pitch, roll, _p(v), _p(t), _p(n))
    return v, t, n


def sphere_mesh(subdivs, radius, yaw=0.0, pitch=0.0, roll=0.0):
    nv = C.c_uint32(); nt = C.c_uint32()
    lib().orc_sphere_mesh(subdivs, radius, yaw, pitch, roll, None, C.byref(nv), None, C.byref(nt), None)
    v = np.zeros((nv.value, 3)); t = np.zeros((nt.value, 3), np.uint32); n = np.zeros((nv.value, 3))
    lib().orc_sphere_mesh(subdivs, radius, yaw, pitch, roll, _p(v), C.byref(nv), _p(t), C.byref(nt), _p(n))
    return v, t, n


def file_mesh(v_file, n_file, yaw=0.0, pitch=0.0, roll=0.0):
    nt = lib().orc_file_mesh(v_file.encode(), n_file.encode(), yaw, pitch, roll, None, None, None, 0)
    if nt < 0:
        raise IOError("oracle file_mesh: cannot open %s" % v_file)
    v = np.zeros((3 * nt, 3)); t = np.zeros((nt, 3), np.uint32); n = np.zeros((3 * nt, 3))
    rc = lib().orc_file_mesh(v_file.encode(), n_file.encode(), yaw, pitch, roll, _p(v), _p(t), _p(n), nt)
    if rc < 0:
        raise IOError("oracle file_mesh failed rc=%d" % rc)
    return v, t, n


def vertex_rotation(verts, yaw, pitch, roll):
    v = np.ascontiguousarray(verts, np.float64).copy()
    lib().orc_vertex_rotation(_p(v), v.shape[0], yaw, pitch, roll)
    return v


def rx_sphere(pos, az, el, radius, theta_span, phi_span):
    """returns dict(centre, radius, minTheta, maxTheta, minPhi, maxPhi)  (ray_tracer.cpp:894-918)"""
    out = np.zeros(9); pos = np.ascontiguousarray(pos, np.float64)
    lib().orc_rx_sphere(_p(pos), az, el, radius, theta_span, phi_span, _p(out))
    return dict(centre=out[0:3].copy(), radius=out[3], minTheta=out[4], maxTheta=out[5], minPhi=out[6], maxPhi=out[7])


def bound(v0, v1, v2):
    out = np.zeros(6, np.float32)
    ok = lib().orc_bound(_p(np.ascontiguousarray(v0, np.float64)), _p(np.ascontiguousarray(v1, np.float64)),
                         _p(np.ascontiguousarray(v2, np.float64)), _p(out))
    return bool(ok), out


# ------------------------------------------------------------------------------- scene + trace
class Scene:
    def __init__(self):
        self.h = C.c_void_p(lib().orc_scene_create())
        self.n_targets = 0
        self.tri_counts = []

    def __del__(self):
        try:
            lib().orc_scene_destroy(self.h)
        except Exception:
            pass

    def clear_meshes(self):
        lib().orc_scene_clear_meshes(self.h); self.n_targets = 0; self.tri_counts = []

    def add_mesh(self, tris, verts_world, normals, refl_coeff=1.0, refr_index=1.0, vel=(0, 0, 0)):
        t = np.ascontiguousarray(tris, np.uint32); v = np.ascontiguousarray(verts_world, np.float64)
        n = np.ascontiguousarray(normals, np.float64); ve = np.ascontiguousarray(vel, np.float64)
        lib().orc_scene_add_mesh(self.h, _p(t), t.shape[0], _p(v), v.shape[0], _p(n), n.shape[0], refl_coeff,
                                 refr_index, _p(ve))
        self.n_targets += 1; self.tri_counts.append(t.shape[0])

    def set_receivers(self, spheres):
        n = len(spheres)
        c = np.array([s["centre"] for s in spheres], np.float64).reshape(n, 3)
        arrs = [np.array([s[k] for s in spheres], np.float64) for k in ("radius", "minTheta", "maxTheta", "minPhi", "maxPhi")]
        lib().orc_set_receivers(self.h, n, _p(c), *[_p(a) for a in arrs])

    def trace(self, origin, tx_span, tx_dir, width, max_refl, max_refr=0, smooth=True, ray_first=0, ray_stride=1,
              n_rays=None, use_bvh=False, threads=1, debug=True, reuse_buffers=False):
        """reuse_buffers (the timed cpu_baseline legs): the output arrays of the previous call of the same shape are handed to
        the library again -- it pre-fills every row itself -- so that a timed pulse does not pay for 2.7 GB of fresh,
        page-faulting allocations; the returned arrays are then overwritten by the next call"""
        p = OPulse()
        p.rayOrigin[:] = list(origin); p.txSpan[:] = list(tx_span); p.txDir[:] = list(tx_dir)
        if max_refr > 0:
            max_refr = 2                                      # ray_tracer.cpp:604-605
        p.width = width; p.maxRefl = max_refl; p.maxRefr = max_refr; p.interpolate_smooth = 1 if smooth else 0
        if n_rays is None:
            n_rays = width ** 3
        rows = lib().orc_rows_per_ray(max_refl, max_refr) * n_rays
        D = max_refl + max_refr
        key = (rows, D, n_rays, max_refl, bool(debug))
        cached = getattr(self, "_buffers", None)
        if reuse_buffers and cached is not None and cached[0] == key:
            res, path, ang, hp, ht = cached[1]
        else:
            res = np.zeros(rows, PRD_DTYPE)
            path = np.zeros((rows, D), np.int32)
            ang = np.zeros((rows, D, 2), np.float64)
            hp = np.zeros((n_rays, max_refl + 1), np.int32) if debug else None
            ht = np.zeros((n_rays, max_refl + 1), np.float32) if debug else None
            self._buffers = (key, (res, path, ang, hp, ht)) if reuse_buffers else None
        cnt = np.zeros(4, np.uint64)
        rc = lib().orc_trace(self.h, C.byref(p), ray_first, ray_stride, n_rays, 1 if use_bvh else 0, threads,
                             _p(res), _p(path), _p(ang), _p(hp), _p(ht), _p(cnt))
        assert rc == 0
        return dict(results=res, path=path, rcs_angle=ang, hit_prim=hp, hit_t=ht,
                    counters=dict(node_visits=int(cnt[0]), tri_tests=int(cnt[1]), segments=int(cnt[2]), shaded=int(cnt[3])))


def filter_finalise(results, path, rcs_per_target, wavelength, gt, gr, carrier, cspeed):
    """ray_tracer.cpp:1190-1258 with constant RCS per target and constant gains."""
    n = results.shape[0]; D = path.shape[1]
    rx = np.zeros(n, PRD_DTYPE); rxi = np.zeros((n, D), np.int32); slots = np.zeros(n, np.uint64)
    rcs = np.ascontiguousarray(rcs_per_target, np.float64)
    R = lib().orc_filter_finalise(_p(np.ascontiguousarray(results)), _p(np.ascontiguousarray(path)), n, D, _p(rcs),
                                  wavelength, gt, gr, carrier, cspeed, _p(rx), _p(rxi), _p(slots))
    return rx[:R].copy(), rxi[:R].copy(), slots[:R].copy()


def filter_finalise_cb(results, path, rcs_angle, origin, rx_positions, tx_index, time_t, wavelength, carrier, cspeed, get_rcs, get_gain):
    """ray_tracer.cpp:1190-1258 with the simulator's callbacks:
    get_rcs(targ, az, el, wl) -> float; get_gain(is_rx, index, (vx, vy, vz), rot_time, wl) -> float"""
    n = results.shape[0]; D = path.shape[1]
    rx = np.zeros(n, PRD_DTYPE); rxi = np.zeros((n, D), np.int32); slots = np.zeros(n, np.uint64)
    cb_r = RCS_FN(lambda user, targ, az, el, wl: float(get_rcs(targ, az, el, wl)))
    cb_g = GAIN_FN(lambda user, is_rx, index, vx, vy, vz, t, wl: float(get_gain(is_rx, index, (vx, vy, vz), t, wl)))
    o = np.ascontiguousarray(origin, np.float64); rp = np.ascontiguousarray(rx_positions, np.float64)
    R = lib().orc_filter_finalise_cb(_p(np.ascontiguousarray(results)), _p(np.ascontiguousarray(path)),
                                     _p(np.ascontiguousarray(rcs_angle, np.float64)), n, D, _p(o), _p(rp), tx_index, time_t,
                                     wavelength, carrier, cspeed, cb_r, cb_g, None, _p(rx), _p(rxi), _p(slots))
    return rx[:R].copy(), rxi[:R].copy(), slots[:R].copy()


def aggregate_literal(rx_results, rx_intersects, cspeed, carrier, ray_total):
    """aggregation.cu:32-97 literal O(R^2 D); caller pre-fill as ray_tracer.cpp:1266-1271."""
    R = rx_results.shape[0]; D = rx_intersects.shape[1] if rx_intersects.ndim == 2 else 0
    res = rx_results.copy()
    npath = np.zeros(R); power = np.zeros(R); dop = np.zeros(R); delay = np.zeros(R); phase = np.zeros(R)
    pm = np.full(R, ray_total + 1, np.int32)
    lib().orc_aggregate_literal(_p(res), _p(np.ascontiguousarray(rx_intersects, np.int32)), R, D, cspeed, carrier,
                                _p(npath), _p(power), _p(dop), _p(delay), _p(phase), _p(pm))
    return dict(results=res, npath=npath, power_sum=power, doppler_sum=dop, delay=delay, phase=phase, pathMatch=pm)


def cube_accumulate(cube, pulse, rx_results, t0, dt, cspeed, carrier, lit=None):
    """adds one pulse's contributions to cube (complex128 [n_rx][n_pulses][n_bins], in place).  lit = None: per received ray
    (rx_results = the FINALISED received rays); lit = aggregate_literal(...) output: per unique path (its results / delay /
    phase / pathMatch)"""
    assert cube.dtype == np.complex128 and cube.flags.c_contiguous
    n_rx, n_pulses, n_bins = cube.shape
    if lit is None:
        res = np.ascontiguousarray(rx_results); dl = ph = pm = None; mode = 0
    else:
        res = np.ascontiguousarray(lit["results"]); dl = np.ascontiguousarray(lit["delay"]); ph = np.ascontiguousarray(lit["phase"])
        pm = np.ascontiguousarray(lit["pathMatch"], np.int32); mode = 1
    lib().orc_cube(_p(res), res.shape[0], _p(dl), _p(ph), _p(pm), mode, n_rx, n_pulses, n_bins, pulse, t0, dt, cspeed, carrier,
                   cube.ctypes.data_as(C.c_void_p))
    return cube


def unique_paths(path_match):
    pm = np.ascontiguousarray(path_match, np.int32); out = np.zeros(pm.shape[0], np.int32)
    n = lib().orc_unique_paths(_p(pm), pm.shape[0], _p(out))
    return out[:n].copy()


def atan2f(y, x):
    return lib().orc_atan2f(float(np.float32(y)), float(np.float32(x)))


def libm_atan2f(y, x):
    return lib().orc_libm_atan2f(float(np.float32(y)), float(np.float32(x)))


COVERAGE_NAMES = ["phi_low", "phi_high", "win_minphi", "win_maxphi", "capture_region1", "capture_region2", "both_roots",
                  "second_root_only", "recapture", "recapture_reflected", "direct_capture", "earth_tested", "earth_root0",
                  "earth_root1", "earth_both", "earth_miss"]


def coverage(reset=True):
    """branch coverage of the miss program (ray_tracer.cu:260-478) since the last reset: dict name -> count"""
    out = np.zeros(len(COVERAGE_NAMES), np.uint64)
    n = lib().orc_coverage(_p(out), 1 if reset else 0)
    assert n == len(COVERAGE_NAMES)
    return {k: int(v) for k, v in zip(COVERAGE_NAMES, out)}
