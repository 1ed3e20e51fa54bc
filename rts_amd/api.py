"""Thin Python harness over the C-ABI (tests and bench.py drive the product through this).

Nothing here computes: every method marshals numpy arrays into one call of librts_amd.so.
The production caller is C++ (include/rts_adapter.hpp, rs::RTS); see INTEGRATION.md.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import PRD_DTYPE, RESPONSE_DTYPE, GROUP_DTYPE, check, ptr


# ------------------------------------------------------------------------------- host scene helpers
def rect_mesh(w, h, d, yaw=0.0, pitch=0.0, roll=0.0):
    v = np.zeros((8, 3)); t = np.zeros((12, 3), np.uint32); n = np.zeros((12, 3))
    check(L.lib().rts_rect_mesh(w, h, d, yaw, pitch, roll, ptr(v), ptr(t), ptr(n)))
    return v, t, n


def sphere_mesh(subdivisions, radius, yaw=0.0, pitch=0.0, roll=0.0):
    nv = C.c_uint32(); nt = C.c_uint32()
    check(L.lib().rts_sphere_mesh(subdivisions, radius, yaw, pitch, roll, None, C.byref(nv), None, C.byref(nt), None))
    v = np.zeros((nv.value, 3)); t = np.zeros((nt.value, 3), np.uint32); n = np.zeros((nv.value, 3))
    check(L.lib().rts_sphere_mesh(subdivisions, radius, yaw, pitch, roll, ptr(v), C.byref(nv), ptr(t), C.byref(nt), ptr(n)))
    return v, t, n


def file_mesh(v_file, n_file, yaw=0.0, pitch=0.0, roll=0.0):
    nt = C.c_uint32(0)
    check(L.lib().rts_file_mesh(v_file.encode(), n_file.encode(), yaw, pitch, roll, None, None, None, C.byref(nt)))
    v = np.zeros((3 * nt.value, 3)); t = np.zeros((nt.value, 3), np.uint32); n = np.zeros((3 * nt.value, 3))
    check(L.lib().rts_file_mesh(v_file.encode(), n_file.encode(), yaw, pitch, roll, ptr(v), ptr(t), ptr(n), C.byref(nt)))
    return v, t, n


def vertex_rotation(verts, yaw, pitch, roll):
    v = np.ascontiguousarray(verts, np.float64).copy()
    check(L.lib().rts_vertex_rotation(ptr(v), v.shape[0], yaw, pitch, roll))
    return v


def rotation_matrix(yaw, pitch, roll):
    r = np.zeros(9)
    check(L.lib().rts_rotation_matrix(yaw, pitch, roll, ptr(r)))
    return r


def rx_sphere(pos, az, el, radius, theta_span, phi_span):
    out = L.RtsReceiverSphere(); p = np.ascontiguousarray(pos, np.float64)
    check(L.lib().rts_rx_sphere(ptr(p), az, el, radius, theta_span, phi_span, C.byref(out)))
    return dict(centre=np.array(out.centre[:]), radius=out.radius, minTheta=out.min_theta, maxTheta=out.max_theta,
                minPhi=out.min_phi, maxPhi=out.max_phi)


def merge_groups(groups, depth):
    g = np.ascontiguousarray(groups, GROUP_DTYPE)
    out = np.zeros(max(len(g), 1), GROUP_DTYPE); n = C.c_uint32(len(out))
    check(L.lib().rts_merge_groups(ptr(g), len(g), depth, ptr(out), C.byref(n)))
    return out[:n.value].copy()


def groups_to_responses(groups):
    g = np.ascontiguousarray(groups, GROUP_DTYPE)
    out = np.zeros(max(len(g), 1), RESPONSE_DTYPE); n = C.c_uint32(0)
    check(L.lib().rts_groups_to_responses(ptr(g), len(g), ptr(out), len(out), C.byref(n)))
    return out[:n.value].copy()


def kernel_wrapper(rx_results, rx_intersects, cspeed, carrier, ray_total, max_threads=1024, max_blocks=65535):
    """rs::kernel_wrapper with the caller-side pre-fill of ray_tracer.cpp:1266-1271."""
    R = rx_results.shape[0]; D = rx_intersects.shape[1] if rx_intersects.ndim == 2 else 0
    res = np.ascontiguousarray(rx_results, PRD_DTYPE).copy()
    paths = np.ascontiguousarray(rx_intersects, np.int32)
    npath = np.zeros(R); power = np.zeros(R); dop = np.zeros(R); delay = np.zeros(R); phase = np.zeros(R)
    pm = np.full(R, ray_total + 1, np.int32)
    check(L.lib().rts_kernel_wrapper(ptr(res), ptr(paths), R, D, max_threads, max_blocks, cspeed, carrier, ptr(npath),
                                     ptr(power), ptr(dop), ptr(delay), ptr(phase), ptr(pm)))
    return dict(results=res, delay=delay, phase=phase, pathMatch=pm)


INTERLEAVE_LIST = 0xffffffff


def deal_tiles(records, total_rays, tile, parts):
    """rts_deal_tiles (host code): plan tiles of `tile` launch indices, longest first, each to the worker with the least cost so far.
    Returns (part_of_tile uint32[ceil(total_rays / tile)], cost_of_part uint64[parts])."""
    r = np.ascontiguousarray(records, np.uint32)
    part = np.zeros((total_rays + tile - 1) // tile, np.uint32); cost = np.zeros(parts, np.uint64)
    check(L.lib().rts_deal_tiles(ptr(r), r.shape[0], total_rays, tile, parts, ptr(part), ptr(cost)))
    return part, cost


def build_hierarchy_host(verts, tris, split_budget=2.0):
    """rts_build_hierarchy_host: the host SAH builder on one mesh (no device): (nodes [n][32] f32 view, leaf_prim, root)"""
    v = np.ascontiguousarray(verts, np.float64); t = np.ascontiguousarray(tris, np.uint32)
    nn = C.c_uint32(0); nl = C.c_uint32(0); root = C.c_int32(-1)
    check(L.lib().rts_build_hierarchy_host(ptr(v), ptr(t), t.shape[0], split_budget, None, 0, None, 0, C.byref(nn), C.byref(nl), C.byref(root)))
    nodes = np.zeros((max(nn.value, 1), 32), np.float32); leaf = np.zeros(max(nl.value, 1), np.uint32)
    check(L.lib().rts_build_hierarchy_host(ptr(v), ptr(t), t.shape[0], split_budget, ptr(nodes), nodes.shape[0], ptr(leaf), leaf.shape[0], C.byref(nn), C.byref(nl), C.byref(root)))
    return nodes[:nn.value], leaf[:nl.value], root.value


def device_count():
    n = C.c_int(0)
    rc = L.lib().rts_device_count(C.byref(n))
    return n.value if rc == 0 else 0


import os as _os
_PY_LAP = {} if _os.environ.get("RTS_PY_LAP") == "1" else None


# ------------------------------------------------------------------------------- the tracer handle
class Tracer:
    """One RtsHandle: scene + receivers + per-pulse launch on one GPU."""

    def __init__(self, width, max_refl, max_refr=0, smooth=True, device=0, keep_all=False, count_traversal=False, device_build=None, pre_filter=True):
        p = L.RtsParams(width, max_refl, max_refr, 1 if smooth else 0, device,
                        (L.RTS_FLAG_KEEP_ALL_RAYS if keep_all else 0) | (L.RTS_FLAG_COUNT_TRAVERSAL if count_traversal else 0) |
                        (0 if device_build is None else (L.RTS_FLAG_DEVICE_BUILD if device_build else L.RTS_FLAG_HOST_BUILD)) |      # None: the library's default (device)
                         (0 if pre_filter else L.RTS_FLAG_NO_PREFILTER))
        self.h = C.c_void_p()
        check(L.lib().rts_create(C.byref(p), C.byref(self.h)))
        self.width = width; self.max_refl = max_refl; self.depth = max_refl + (2 if max_refr else 0)
        self.n_targets = 0; self.keep_all = keep_all
        self._keep = []

    def close(self):
        if self.h:
            L.lib().rts_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, meshes):
        """meshes: list of dict(tris, verts, normals, refl_coeff, refr_index) in the target's own frame."""
        arr = (L.RtsMesh * max(len(meshes), 1))()
        keep = []
        for i, m in enumerate(meshes):
            t = np.ascontiguousarray(m["tris"], np.uint32); v = np.ascontiguousarray(m["verts"], np.float64)
            n = np.ascontiguousarray(m["normals"], np.float64)
            keep += [t, v, n]
            arr[i] = L.RtsMesh(t.ctypes.data, v.ctypes.data, n.ctypes.data, t.shape[0], v.shape[0], n.shape[0], 0,
                               float(m.get("refl_coeff", 1.0)), float(m.get("refr_index", 1.0)))
        check(L.lib().rts_set_scene(self.h, arr, len(meshes)))
        self.n_targets = len(meshes)

    def share_scene(self, other):
        """rts_share_scene: use `other`'s immutable scene (meshes, hierarchy, leaf order) instead of an own copy"""
        check(L.lib().rts_share_scene(self.h, other.h))
        self.n_targets = other.n_targets

    def scene_info(self):
        s = L.RtsSceneInfo()
        check(L.lib().rts_scene_info(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in L.RtsSceneInfo._fields_}

    def set_receivers(self, spheres):
        arr = (L.RtsReceiverSphere * max(len(spheres), 1))()
        for i, s in enumerate(spheres):
            arr[i] = L.RtsReceiverSphere((C.c_double * 3)(*s["centre"]), s["radius"], s["minTheta"], s["maxTheta"],
                                         s["minPhi"], s["maxPhi"])
        check(L.lib().rts_set_receivers(self.h, arr, len(spheres)))

    def trace(self, origin, tx_span, tx_dir, motion=None, ray_first=0, ray_count=0, want_stats=True, interleave=None):
        """motion: list of dict(position, velocity[, rotation(9)]) per target, or None to keep placement."""
        check(L.lib().rts_trace_pulse(self.h, C.byref(self._pulse(origin, tx_span, tx_dir, motion, ray_first, ray_count, interleave))))
        return self.stats() if want_stats else None     # reading the stage timers drains the stream

    def reserve(self, n_rays=0):
        """rts_reserve: allocate the per-launch device buffers now (0 = W^3 launch indices)"""
        check(L.lib().rts_reserve(self.h, n_rays))

    def trace_begin(self, origin, tx_span, tx_dir, motion=None, ray_first=0, ray_count=0, interleave=None):
        """enqueue a pulse (rts_trace_pulse_begin); trace_end() -- or any accessor -- completes it"""
        if _PY_LAP is None:
            check(L.lib().rts_trace_pulse_begin(self.h, C.byref(self._pulse(origin, tx_span, tx_dir, motion, ray_first, ray_count, interleave))))
            return
        import time                                                  # RTS_PY_LAP=1: where this call's time goes on the Python side
        t0 = time.perf_counter(); p = self._pulse(origin, tx_span, tx_dir, motion, ray_first, ray_count, interleave)
        t1 = time.perf_counter(); f = L.lib().rts_trace_pulse_begin
        t2 = time.perf_counter(); rc = f(self.h, C.byref(p))
        t3 = time.perf_counter(); check(rc)
        t4 = time.perf_counter()
        for k, v in zip(("marshal", "lookup", "call", "check"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            _PY_LAP[k] = _PY_LAP.get(k, 0.0) + v
        _PY_LAP["n"] = _PY_LAP.get("n", 0) + 1

    def block_timeline(self):
        """rts_get_block_timeline (a tracer created with RTS_TIMELINE_BLOCKS=1): dict of the last launch's block start / end times in us after the first start"""
        out = np.zeros(9, np.float64)
        check(L.lib().rts_get_block_timeline(self.h, ptr(out), 9))
        return dict(zip(("start_first", "start_p50", "start_last", "end_first", "end_p10", "end_p50", "end_p90", "end_last", "blocks"), out.tolist()))

    def trace_end(self):
        check(L.lib().rts_trace_pulse_end(self.h))

    # ---- ray sharding dealt by last-seen cost (rts_amd.h: rts_tile_records_get / _set, rts_set_tile_list; deal_tiles below)
    def tile_records_get(self):
        """cost records (uint32 per 64 launch indices) of the tiles this tracer's LAST launch traced, 0 elsewhere"""
        n = (self.width ** 3 + 63) // 64
        out = np.zeros(n, np.uint32)
        check(L.lib().rts_tile_records_get(self.h, ptr(out), n))
        return out

    def tile_records_set(self, records):
        r = np.ascontiguousarray(records, np.uint32)
        check(L.lib().rts_tile_records_set(self.h, ptr(r), r.shape[0]))

    def set_tile_list(self, tile, tile_ids):
        """the plan tiles (of `tile` launch indices, ascending) that launches with interleave=(tile, INTERLEAVE_LIST, 0) trace; no ids: an
        empty list (such launches trace nothing); tile = 0: no list any more"""
        ids = np.ascontiguousarray(tile_ids, np.uint32)
        check(L.lib().rts_set_tile_list(self.h, tile, ptr(ids) if ids.shape[0] else None, ids.shape[0]))

    def link(self, other):
        """rts_link_handles: trace kernels of linked tracers run one at a time, everything else overlaps"""
        check(L.lib().rts_link_handles(self.h, other.h))

    def _pulse(self, origin, tx_span, tx_dir, motion, ray_first, ray_count, interleave):
        # ONE RtsPulse and ONE motion array per tracer, refilled per call: the library copies what it needs before
        # rts_trace_pulse_begin returns, and a per-call ctypes allocation is a gc-tracked object -- in a long pulse loop the
        # collector's full passes over everything torch has imported then land in this function (0.15 ms per call at 256 pulses)
        p = self._pulse_struct = getattr(self, "_pulse_struct", None) or L.RtsPulse()
        p.ray_origin[:] = origin; p.tx_span[:] = tx_span; p.tx_dir[:] = tx_dir
        p.ray_first = ray_first; p.ray_count = ray_count
        p.interleave_tile, p.interleave_parts, p.interleave_part = interleave if interleave is not None else (0, 0, 0)      # (tile, parts, part)
        if motion is not None:
            assert len(motion) == self.n_targets
            marr = getattr(self, "_motion_arr", None)
            if marr is None or len(marr) != max(len(motion), 1):
                marr = self._motion_arr = (L.RtsTargetMotion * max(len(motion), 1))()
                self._motion_ptr = C.cast(marr, C.POINTER(L.RtsTargetMotion))
            for i, m in enumerate(motion):
                q = marr[i]
                q.position[:] = m["position"]; q.velocity[:] = m.get("velocity", (0.0, 0.0, 0.0))
                rot = m.get("rotation")
                if rot is not None:
                    q.rotation[:] = np.asarray(rot, np.float64).reshape(9).tolist(); q.has_rotation = 1
                else:
                    q.has_rotation = 0
            p.motion = self._motion_ptr
        else:
            p.motion = None
        return p

    def stats(self):
        s = L.RtsStats()
        check(L.lib().rts_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in L.RtsStats._fields_}

    def stats_raw(self):
        """rts_get_stats into ONE struct kept by the tracer (a caller in a per-pulse loop reads the few fields it wants)"""
        s = getattr(self, "_stats_struct", None)
        if s is None:
            s = self._stats_struct = L.RtsStats()
        check(L.lib().rts_get_stats(self.h, C.byref(s)))
        return s

    def received_count(self):
        n = C.c_uint64(0)
        check(L.lib().rts_received_count(self.h, C.byref(n)))
        return n.value

    def received(self):
        R = self.received_count(); D = self.depth
        rays = np.zeros(R, PRD_DTYPE); paths = np.zeros((R, D), np.int32); ang = np.zeros((R, D, 2)); slots = np.zeros(R, np.uint64)
        check(L.lib().rts_get_received(self.h, ptr(rays), ptr(paths), ptr(ang), ptr(slots), R))
        return dict(results=rays, path=paths, rcs_angle=ang, slots=slots)

    def received_prefetch(self):
        """rts_received_prefetch: the pulse in flight delivers its received set into the handle's pinned host mirror"""
        check(L.lib().rts_received_prefetch(self.h))

    def received_view(self):
        """rts_received_view: COPIES of the mirror's records (the views themselves only live until the next pulse)"""
        pr, pp, pa, ps = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(); n = C.c_uint64(0)
        check(L.lib().rts_received_view(self.h, C.byref(pr), C.byref(pp), C.byref(pa), C.byref(ps), C.byref(n)))
        R = n.value; D = self.depth

        def arr(p, dtype, shape):
            count = int(np.prod(shape))
            if count == 0 or not p.value:
                return np.zeros(shape, dtype)
            return np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p.value), dtype=dtype).reshape(shape).copy()
        return dict(results=arr(pr, PRD_DTYPE, (R,)), path=arr(pp, np.int32, (R, D)), rcs_angle=arr(pa, np.float64, (R, D, 2)), slots=arr(ps, np.uint64, (R,)))

    def finalise_values(self, power, doppler):
        p = np.ascontiguousarray(power, np.float64); d = np.ascontiguousarray(doppler, np.float64)
        check(L.lib().rts_finalise_values(self.h, ptr(p), ptr(d), len(p)))

    def aggregated_view(self):
        ps = [C.c_void_p() for _ in range(5)]; n = C.c_uint64(0)
        check(L.lib().rts_aggregated_view(self.h, *[C.byref(q) for q in ps], C.byref(n)))
        R = n.value
        out = {}
        for name, q, dt in zip(("power", "doppler", "delay", "phase", "pathMatch"), ps, (np.float64,) * 4 + (np.int32,)):
            out[name] = np.frombuffer((C.c_char * (R * np.dtype(dt).itemsize)).from_address(q.value), dtype=dt).copy() if R and q.value else np.zeros(R, dt)
        return out

    def all_rays(self, n_rays, rows=None):
        """full per-row buffers (keep_all): rows * n_rays records (rows = max_refl + 3 with refraction, else 1)"""
        D = self.depth; H = self.max_refl + 1
        rows = rows if rows is not None else (self.max_refl + 3 if self.depth > self.max_refl else 1)
        n = n_rays * rows
        res = np.zeros(n, PRD_DTYPE); path = np.zeros((n, D), np.int32); ang = np.zeros((n, D, 2))
        hp = np.zeros((n_rays, H), np.int32); ht = np.zeros((n_rays, H), np.float32)
        check(L.lib().rts_get_all_rays(self.h, ptr(res), ptr(path), ptr(ang), ptr(hp), ptr(ht), n))
        return dict(results=res, path=path, rcs_angle=ang, hit_prim=hp, hit_t=ht)

    def finalise_uniform(self, rcs_per_target, wavelength, gt, gr, carrier, cspeed):
        r = np.ascontiguousarray(rcs_per_target, np.float64) if rcs_per_target is not None else None
        check(L.lib().rts_finalise_uniform(self.h, ptr(r), wavelength, gt, gr, carrier, cspeed))

    def trace_end_uniform(self, rcs_per_target, wavelength, gt, gr, carrier, cspeed, cube_pulse=-1, recv_index_base=0):
        """rts_trace_pulse_end + rts_finalise_uniform (+ rts_cube_accumulate) + rts_aggregate in one call that does not wait for the
        trace when the handle's previous pulse received few rays; received_count() / stats() / groups() wait"""
        r = np.ascontiguousarray(rcs_per_target, np.float64) if rcs_per_target is not None else None
        check(L.lib().rts_trace_pulse_end_uniform(self.h, ptr(r), wavelength, gt, gr, carrier, cspeed, cube_pulse, recv_index_base))

    def aggregate(self, cspeed, carrier, recv_index_base=0, fetch=True):
        """fetch=False: only enqueue (the library reads the group table when it is first asked for: groups())"""
        check(L.lib().rts_aggregate(self.h, cspeed, carrier, recv_index_base))
        return self.groups() if fetch else None

    def groups(self):
        n = C.c_uint32(0)
        check(L.lib().rts_group_count(self.h, C.byref(n)))
        g = np.zeros(max(n.value, 1), GROUP_DTYPE)
        check(L.lib().rts_get_groups(self.h, ptr(g), len(g)))
        return g[:n.value].copy()

    def aggregated(self):
        R = self.received_count()
        rays = np.zeros(R, PRD_DTYPE); delay = np.zeros(R); phase = np.zeros(R); pm = np.zeros(R, np.int32)
        check(L.lib().rts_get_aggregated(self.h, ptr(rays), ptr(delay), ptr(phase), ptr(pm), R))
        return dict(results=rays, delay=delay, phase=phase, pathMatch=pm)

    def cube_attach(self, n_rx, n_pulses, n_bins, t0, dt, device_ptr=None):
        """complex return cube [n_rx][n_pulses][n_bins]; device_ptr = data_ptr() of a zeroed complex128 device tensor, or None"""
        q = L.RtsCubeParams(n_rx, n_pulses, n_bins, 0, t0, dt)
        check(L.lib().rts_cube_attach(self.h, C.byref(q), C.c_void_p(device_ptr) if device_ptr else None))
        self._cube_shape = (n_rx, n_pulses, n_bins)

    def cube_accumulate(self, pulse_index, cspeed, carrier):
        check(L.lib().rts_cube_accumulate(self.h, pulse_index, cspeed, carrier))

    def cube_accumulate_paths(self, pulse_index):
        """one contribution per unique (receiver, path) of the pulse: the group values of rts_aggregate"""
        check(L.lib().rts_cube_accumulate_paths(self.h, pulse_index))

    def cube_doppler(self, n_fft, device_ptr=None, fetch=True):
        """slow-time DFT over the pulse axis (n_fft: power of two >= n_pulses): complex [n_rx][n_fft][n_bins]"""
        check(L.lib().rts_cube_doppler(self.h, n_fft, C.c_void_p(device_ptr) if device_ptr else None))
        if not fetch:
            return None
        out = np.zeros((self._cube_shape[0], n_fft, self._cube_shape[2], 2), np.float64)
        check(L.lib().rts_cube_doppler_get(self.h, ptr(out), out.size))
        return out[..., 0] + 1j * out[..., 1]

    def cube(self):
        out = np.zeros(self._cube_shape + (2,), np.float64)
        check(L.lib().rts_cube_get(self.h, ptr(out), out.size))
        return out[..., 0] + 1j * out[..., 1]

    def bvh(self):
        """static target-space hierarchy: (nodes [n][32] float32 view of the 128-byte records, leaf_prim, roots)"""
        s = self.stats()
        nl = C.c_uint32(0)
        check(L.lib().rts_get_bvh(self.h, None, None, None, 0, 0, C.byref(nl)))        # leaf slots (>= primitives: split references)
        nodes = np.zeros((max(s["n_nodes"], 1), 32), np.float32); leaf = np.zeros(max(nl.value, 1), np.uint32)
        roots = np.zeros(max(self.n_targets, 1), np.int32)
        check(L.lib().rts_get_bvh(self.h, ptr(nodes), ptr(leaf), ptr(roots), nodes.shape[0], leaf.shape[0], C.byref(nl)))
        return nodes[:s["n_nodes"]], leaf[:nl.value], roots[:self.n_targets]

    def self_test_math(self, y, x, a, b):
        y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
        n = len(y); at = np.zeros(n, np.float32); dv = np.zeros(n); sq = np.zeros(n)
        check(L.lib().rts_self_test_math(self.h, ptr(y), ptr(x), ptr(at), ptr(a), ptr(b), ptr(dv), ptr(sq), n))
        return at, dv, sq
