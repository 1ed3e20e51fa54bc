"""Loader + ctypes signatures for librts_amd.so (the C-ABI of include/rts_amd.h).

The library is built in-tree (rts_amd/librts_amd.so) by `make -C rts_amd/csrc` /
__graft_entry__.build().  There is no Python or CPU implementation behind this module: if the
shared library is missing or cannot be loaded, importing a compute entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTS_AMD_LIB", os.path.join(_HERE, "librts_amd.so"))   # override: kernel A/B experiments
CSRC = os.path.join(_HERE, "csrc")

RTS_OK, RTS_ERR_INVALID, RTS_ERR_NO_DEVICE, RTS_ERR_HIP, RTS_ERR_UNSUPPORTED, RTS_ERR_CAPACITY, RTS_ERR_IO = range(7)
RTS_FLAG_KEEP_ALL_RAYS = 1
RTS_FLAG_COUNT_TRAVERSAL = 2
RTS_FLAG_DEVICE_BUILD = 4
RTS_FLAG_HOST_BUILD = 16
RTS_FLAG_NO_PREFILTER = 8
RTS_MAX_DEPTH = 16
RTS_BASE_USE_ROWS = 0xffffffffffffffff

# PerRayData (include/rts_prd.h == reference ray_tracer.h:13-28)
PRD_DTYPE = np.dtype({
    "names": ["rayLength", "refrIndex", "reflDepth", "refrDepth", "maxRayIndex", "rayDirection",
              "firstHitPoint", "prevHitPoint", "power", "doppler", "received", "end"],
    "formats": ["<f8", ("<f8", 2), "<u4", "<u4", "<u4", ("<f8", 3), ("<f8", 3), ("<f8", 3), "<f8", "<f8", "<i4", "u1"],
    "offsets": [0, 16, 32, 36, 40, 48, 72, 96, 120, 128, 136, 140],
    "itemsize": 144,
})


class RtsParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("max_refl", C.c_uint32), ("max_refr", C.c_uint32),
                ("interpolate_smooth", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32)]


class RtsMesh(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("vertices", C.c_void_p), ("normals", C.c_void_p),
                ("n_triangles", C.c_uint32), ("n_vertices", C.c_uint32), ("n_normals", C.c_uint32),
                ("reserved", C.c_uint32), ("refl_coeff", C.c_double), ("refr_index", C.c_double)]


class RtsTargetMotion(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("velocity", C.c_double * 3), ("rotation", C.c_double * 9),
                ("has_rotation", C.c_int32), ("reserved", C.c_int32)]


class RtsReceiverSphere(C.Structure):
    _fields_ = [("centre", C.c_double * 3), ("radius", C.c_double), ("min_theta", C.c_double),
                ("max_theta", C.c_double), ("min_phi", C.c_double), ("max_phi", C.c_double)]


class RtsPulse(C.Structure):
    _fields_ = [("ray_origin", C.c_double * 3), ("tx_span", C.c_double * 3), ("tx_dir", C.c_double * 2),
                ("ray_first", C.c_uint64), ("ray_count", C.c_uint64), ("motion", C.POINTER(RtsTargetMotion)),
                ("interleave_tile", C.c_uint32), ("interleave_parts", C.c_uint32), ("interleave_part", C.c_uint32), ("reserved", C.c_uint32)]


class RtsCubeParams(C.Structure):
    _fields_ = [("n_rx", C.c_uint32), ("n_pulses", C.c_uint32), ("n_bins", C.c_uint32), ("reserved", C.c_uint32),
                ("t0", C.c_double), ("dt", C.c_double)]


class RtsPlanItem(C.Structure):
    _fields_ = [("pulse", C.c_uint32), ("interleave_tile", C.c_uint32), ("interleave_parts", C.c_uint32), ("interleave_part", C.c_uint32),
                ("ray_first", C.c_uint64), ("ray_count", C.c_uint64)]


RTS_SHARD_PULSES, RTS_SHARD_RAYS, RTS_SHARD_PULSES_WHOLE = 0, 1, 2


class RtsSceneInfo(C.Structure):
    _fields_ = [("n_targets", C.c_uint32), ("n_prims", C.c_uint32), ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32),
                ("handles_sharing", C.c_uint32), ("builder", C.c_uint32), ("build_ms", C.c_double),
                ("shared_device_bytes", C.c_uint64), ("handle_device_bytes", C.c_uint64), ("version_bytes", C.c_uint64)]


class RtsStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("segments", C.c_uint64), ("shaded", C.c_uint64), ("received", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("n_prims", C.c_uint32), ("n_nodes", C.c_uint32),
                ("ms_scene", C.c_float), ("ms_trace", C.c_float), ("ms_compact", C.c_float), ("ms_aggregate", C.c_float),
                ("bvh_rebuilt", C.c_uint32), ("stack_overflows", C.c_uint32), ("walked_segments", C.c_uint64), ("coop_tiles", C.c_uint32), ("cost_records_dropped", C.c_uint32)]


RESPONSE_DTYPE = np.dtype([("ray", "<u8"), ("rx", "<i4"), ("n", "<u4"), ("power", "<f8"), ("delay", "<f8"),
                           ("doppler", "<f8"), ("phase", "<f8")])
GROUP_DTYPE = np.dtype([("rx", "<i4"), ("direct", "<u4"), ("path", "<i4", (RTS_MAX_DEPTH,)), ("min_ray", "<u8"),
                        ("n", "<f8"), ("sum_sqrt_power", "<f8"), ("sum_delay", "<f8"), ("sum_phase", "<f8"),
                        ("sum_doppler", "<f8")])
assert RESPONSE_DTYPE.itemsize == 48 and GROUP_DTYPE.itemsize == 120


def is_stale():
    """True if librts_amd.so is missing or was built from other sources than the tree's (hash baked in by the Makefile)"""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        return build_id() != source_hash()
    except (OSError, AttributeError):              # unloadable, or a library from before rts_build_id
        return True


def source_hash():
    """the hash rts_amd/csrc/Makefile bakes into the library (rts_build_id): SHA-256, first 16 hex digits, of the sources in
    byte order of their names, then the two public headers"""
    import hashlib
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")) and f != "rts_build_id.h")
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, f) for f in names] + [os.path.join(_HERE, "..", "include", f) for f in ("rts_amd.h", "rts_prd.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_id(path=None):
    """rts_build_id() of a built library (default: the one lib() loads), without going through lib()'s prototypes"""
    L = C.CDLL(path or LIB_PATH)
    L.rts_build_id.restype = C.c_char_p
    return L.rts_build_id().decode()


def require_built():
    """bench.py and the timed tools never build: under `rocprofv3 -- python bench.py` a make -> sh -> hipcc chain would be
    an exec from a process whose GPU the profiler's preload has already initialised.  Build beforehand
    (__graft_entry__.build() or make -C rts_amd/csrc).  A library that was built from other sources than the tree's is an
    error too (file times do not survive the copy to a GPU box; the hash baked into the library does) -- unless RTS_AMD_LIB
    names another build on purpose (same-box A/B runs)."""
    if not os.path.exists(LIB_PATH):
        raise SystemExit("librts_amd.so is not built: run `python __graft_entry__.py` (or make -C rts_amd/csrc) first")
    if "RTS_AMD_LIB" not in os.environ:
        have, want = build_id(), source_hash()
        if have != want:
            raise SystemExit("librts_amd.so is stale: built from sources %s, the tree is %s -- run `python __graft_entry__.py` first" % (have, want))


def build(force=False, verbose=False):
    """Compile librts_amd.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or is_stale():
        cmd = ["make", "-C", CSRC, "-j", "5"] + ([] if verbose else ["-s"])
        subprocess.check_call(cmd)
    return LIB_PATH


_lib = None

# every symbol include/rts_amd.h declares
EXPORTS = ["rts_create", "rts_destroy", "rts_last_error", "rts_device_count", "rts_set_scene", "rts_share_scene", "rts_scene_info", "rts_set_receivers",
           "rts_trace_pulse", "rts_reserve", "rts_trace_pulse_begin", "rts_trace_pulse_end", "rts_link_handles", "rts_get_stats", "rts_get_block_timeline", "rts_received_count", "rts_get_received", "rts_get_all_rays",
           "rts_finalise_uniform", "rts_trace_pulse_end_uniform", "rts_aggregate", "rts_group_count", "rts_get_groups", "rts_get_aggregated",
           "rts_merge_groups", "rts_groups_to_responses", "rts_kernel_wrapper", "rts_vertex_rotation",
           "rts_rotation_matrix", "rts_rect_mesh", "rts_sphere_mesh", "rts_file_mesh", "rts_rx_sphere", "rts_get_bvh",
           "rts_build_id", "rts_bind_host_to_device", "rts_get_lane_stats", "rts_get_walk_stats", "rts_self_test_math", "rts_cube_attach", "rts_cube_accumulate", "rts_cube_get", "rts_cube_accumulate_paths", "rts_cube_doppler", "rts_cube_doppler_get", "rts_plan_cpi", "rts_cube_reduce", "rts_kernel_wrapper_on",
           "rts_received_prefetch", "rts_received_view", "rts_finalise_values", "rts_aggregated_view", "rts_build_hierarchy_host",
           "rts_tile_records_get", "rts_tile_records_set", "rts_deal_tiles", "rts_set_tile_list"]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librts_amd.so is not built (run `make -C rts_amd/csrc` or __graft_entry__.build()); "
                           "there is no Python/CPU fallback for the RTS hot path")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, dbl, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_int32
    L.rts_last_error.restype = C.c_char_p
    sig = {
        "rts_create": [C.POINTER(RtsParams), C.POINTER(vp)],
        "rts_destroy": [vp],
        "rts_device_count": [C.POINTER(C.c_int)],
        "rts_set_scene": [vp, C.POINTER(RtsMesh), u32],
        "rts_share_scene": [vp, vp],
        "rts_scene_info": [vp, C.POINTER(RtsSceneInfo)],
        "rts_set_receivers": [vp, C.POINTER(RtsReceiverSphere), u32],
        "rts_trace_pulse": [vp, C.POINTER(RtsPulse)],
        "rts_reserve": [vp, u64],
        "rts_trace_pulse_begin": [vp, C.POINTER(RtsPulse)],
        "rts_get_block_timeline": [vp, vp, u32],
        "rts_trace_pulse_end": [vp],
        "rts_link_handles": [vp, vp],
        "rts_get_stats": [vp, C.POINTER(RtsStats)],
        "rts_received_count": [vp, C.POINTER(u64)],
        "rts_get_received": [vp, vp, vp, vp, vp, u64],
        "rts_get_all_rays": [vp, vp, vp, vp, vp, vp, u64],
        "rts_finalise_uniform": [vp, vp, dbl, dbl, dbl, dbl, dbl],
        "rts_build_hierarchy_host": [vp, vp, u32, dbl, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_int32)],
        "rts_received_prefetch": [vp],
        "rts_received_view": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)],
        "rts_finalise_values": [vp, vp, vp, u64],
        "rts_aggregated_view": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)],
        "rts_aggregate": [vp, dbl, dbl, u64],
        "rts_trace_pulse_end_uniform": [vp, vp, dbl, dbl, dbl, dbl, dbl, C.c_int32, u64],
        "rts_group_count": [vp, C.POINTER(u32)],
        "rts_get_groups": [vp, vp, u32],
        "rts_get_aggregated": [vp, vp, vp, vp, vp, u64],
        "rts_merge_groups": [vp, u32, u32, vp, C.POINTER(u32)],
        "rts_groups_to_responses": [vp, u32, vp, u32, C.POINTER(u32)],
        "rts_kernel_wrapper": [vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, dbl, dbl, vp, vp, vp, vp, vp, vp],
        "rts_kernel_wrapper_on": [vp, vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, dbl, dbl, vp, vp, vp, vp, vp, vp],
        "rts_vertex_rotation": [vp, u32, C.c_float, C.c_float, C.c_float],
        "rts_rotation_matrix": [C.c_float, C.c_float, C.c_float, vp],
        "rts_rect_mesh": [C.c_float] * 6 + [vp, vp, vp],
        "rts_sphere_mesh": [u32, C.c_float, C.c_float, C.c_float, C.c_float, vp, C.POINTER(u32), vp, C.POINTER(u32), vp],
        "rts_file_mesh": [C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.c_float, vp, vp, vp, C.POINTER(u32)],
        "rts_rx_sphere": [vp, dbl, dbl, dbl, dbl, dbl, C.POINTER(RtsReceiverSphere)],
        "rts_get_bvh": [vp, vp, vp, vp, u32, u32, vp],
        "rts_cube_attach": [vp, C.POINTER(RtsCubeParams), vp],
        "rts_cube_accumulate": [vp, u32, dbl, dbl],
        "rts_cube_get": [vp, vp, u64], "rts_get_lane_stats": [vp, vp], "rts_get_walk_stats": [vp, vp, C.c_uint32], "rts_bind_host_to_device": [C.c_int, C.POINTER(C.c_int)],
        "rts_cube_accumulate_paths": [vp, u32], "rts_cube_doppler": [vp, u32, vp], "rts_cube_doppler_get": [vp, vp, u64],
        "rts_plan_cpi": [u64, u32, u32, u32, u32, u32, u32, vp, u32, C.POINTER(u32)],
        "rts_cube_reduce": [vp, u32, C.c_int],
        "rts_tile_records_get": [vp, vp, u32], "rts_tile_records_set": [vp, vp, u32], "rts_deal_tiles": [vp, u32, u64, u32, u32, vp, vp], "rts_set_tile_list": [vp, u32, vp, u32],
        "rts_self_test_math": [vp, vp, vp, vp, vp, vp, vp, vp, u32],
    }
    for name, args in sig.items():
        fn = getattr(L, name, None)
        if fn is None:                     # an older build named by RTS_AMD_LIB (same-box A/B runs): its missing entry points raise on use
            continue
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = L
    return L


class RtsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rts error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != RTS_OK:
        raise RtsError(rc, lib().rts_last_error().decode(errors="replace"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
