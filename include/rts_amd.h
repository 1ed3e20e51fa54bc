/* rts_amd.h -- C-ABI of the MI355X-native RTS hot path (librts_amd.so).
 *
 * Drop-in boundary for the ray-traced radar return path of ymartin101/RTS:
 *   ray launch -> closest triangle hit (static target-space BVH4, f64 triangle test) -> reflect shading ->
 *   receiver-sphere capture -> host/device finalisation -> per-receiver path aggregation.
 * Each entry point cites the reference interface it replaces (file:line in the reference
 * repository).  Plain pointers and sizes only: no HIP, torch or C++ types.
 *
 * Conventions
 *   - every function returns an int status (RTS_OK == 0); nothing ever exit()s or aborts
 *     (the reference aborts the process: RT_CHECK_ERROR, aggregation.cu:17-27);
 *     rts_last_error() returns a description of the last failure on the calling thread.
 *   - the library owns all device memory; the caller owns every host array it passes.
 *   - a handle is not thread-safe: one handle per host thread (and per GPU).
 *   - there is NO CPU fallback: every compute entry point fails with RTS_ERR_NO_DEVICE when
 *     no gfx950 device is usable.
 */
#ifndef RTS_AMD_H
#define RTS_AMD_H

#include <stdint.h>
#include "rts_prd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtsContext* RtsHandle;

enum {
    RTS_OK = 0,
    RTS_ERR_INVALID = 1,      /* bad argument                                       */
    RTS_ERR_NO_DEVICE = 2,    /* no usable HIP device / kernels not loadable        */
    RTS_ERR_HIP = 3,          /* a HIP runtime call failed                          */
    RTS_ERR_UNSUPPORTED = 4,  /* feature outside the built scope    */
    RTS_ERR_CAPACITY = 5,     /* caller buffer too small                            */
    RTS_ERR_IO = 6            /* mesh file could not be read                        */
};

/* Launch-invariant parameters: rsParameters::GetRTSVariables() and friends
 * (ray_tracer.cpp:600-605, 645-648). */
typedef struct RtsParams {
    uint32_t width;              /* W: rays per lattice dimension, rayTotal = W^3 (rts_vars.x)          */
    uint32_t max_refl;           /* h_maxReflDepth (rts_vars.y)                                         */
    uint32_t max_refr;           /* h_maxRefrDepth (rts_vars.z); >0 is clamped to 2 as the reference:
                                    every output buffer then has max_refl + 3 rows per launch index,
                                    row = launch index + k * W^3 (ray_tracer.cpp:604-626)               */
    uint32_t interpolate_smooth; /* rsParameters::interpolate_smooth()                                  */
    int32_t device;              /* HIP device ordinal                                                  */
    uint32_t flags;              /* RTS_FLAG_*                                                          */
} RtsParams;

#define RTS_FLAG_KEEP_ALL_RAYS 1u /* also keep the full per-ray output buffers of the reference
                                     (dbuf_results / dbuf_targ_intersect / dbuf_rcs_angle for EVERY
                                     launch index) plus the per-segment hit trace -- parity/debug */
#define RTS_FLAG_DEVICE_BUILD 4u  /* (the default since round 3; kept for callers that set it) rts_set_scene builds the hierarchy ON THE
                                   * DEVICE: slab-split references, top-down binned SAH level by level, 4-wide collapse, compaction --
                                   * milliseconds for 10^5-10^6 triangles (the reference has OptiX rebuild its acceleration on the
                                   * device every pulse, ray_tracer.cpp:1126-1130; here once per scene) */
#define RTS_FLAG_HOST_BUILD 16u   /* build the hierarchy with the host SAH builder instead (rts_sah.cpp: seconds; a slightly better tree
                                   * on meshes with fans of sliver triangles) */
#define RTS_FLAG_NO_PREFILTER 8u   /* primary rays skip the conservative f32 pre-filter (direction mask over the placed triangles +
                                     widened receiver spheres) that lets rays which can meet nothing bypass the exact ray
                                     generation and the walk.  Results are identical either way (tested); the filter also switches
                                     itself off while more than half of a handle's launch indices hit.  RTS_PRIMARY_MASK=0 does the
                                     same from the environment. */
#define RTS_FLAG_COUNT_TRAVERSAL 2u /* run the counting build of the trace kernel (node visits and
                                     triangle tests per segment, for the roofline accounting)        */

/* One target mesh in its own frame, i.e. after the t = 0 rotation of the mesh builders and
 * BEFORE the per-pulse displacement (ray_tracer.cpp:963-987).  Mirrors the per-instance
 * buffers dbuf_triangles / dbuf_triVertices / dbuf_normals and variables d_targReflCoeff /
 * d_targRefrIndex (ray_tracer.cpp:1043-1114).  n_normals > n_vertices selects the "rect"
 * per-face normal rule of triangle_mesh.cu:178-180. */
typedef struct RtsMesh {
    const uint32_t* triangles;   /* [n_triangles][3] vertex indices                     */
    const double* vertices;      /* [n_vertices][3]                                     */
    const double* normals;       /* [n_normals][3]                                      */
    uint32_t n_triangles, n_vertices, n_normals;
    uint32_t reserved;
    double refl_coeff;           /* Target::GetReflCoeff()                              */
    double refr_index;           /* Target::GetRefrIndex()                              */
} RtsMesh;

/* Per-pulse placement of one target: ray_tracer.cpp:941-948 (positions), :993-1007 (time
 * varying rotation applied to the t = 0 mesh), :1010-1014 (displacement), :1144-1145
 * (velocity = (pos(t + Ts) - pos(t)) / Ts). */
typedef struct RtsTargetMotion {
    double position[3];
    double velocity[3];
    double rotation[9];          /* row-major R_total = Rz*Ry*Rx, used only if has_rotation; must be a rotation
                                  * (|R^T R - I| < 1e-3): the hierarchy is rigid, else RTS_ERR_UNSUPPORTED   */
    int32_t has_rotation;
    int32_t reserved;
} RtsTargetMotion;

/* Receiver capture sphere: dbuf_sphCentre / sphRadius / min,maxTheta / min,maxPhi
 * (ray_tracer.cu:33-38), values as computed at ray_tracer.cpp:894-918. */
typedef struct RtsReceiverSphere {
    double centre[3];
    double radius;
    double min_theta, max_theta, min_phi, max_phi;
} RtsReceiverSphere;

/* One launch (one transmitter, one pulse): d_rayOrigin / d_txSpan / d_txDir
 * (ray_tracer.cu:41-43; ray_tracer.cpp:818,881-890).  ray_first/ray_count select a contiguous
 * range of the W^3 launch indices (multi-GPU sharding); ray_count == 0 means all. */
typedef struct RtsPulse {
    double ray_origin[3];
    double tx_span[3];           /* azimuth span, elevation span, launch range           */
    double tx_dir[2];            /* boresight azimuth, elevation                         */
    uint64_t ray_first;
    uint64_t ray_count;
    const RtsTargetMotion* motion; /* [n_targets]; NULL = keep the previous placement    */
    /* Interleaved sharding inside [ray_first, ray_first + ray_count): the range is cut into tiles of
     * interleave_tile launch indices and this launch traces tiles part, part + parts, part + 2 parts, ...
     * (parts <= 1: the whole range).  Rays that hit cluster in launch-index space, so ranks that share one
     * pulse balance far better with interleaved tiles than with contiguous sub-ranges.
     * interleave_parts == RTS_INTERLEAVE_LIST: the launch traces the tiles DEALT to the handle with rts_set_tile_list
     * (interleave_tile must be that list's tile size; interleave_part is ignored). */
    uint32_t interleave_tile, interleave_parts, interleave_part, reserved;
} RtsPulse;
#define RTS_INTERLEAVE_LIST 0xffffffffu

/* Per-pulse counters and stage timings (the reference prints four wall-clock timers,
 * ray_tracer.cpp:1158,1170,1332; aggregation.cu:166). */
typedef struct RtsStats {
    uint64_t rays;               /* launch indices traced in this call                  */
    uint64_t segments;           /* rtTrace calls: primary + bounce segments            */
    uint64_t shaded;             /* closest-hit invocations that passed the depth gate   */
    uint64_t received;           /* rays with received >= 0                             */
    uint64_t node_visits;        /* BVH nodes fetched   (RTS_FLAG_COUNT_TRAVERSAL only) */
    uint64_t tri_tests;          /* triangle tests      (RTS_FLAG_COUNT_TRAVERSAL only) */
    uint32_t n_prims, n_nodes;
    float ms_scene;              /* per-pulse scene placement (vertices, normals, leaves)*/
    float ms_trace;              /* trace kernel                                        */
    float ms_compact;            /* received-ray ordering + record expansion            */
    float ms_aggregate;          /* finalise + group-by                                 */
    uint32_t bvh_rebuilt;        /* 1 if a target moved and the scene was re-placed      */
    uint32_t stack_overflows;    /* traversal stack spills to global memory             */
    uint64_t walked_segments;    /* segments that entered a target's hierarchy at all -- the rest were cleared by the primary-ray
                                    pre-filter or by the targets' bounding spheres (RTS_FLAG_COUNT_TRAVERSAL only)              */
    uint32_t coop_tiles;         /* wave tiles of this launch traced as cooperative units (one launch index per wave)           */
    uint32_t cost_records_dropped;/* tiles whose duration read as nonsense and left no cost record (never, since the records are
                                    taken on the constant-rate counter; counted, not assumed)                                    */
} RtsStats;

/* One aggregated return: what ray_tracer.cpp:1301-1321 turns into an InterpPoint/Response. */
typedef struct RtsResponse {
    uint64_t ray;                /* index in the received list of the representative ray */
    int32_t rx;                  /* receiver index                                       */
    uint32_t n;                  /* rays aggregated                                      */
    double power, delay, doppler, phase;
} RtsResponse;

/* Partial sums of one (receiver, path) group -- the unit exchanged between GPUs. */
#define RTS_MAX_DEPTH 16
typedef struct RtsGroup {
    int32_t rx;
    uint32_t direct;             /* 1 if this is the direct-transmission group           */
    int32_t path[RTS_MAX_DEPTH]; /* target index per depth, -1 padded                    */
    uint64_t min_ray;            /* smallest received-list index in the group (global)   */
    double n, sum_sqrt_power, sum_delay, sum_phase, sum_doppler;
} RtsGroup;

/* ---------------------------------------------------------------- life cycle */
int rts_create(const RtsParams* params, RtsHandle* out);      /* rtContextCreate .. ray_tracer.cpp:532-800 */
int rts_destroy(RtsHandle h);                                 /* ray_tracer.cpp:1342-1360                  */
const char* rts_last_error(void);
/* 16 hex digits: SHA-256 (truncated) of the sources the library was built from (the .hip / .cpp / .h files of rts_amd/csrc + rts_amd.h + rts_prd.h,
 * in byte order of their names).  Profiles record it; bench.py refuses to price a run with counters of another build. */
const char* rts_build_id(void);
/* Restrict the calling process's threads to the CPUs of `device`'s NUMA node (one process per GPU, on the GPU's socket: kernel
 * launches and completion signals then stay on one socket -- 7 % of a pipelined pulse on a two-socket host).  Optional; call it
 * before creating handles, so that their pinned host blocks are allocated on that node too.  *numa_node (may be NULL): the node,
 * -1 if the platform does not report one (nothing is changed then). */
int rts_bind_host_to_device(int device, int* numa_node);
int rts_device_count(int* n);

/* ---------------------------------------------------------------- scene */
int rts_set_scene(RtsHandle h, const RtsMesh* meshes, uint32_t n_targets);          /* ray_tracer.cpp:1020-1117 */
int rts_set_receivers(RtsHandle h, const RtsReceiverSphere* rx, uint32_t n_rx);     /* ray_tracer.cpp:670-715,894-925 */
/* Several handles on one device (pulses in flight, rts_trace_pulse_begin/_end) can SHARE the immutable part of a scene:
 * dst gives up its own and reads src's meshes, hierarchy and leaf order (reference counted; the hierarchy is built once,
 * by the rts_set_scene of the handle that owns it).  Each handle still places, traces and post-processes into its own
 * per-pulse buffers.  A later rts_set_scene on any of the handles gives that handle a fresh scene of its own. */
int rts_share_scene(RtsHandle dst, RtsHandle src);
typedef struct RtsSceneInfo {
    uint32_t n_targets, n_prims, n_nodes, n_leaves;
    uint32_t handles_sharing;    /* handles that point at this scene                                         */
    uint32_t builder;            /* 0: host SAH (rts_sah.cpp), 1: device LBVH (rts_lbvh.hip)                 */
    double build_ms;             /* wall time of the hierarchy build inside rts_set_scene                    */
    uint64_t shared_device_bytes;/* device memory of the shared, immutable part                              */
    uint64_t handle_device_bytes;/* device memory of this handle's placement of it (leaf records, world-space
                                    vertices and normals); per-launch buffers (rts_reserve) not included       */
    uint64_t version_bytes;      /* of shared_device_bytes: the eight octant versions of the node records (0: the
                                    scene has none -- RTS_NODE_VERSIONS=0, or too large for them)               */
} RtsSceneInfo;
int rts_scene_info(RtsHandle h, RtsSceneInfo* out);

/* ---------------------------------------------------------------- launch
 * Replaces rtContextValidate/Compile/Launch3D (ray_tracer.cpp:1126-1165): places the targets,
 * re-places the scene on the device if any target moved (the hierarchy is static), traces ray_count launch indices and
 * leaves the received rays on the device, ordered by ascending launch index (the order of the
 * host scan at ray_tracer.cpp:1190).  Blocking. */
int rts_trace_pulse(RtsHandle h, const RtsPulse* pulse);

/* Pulse pipelining.  The reference runs its pulses strictly one after the other (the loop at ray_tracer.cpp:843:
 * acceleration rebuild -> rtContextLaunch3D -> host read-back -> aggregation, each blocking).  Pulses are independent,
 * so a caller may keep two (or more) handles holding the same scene, link them once, and alternate pulses between them:
 *     rts_trace_pulse_begin(hB, pulse k+1);   // enqueued, returns at once
 *     rts_trace_pulse_end(hA);                // pulse k: wait for its trace, order + expand its received rays
 *     ... rts_finalise_uniform / rts_cube_accumulate / rts_aggregate on hA ...
 * rts_trace_pulse == begin + end.  Trace kernels of linked handles execute one at a time in begin order (each has the
 * whole GPU, so rts_get_stats().ms_trace stays a single-kernel time); the scene placement of the next pulse and the
 * ordering / finalisation / aggregation of the previous one overlap with them on the handles' own HIP streams.
 * Linking is optional: un-linked handles also overlap their trace kernels (the tail of one launch -- a few slow tiles --
 * is filled by the next handle's blocks), which is faster; linking trades that for strictly serial, cleanly timed kernels.
 * Entry points that read a pulse's results end a begun pulse implicitly.  One host thread per link group. */
/* Optional: allocate (and touch) the device buffers of launches of up to n_rays launch indices now (0 = W^3) instead of
 * inside the first rts_trace_pulse; keeps multi-GB allocations out of a timed or latency-critical region. */
int rts_reserve(RtsHandle h, uint64_t n_rays);
int rts_trace_pulse_begin(RtsHandle h, const RtsPulse* pulse);
int rts_trace_pulse_end(RtsHandle h);
int rts_link_handles(RtsHandle a, RtsHandle b);               /* same device; groups grow by linking a member to a new handle */
int rts_get_stats(RtsHandle h, RtsStats* out);
/* Diagnostic (handles created with RTS_TIMELINE_BLOCKS=1; product builds): when the persistent blocks of the last launch started and ended -- out[0..2] first / median /
 * last block start, out[3..7] first / 10th percentile / median / 90th percentile / last block end, microseconds after the first start, out[8] the number of blocks. */
int rts_get_block_timeline(RtsHandle h, double* out, uint32_t n);

/* Received rays of the last pulse (ray_tracer.cpp:1186-1257 before the gain/RCS update):
 * rays[R], paths[R][D] (h_rx_intersects, D = max_refr + max_refl), rcs_angles[R][D][2],
 * slots[R] = global launch index.  Any output pointer may be NULL.  capacity in rays. */
/* Lane statistics of the last launch's traversal (counting build, RTS_FLAG_COUNT_TRAVERSAL): out3[0] walk iterations issued x 64
 * lanes, out3[1] the part issued to lanes that took part in their tile's bounce round, out3[2] walk steps actually taken. */
int rts_get_lane_stats(RtsHandle h, uint64_t* out3);
int rts_get_walk_stats(RtsHandle h, uint64_t* out, uint32_t n);   /* out[0..2] as rts_get_lane_stats; [3] segments that walked; [4] lane-steps issued to lanes that are in a bounce round but never started a walk in it (counting builds) */
int rts_received_count(RtsHandle h, uint64_t* count);
int rts_get_received(RtsHandle h, struct PerRayData* rays, int32_t* paths, double* rcs_angles, uint64_t* slots,
                     uint64_t capacity);

/* The same without copy calls, for a caller in a pulse loop that needs the received rays on the HOST (the simulator's RCS and
 * antenna-gain callbacks of ray_tracer.cpp:1198-1256; include/rts_adapter.hpp):
 *     rts_trace_pulse_begin(h, pulse); rts_received_prefetch(h);       // enqueued, returns at once
 *     ... other handles' pulses ...
 *     rts_received_view(h, &rays, &paths, &angles, NULL, &R);          // ONE wait; pointers into the handle's pinned host mirror
 *     for i < R: power[i], doppler[i] = callbacks(rays[i], paths[i], angles[i])
 *     rts_finalise_values(h, power, doppler, R); rts_aggregate(h, c, fc, 0);      // enqueued
 *     ... other handles' pulses ...
 *     rts_aggregated_view(h, &power, &doppler, &delay, &phase, &path_match, &R);   // ONE wait; what rs::kernel_wrapper returns
 * rts_received_prefetch enqueues the ordering + expansion of the pulse's received rays behind its trace and has a kernel store
 * the result into mapped host memory -- without waiting for the trace when the handle's previous pulse received at most 3 072
 * rays (1 536 with refraction chains), otherwise it only marks the pulse and the first accessor's (blocking) rts_trace_pulse_end
 * feeds the mirror.  Either way the views equal rts_get_received / rts_get_aggregated bit for bit; sets beyond the mirror
 * (4 096 rays, or what the handle has seen) are served by copies.  View pointers stay valid until the handle's next
 * rts_trace_pulse_begin and keep their content: rts_received_view's records are the set as it was at the pulse's FIRST call of it --
 * AS RECEIVED when that call precedes rts_finalise_values (the adapter's order) -- whatever rts_finalise_values / rts_aggregate /
 * rts_aggregated_view do afterwards, also for sets beyond the mirror's capacity (served from copies made once per pulse). */
int rts_received_prefetch(RtsHandle h);
int rts_received_view(RtsHandle h, const struct PerRayData** rays, const int32_t** paths, const double** rcs_angles, const uint64_t** slots,
                      uint64_t* count);
int rts_finalise_values(RtsHandle h, const double* power, const double* doppler, uint64_t count);
int rts_aggregated_view(RtsHandle h, const double** power, const double** doppler, const double** delay, const double** phase,
                        const int32_t** path_match, uint64_t* count);

/* Full per-launch-index buffers (RTS_FLAG_KEEP_ALL_RAYS): results[n] / targ_intersect[n][D] /
 * rcs_angle[n][D][2] as mapped at ray_tracer.cpp:1180-1182, plus hit_prim[n][max_refl+1] (global
 * primitive id of the closest hit of each segment of the reflection chain, -1 = miss, -2 = not
 * traced) and hit_t[n][max_refl+1] (its f32 distance). */
int rts_get_all_rays(RtsHandle h, struct PerRayData* results, int32_t* targ_intersect, double* rcs_angle,
                     int32_t* hit_prim, float* hit_t, uint64_t capacity);

/* ---------------------------------------------------------------- finalise + aggregate on the device
 * rts_finalise_uniform: the per-received-ray update of ray_tracer.cpp:1219-1253 for the case
 * where RCS is a constant per target and the antenna gains are constants (isotropic antennas):
 *   power *= prod RCS[targ_k] ; power *= wavelength^2 * gt * gr ;
 *   doppler = carrier * ((1 + Vr/c) / (1 - Vr/c) - 1), Vr = doppler / 2.
 * With SOARS antenna / RCS callbacks use rts_get_received + rs::kernel_wrapper instead. */
int rts_finalise_uniform(RtsHandle h, const double* rcs_per_target, double wavelength, double gt, double gr,
                         double carrier, double cspeed);

/* One call instead of rts_trace_pulse_end + rts_finalise_uniform (+ rts_cube_accumulate when cube_pulse >= 0) + rts_aggregate, for a
 * caller that needs no host callbacks between them (ray_tracer.cpp:1180-1285 with constant RCS / gains).  When the handle's previous
 * pulse received few rays (at most 3 072 -- 1 536 with refraction chains or a (receiver, path) key beyond 31 bits) the whole chain is
 * enqueued behind the trace WITHOUT waiting for its received count -- the kernels take it from the device -- and the call returns
 * at once; the count, the statistics and the group table come home with the first call that asks for them (rts_received_count,
 * rts_get_stats, rts_group_count, rts_get_groups, rts_get_received, ...; the handle's next rts_trace_pulse_begin at the latest).  A
 * pulse that then turns out to have received more than the chain was sized for (4 096 / 2 048 rays) is post-processed again, the
 * ordinary way, at that point.  Results are those of the four calls, bit for bit.  (Measured on an MI355X with ROCm 7: the
 * submitting thread's wait for the trace disappears, the pulse rate does not change -- DESIGN.md section 5.) */
int rts_trace_pulse_end_uniform(RtsHandle h, const double* rcs_per_target, double wavelength, double gt, double gr,
                                double carrier, double cspeed, int32_t cube_pulse, uint64_t recv_index_base);

/* rts_aggregate: myKernel1 + myKernel2 + unique paths (aggregation.cu:32-97,
 * ray_tracer.cpp:1283-1292) on the device-resident received set, as a sort/group-by.
 * recv_index_base offsets the received-list indices (multi-GPU with contiguous ranges: number of received
 * rays on lower ranks).  RTS_BASE_USE_ROWS makes RtsGroup.min_ray the GLOBAL BUFFER ROW (launch index + k W^3)
 * of the group's first ray instead: rows order rays exactly as received-list indices do, and they are comparable
 * across ranks whatever the sharding (interleaved tiles).
 * Rays are grouped by a packed (receiver, path) key of D x ceil(log2(targets + 1)) + ceil(log2(receivers)) bits,
 * D = max_refl + max_refr: one 64-bit radix sort when that fits 64 bits (every BASELINE configuration: C3 6 x 1 + 2,
 * C4 8 x 3 + 3), two or three stable passes over 64-bit words otherwise (e.g. 16 bounces in a scene of 100 targets: 115 bits). */
#define RTS_BASE_USE_ROWS 0xffffffffffffffffULL
int rts_aggregate(RtsHandle h, double cspeed, double carrier, uint64_t recv_index_base);
int rts_group_count(RtsHandle h, uint32_t* count);
int rts_get_groups(RtsHandle h, RtsGroup* groups, uint32_t capacity);
/* Per-ray aggregation outputs of the last rts_aggregate (what kernel_wrapper returns). */
int rts_get_aggregated(RtsHandle h, struct PerRayData* rays, double* delay, double* phase, int32_t* path_match,
                       uint64_t capacity);

/* Host-side: merge group tables (concatenated from several GPUs) and derive the responses
 * that ray_tracer.cpp:1290-1321 would emit.  Pure host code, no device needed. */
int rts_merge_groups(const RtsGroup* in, uint32_t n_in, uint32_t depth, RtsGroup* out, uint32_t* n_out);
int rts_groups_to_responses(const RtsGroup* groups, uint32_t n_groups, RtsResponse* out, uint32_t capacity,
                            uint32_t* n_out);

/* ---------------------------------------------------------------- complex return cube (derived product)
 * Not in the reference (SOARS' rsresponse renders the responses); the north star asks for "atomic complex
 * accumulation into per-receiver range-Doppler bins" with an RCCL reduce of the per-receiver return buffers.
 * cube[rx][pulse][bin] (complex128, interleaved re/im) += sqrt(power) * exp(j * phase) for every received,
 * finalised ray of the last pulse, with delay = rayLength / c, phase = -fmod(2 pi fc delay, 2 pi) (the phase
 * convention of aggregation.cu:59-60) and bin = floor((delay - t0) / dt); rays outside [0, n_bins) are dropped.
 * The storage may be caller-owned DEVICE memory (device_ptr != NULL, e.g. a torch tensor that is then
 * all-reduced over RCCL) or library-owned (device_ptr == NULL). */
typedef struct RtsCubeParams {
    uint32_t n_rx, n_pulses, n_bins, reserved;
    double t0, dt;
} RtsCubeParams;
int rts_cube_attach(RtsHandle h, const RtsCubeParams* params, void* device_ptr);
int rts_cube_accumulate(RtsHandle h, uint32_t pulse_index, double cspeed, double carrier);
int rts_cube_get(RtsHandle h, double* host_out, uint64_t capacity_doubles);
/* The same product per UNIQUE PATH instead of per received ray: one contribution per response the reference would emit for
 * the pulse (ray_tracer.cpp:1290-1321) -- sqrt(P_group) e^{j phase_group} at delay_group, the group values of
 * aggregation.cu:88-93 -- i.e. the reference's own (incoherent: mean of sqrt p, mean of wrapped phases) combination of
 * the rays of a path.  Needs rts_aggregate of the pulse on this handle.  rts_cube_accumulate is the COHERENT sum over the
 * rays (sum_i sqrt(p_i) e^{j phi_i}); the two differ by design, a caller uses one of them per cube. */
int rts_cube_accumulate_paths(RtsHandle h, uint32_t pulse_index);
/* Slow-time (Doppler) transform: for every receiver and range bin the n_fft-point DFT over the pulse axis, n_fft a power
 * of two with n_pulses <= n_fft <= 4096 (pulses beyond n_pulses count as zeros):
 *     out[rx][k][bin] = sum_p cube[rx][p][bin] e^{-2 pi j k p / n_fft}          (complex128, [n_rx][n_fft][n_bins])
 * device_out: caller-owned device memory of 2 n_rx n_fft n_bins doubles, or NULL: library-owned (rts_cube_doppler_get). */
int rts_cube_doppler(RtsHandle h, uint32_t n_fft, void* device_out);
int rts_cube_doppler_get(RtsHandle h, double* host_out, uint64_t capacity_doubles);

/* ---------------------------------------------------------------- several GPUs (not in the reference: it is single-GPU)
 * Rays are independent (each launch index writes only its own rows, ray_tracer.cu:227-253) and so are pulses
 * (ray_tracer.cpp:843).  rts_plan_cpi deals the n_pulses x total_rays (pulse, launch index) pairs of one coherent
 * processing interval to `world` workers -- one handle set per GPU, in one process (rts_adapter.hpp) or one process per
 * GPU (bench.py) -- and returns worker `rank`'s share as RtsPulse-ready items:
 *   RTS_SHARD_PULSES  n_pulses / world whole pulses per worker; each of the n_pulses % world left-over pulses is shared by a
 *                     group of consecutive workers in interleaved tiles (RtsPulse.interleave_*)
 *   RTS_SHARD_RAYS    every pulse is split over all workers in interleaved tiles (work per worker independent of how
 *                     n_pulses divides by world)
 *   RTS_SHARD_PULSES_WHOLE  whole pulses only, contiguous runs, the first n_pulses % world workers one pulse more (a part of a pulse
 *                     is a launch of another shape whose schedule starts from nothing: on short intervals of small pulses that costs
 *                     more than one pulse of imbalance); with fewer pulses than workers: as RTS_SHARD_PULSES
 * min_items > 1 splits items further (part p of P -> parts p and p + P of 2 P) until the worker owns that many, so that
 * it can keep min_items pulses (or parts) in flight.  Parts of one pulse are merged again through their group tables
 * (rts_aggregate with RTS_BASE_USE_ROWS, rts_merge_groups) or through their received sets ordered by RtsResponse.ray /
 * slots.  Pure host code. */
typedef struct RtsPlanItem {
    uint32_t pulse;
    uint32_t interleave_tile, interleave_parts, interleave_part;   /* parts <= 1: the whole range */
    uint64_t ray_first, ray_count;
} RtsPlanItem;
#define RTS_SHARD_PULSES 0u
#define RTS_SHARD_RAYS 1u
#define RTS_SHARD_PULSES_WHOLE 2u
#define RTS_PLAN_TILE 4096u      /* default launch indices per interleaved tile (a multiple of 64 keeps the tile-cost history) */
int rts_plan_cpi(uint64_t total_rays, uint32_t n_pulses, uint32_t rank, uint32_t world, uint32_t mode, uint32_t min_items,
                 uint32_t tile /* launch indices per interleaved tile; 0 = RTS_PLAN_TILE */, RtsPlanItem* out, uint32_t capacity,
                 uint32_t* n_out);
/* Ray sharding BALANCED BY LAST-SEEN COST (instead of the static interleave): the tiles of a pulse are dealt to the workers
 * longest-first from what every tile cost the last time any worker traced it -- one exchange of a cost table per interval.
 * (The reference is single-GPU, ray_tracer.cpp:1165 is one rtContextLaunch1D over all W^3 indices; rays are independent,
 * ray_tracer.cu:227-253, so any partition of the launch indices gives the same rows.)
 *   rts_tile_records_get   the handle's cost records (one uint32 per 64 consecutive launch indices of the W^3 lattice, n =
 *                          ceil(W^3 / 64); bits 0-29 wave time in units of 64 shader clocks, bits 30-31 the walk-length flags of
 *                          the cooperative kernel's head rule) of the tiles its LAST launch traced, 0 for every other tile:
 *                          tables of workers that traced disjoint parts of a pulse merge with a plain sum (or max)
 *   rts_tile_records_set   replaces the handle's history with a merged table: its next launches order their tiles -- and pick
 *                          the cooperative kernel's head tiles -- from what ANY worker measured
 *   rts_deal_tiles         host code, deterministic (every worker computes the same map from the same table): first the plan tiles of `tile`
 *                          launch indices (a multiple of 64) that hold walk-length flags (the cooperative kernel's candidates), in descending
 *                          cost, each to the worker holding the fewest flagged wave tiles so far; then the others in descending cost, each to
 *                          the worker with the least cost so far; tiles without a record are then dealt by COUNT (ascending, each to the worker holding the fewest tiles).  part_of_tile[ceil(total_rays / tile)] <- worker
 *   rts_set_tile_list      the plan tiles (ascending, unique, < ceil(range / tile)) the handle's launches with
 *                          interleave_parts == RTS_INTERLEAVE_LIST trace; n_ids == 0 is an EMPTY list (a worker that was dealt nothing:
 *                          its launches trace no launch index and return empty sets); tile == 0 forgets the list */
int rts_tile_records_get(RtsHandle h, uint32_t* records, uint32_t n);
int rts_tile_records_set(RtsHandle h, const uint32_t* records, uint32_t n);
int rts_deal_tiles(const uint32_t* records, uint32_t n_records, uint64_t total_rays, uint32_t tile, uint32_t parts, uint32_t* part_of_tile, uint64_t* cost_of_part /* [parts] or NULL */);
int rts_set_tile_list(RtsHandle h, uint32_t tile, const uint32_t* tile_ids, uint32_t n_ids);
/* Sum of the complex return cubes of several handles (same RtsCubeParams; one handle per GPU, or several per GPU), left in
 * EVERY handle's cube: the "RCCL reduce over the per-receiver return buffers" of a multi-GPU interval when all GPUs belong
 * to one process.  transport 0: RCCL (ncclCommInitAll + ncclAllReduce, loaded on first use) when the handles sit on
 * distinct devices and librccl can be loaded, otherwise peer copies; 1: RCCL or fail; 2: peer copies (hipMemcpyPeer +
 * add, in handle order -- bit-reproducible). */
int rts_cube_reduce(RtsHandle* handles, uint32_t n_handles, int transport);

/* ---------------------------------------------------------------- the reference's inner C-like boundary
 * Same argument list and in/out behaviour as rs::kernel_wrapper (aggregation.cuh:19-22,
 * aggregation.cu:103-184); the C++ symbol rs::kernel_wrapper is exported by the library too
 * (include/rts_adapter.hpp).  Returns a status instead of exiting. */
int rts_kernel_wrapper(struct PerRayData* h_rx_results_arr, int* h_rx_intersects_arr, unsigned int receivedRays,
                       unsigned int depthTotal, unsigned int MaxThreads, unsigned int MaxBlocks, double cspeed,
                       double carrier, double* h_npath_arr, double* h_power_arr, double* h_doppler_arr,
                       double* h_delay_arr, double* h_phase_arr, int* h_pathMatch);
/* The same on the device and stream of a handle (NULL: a process-wide context of the calling thread's current device, one
 * per device, which is what rts_kernel_wrapper and rs::kernel_wrapper use).  The handle's own received set is overwritten.
 * rs::kernel_wrapper (void in the reference, which exit(1)s on error) throws std::runtime_error on failure. */
int rts_kernel_wrapper_on(RtsHandle h, struct PerRayData* h_rx_results_arr, int* h_rx_intersects_arr, unsigned int receivedRays,
                          unsigned int depthTotal, unsigned int MaxThreads, unsigned int MaxBlocks, double cspeed,
                          double carrier, double* h_npath_arr, double* h_power_arr, double* h_doppler_arr,
                          double* h_delay_arr, double* h_phase_arr, int* h_pathMatch);

/* ---------------------------------------------------------------- host scene helpers (ray_tracer.cpp:85-504, 894-918)
 * Two-call pattern for the variable-size builders: pass NULL outputs to obtain the sizes. */
int rts_vertex_rotation(double* vertices, uint32_t n, float yaw, float pitch, float roll);          /* :156-170 */
int rts_rotation_matrix(float yaw, float pitch, float roll, double* r9);                             /* :159-162 */
int rts_rect_mesh(float w, float h, float d, float yaw, float pitch, float roll, double* vertices24,
                  uint32_t* triangles36, double* normals36);                                         /* :226-297 */
int rts_sphere_mesh(uint32_t subdivisions, float radius, float yaw, float pitch, float roll, double* vertices,
                    uint32_t* n_vertices, uint32_t* triangles, uint32_t* n_triangles, double* normals); /* :300-426 */
int rts_file_mesh(const char* v_file, const char* n_file, float yaw, float pitch, float roll, double* vertices,
                  uint32_t* triangles, double* normals, uint32_t* n_triangles);                      /* :429-504 */
int rts_rx_sphere(const double* rx_position, double azimuth, double elevation, double radius, double theta_span,
                  double phi_span, RtsReceiverSphere* out);                                          /* :894-918 */

/* ---------------------------------------------------------------- introspection (tests)
 * The static target-space hierarchy built by rts_set_scene: nodes[RtsStats.n_nodes] as stored (128-byte records: lo x,y,z /
 * hi x,y,z planes of the four children as 6 x float[4], int32 child[4] (>= 0 node, < 0 ~leaf slot, 0x7fffffff unused),
 * int32 pad[4]); leaf_prim[*n_leaves] = global primitive id per leaf slot (primitives with a non-finite vertex have
 * none; a primitive whose box is mostly empty has several slots, each boxing a part of it -- "split references");
 * roots[n_targets] = root node per target (-1: no geometry).  Any output may be NULL (sizes: call with NULLs first). */
int rts_get_bvh(RtsHandle h, void* nodes128, uint32_t* leaf_prim, int32_t* roots, uint32_t node_capacity,
                uint32_t leaf_capacity, uint32_t* n_leaves);
/* The host SAH builder (rts_sah.cpp; RTS_FLAG_HOST_BUILD) on one mesh, without a device: nodes128 / leaf_prim as rts_get_bvh returns
 * them (leaf_prim[slot] = triangle index of the mesh).  Null outputs: sizes only.  Pure host code. */
int rts_build_hierarchy_host(const double* vertices, const uint32_t* triangles, uint32_t n_triangles, double split_budget, void* nodes128,
                             uint32_t node_capacity, uint32_t* leaf_prim, uint32_t leaf_capacity, uint32_t* n_nodes, uint32_t* n_leaves,
                             int32_t* root);
int rts_self_test_math(RtsHandle h, const float* y, const float* x, float* atan2f_out, const double* a,
                       const double* b, double* div_out, double* sqrt_out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RTS_AMD_H */
