// rts_internal.h -- device-side data layout and the host context of librts_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>
#include "../../include/rts_amd.h"
#include "rts_device_math.h"

// ----------------------------------------------------------------------------- HBM layout
// BVH4 node, 128 B = one cache line, of the static target-space hierarchy (rts_sah.cpp): half the dependent fetch
// round-trips of a binary tree per ray.  SoA over the four children so that one dwordx4 load brings the same plane of
// all four (padded, f32, TARGET-SPACE) boxes.  Unused slots: empty box (never hit).
struct __attribute__((aligned(128))) RtsNode4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int32_t child[4];               // >= 0 -> node index, < 0 -> ~leaf slot, 0x7fffffff -> unused
    int32_t pad[4];
};
static_assert(sizeof(RtsNode4) == 128, "node4 size");

// Leaf record, 80 B, in the hierarchy's (depth-first) leaf order, refreshed every pulse: the three f64 WORLD-space
// vertices the f64 intersection test needs, pre-gathered (the reference gathers through dbuf_triangles ->
// dbuf_triVertices per test, triangle_mesh.cu:147-154), plus the global primitive id.
struct __attribute__((aligned(16))) RtsLeafTri {
    double p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y, p2z;
    uint32_t prim;                  // global primitive id (targets concatenated in order)
    uint32_t targ;                  // target index
};
static_assert(sizeof(RtsLeafTri) == 80, "leaf size");

struct RtsTargetDev {               // per target, per pulse
    double reflCoeff;               // d_targReflCoeff
    double refrIndex;               // d_targRefrIndex
    double vx, vy, vz;              // dbuf_targ_vel[targ]
    uint32_t tri_base;              // first global primitive id
    uint32_t perface_normals;       // triangle_mesh.cu:178 (normals.size() > vertices.size())
    // placement of the target's static hierarchy for this pulse: local = rinv * (world - pos)
    double rinv[9];                 // inverse of the pulse's rotation (identity when the target does not rotate)
    double px, py, pz;
    double cx, cy, cz, r2;          // bounding sphere of the placed target (world space, radius^2, padded)
    float ew;                       // extra origin slack of the target-space slab test [m]: covers the rounding of the world
                                    // vertices, of the exact test and of the world -> target mapping (~1e-12 of the world scale)
    int32_t root;                   // root node of the target's hierarchy, < 0: no (finite) geometry
};

// One mesh's slice of the static hierarchy (host side).
struct RtsBlasInfo { int32_t root; uint32_t n_nodes, n_leaves, depth; double lo[3], hi[3], max_abs; };

struct RtsRxDev { double cx, cy, cz, radius, minTheta, maxTheta, minPhi, maxPhi; };

// End-of-ray record written by the trace kernel (received rays; every ray in KEEP_ALL mode).
struct __attribute__((aligned(16))) RtsEndRecord {
    double rayLength, power, doppler;
    double prevx, prevy, prevz;
    double firstx, firsty, firstz;
    uint64_t path_lo, path_hi;      // (targ + 1) per depth, 8 bits each, depth 0 in the low byte
    uint32_t slot;                  // local launch index (ray_first + slot = global index)
    int32_t received;
    uint32_t reflDepth;
    uint32_t pad;
};
static_assert(sizeof(RtsEndRecord) == 112, "end record size");
// RtsEndRecord::pad : bits 0-1 chain (output row = chain * n + slot), 2-3 refrDepth, 8-15 (target + 1) of chain 0's
// refraction (path prefill of rows >= 3), 16-17 children spawned

// State of a refracted child ray parked between chains (normal_shader.cu:191-256: the copy prd_refr).
struct __attribute__((aligned(16))) RtsChildState {
    double prevx, prevy, prevz, firstx, firsty, firstz;
    double rayLength, power, doppler, refx, refy;
    float dx, dy, dz;               // new_direction of refract(), exactly
    uint32_t refrDepth, refr_code, end, pad0, pad1;
};
static_assert(sizeof(RtsChildState) == 128, "child state size");

#ifndef RTS_BLOCK
#define RTS_BLOCK 256
#endif
#define RTS_WTILE 64               // work unit of the trace kernel: launch indices per wave tile
#ifndef RTS_TILE_CTRS
#define RTS_TILE_CTRS 64           // striped draw counters of the tile queue
#endif
#define RTS_TILE_BUCKETS 1024       // bins of the tiles' counting order (rts_post.hip: k_tile_bucket_*)
#define RTS_TILE_CTR_STRIDE 32     // ... one per 128-byte line: same-LINE atomics serialise in L2 (~10 ns each) whatever their address
// XCD-AFFINE sub-orders (big launches, rts_post.hip: rts_tile_order_build): the order is cut into a head-rest segment + one segment
// per XCD (a contiguous band of the lattice holding an eighth of the cost last seen), each drawn through 8 counters of its own
#define RTS_XCD 8
#define RTS_SEG_STRIPES 8
#define RTS_TILE_CTRS_LAYOUT ((RTS_XCD + 1) * RTS_SEG_STRIPES + 8)       // counters reserved per kernel in the zero block (>= RTS_TILE_CTRS)
#define RTS_COARSE_CELLS 1024       // cost of the last launch by 1/1024 of its tile range (the bands' boundaries come from it)
// the zero block: ONE fill per launch clears [draw counters of the ordinary kernel | of the cooperative kernel | head words: cost
// sum lo, hi, head count, ticket | the launch's 16 u64 counters | the order's bins | the coarse cost cells]
#define RTS_OFF_CTR_COOP (RTS_TILE_CTRS_LAYOUT * RTS_TILE_CTR_STRIDE)
#define RTS_OFF_HEAD (2 * RTS_TILE_CTRS_LAYOUT * RTS_TILE_CTR_STRIDE)
#define RTS_OFF_LIVE (RTS_OFF_HEAD + 4)       // 1 + the number of tiles at the front of this launch's order that cost more than a dead tile (0: unknown) -- written by the order build (rts_post.hip)
#define RTS_OFF_COUNTERS (RTS_OFF_HEAD + 8)
#define RTS_OFF_BINS (RTS_OFF_COUNTERS + 32)
#define RTS_OFF_COARSE (RTS_OFF_BINS + RTS_TILE_BUCKETS)
#define RTS_ZERO_WORDS (RTS_OFF_COARSE + RTS_COARSE_CELLS)
#ifndef RTS_STACK_LDS
#define RTS_STACK_LDS 24            // traversal stack entries kept in LDS per lane
#endif
#ifndef RTS_RX_LDS
#define RTS_RX_LDS 16               // receivers whose capture spheres the trace kernel keeps in LDS (the rest are read from memory)
#endif
#define RTS_STACK_OVF 128           // further entries spilled to global memory (rare); a BVH4 node pushes up to 3 entries

// Launch constants of ray_generation (hoisted trig, ray_tracer.cu:155-203).  Device resident and
// read through a pointer: keeping these 30 doubles as by-value kernel arguments pinned ~60
// SGPRs for the whole kernel (SGPR spills to VGPR lanes).
// Primary-ray pre-filter and mask.  A primary ray that meets no triangle and no receiver sphere leaves NO trace in any output
// (not even with RTS_FLAG_KEEP_ALL_RAYS: its record is the initial payload, whatever its direction), and on C3 that is 84 % of
// the launch indices.  For those the exact f64 ray generation (two normalisations: a square root and three divisions each),
// the bounding-sphere tests, the target mapping, the root visits and the receiver quadratics buy nothing.  The trace kernel
// therefore first forms the direction in f32 (good to ~2e-6 rad) and asks two CONSERVATIVE questions -- is any triangle's
// projection near this direction (the mask below), can the ray come within a widened radius of a receiver sphere -- and only
// a ray that may hit something gets the exact treatment (everything a ray can be observed through is still computed exactly
// as before: the pre-filter can only say "certainly nothing").
// The mask: all primary rays of a launch start at the transmitter, so which of them CAN meet a triangle is a question
// about their direction alone.  Per pulse every placed triangle marks the cells of a 2-D bitmap over the beam -- perspective
// coordinates (p.u / p.b, p.v / p.b) of p = vertex - origin in a frame (b, u, v) around the boresight -- that the bounding
// rectangle of its projection touches (k_primary_mask, rts_bvh.hip); a primary ray whose own cell is clear skips the targets
// altogether (bounding spheres, target mapping, slab set-up, root visits: about half the instructions of a launch index that
// hits nothing -- 84 % of them on C3).  Conservative: a ray that hits a triangle has its direction inside the triangle's
// projection, hence inside the rectangle; the rectangle is widened by a cell, the ray's f32 coordinates are good to 1e-6 of the
// extent.  Disabled for the pulse (n = 0 / flag set) when a vertex is not in front of the transmitter, when a single triangle
// would cover more than RTS_MASK_MAX_CELLS cells, for W = 1 and for beams wider than ~120 degrees.  Cells are never finer than
// RTS_MASK_MIN_CELL (tangent units = radians): the f32 direction of the pre-filter must stay inside the one-cell margin.
#define RTS_MASK_MIN_CELL 1.0e-5
#define RTS_MASK_N 1024u
#define RTS_MASK_WORDS (RTS_MASK_N * RTS_MASK_N / 32u + 1u)      // the mask lives behind the handle's zero block (RTS_ZERO_WORDS): ONE fill per pulse clears both
#define RTS_MASK_MAX_CELLS 4096
struct RtsMaskFrame { float bx, by, bz, ux, uy, uz, vx, vy, vz; float u0, v0, inv_du, inv_dv; uint32_t n, pad0, pad1; };

struct RtsLaunchConsts {
    double ox, oy, oz;              // d_rayOrigin
    double bsx, bsy, bsz;           // beamStart
    double stx, sty, stz;           // lattice step per launch index
    double rot[9];                  // Rot  (azimuth)
    double rot1[9];                 // Rot1 (elevation about the rotated y axis)
    double w1x, w1y, w1z;           // direction for W == 1
    uint64_t ray_first;
    uint32_t W, pad;
    uint32_t w_magic, w_more;       // division by W (>= 2) without a divide: q = mulhi(magic, g); ((g - q) >> 1) + q >> more  (fill_launch_constants)
    uint32_t il_tile, il_parts, il_part, pad2;     // interleaved tiles (il_parts <= 1: contiguous)
    const uint32_t* il_list;        // != nullptr: local tile j of il_tile launch indices is tile il_list[j] of the range (rts_set_tile_list: tiles DEALT to this launch, ascending) instead of j * il_parts + il_part
    RtsMaskFrame mask;              // primary-ray mask frame (n = 0: no mask this launch)
    // f32 constants of the primary-ray PRE-FILTER (rts_trace.hip): beamStart, lattice step, and Rot1 * Rot in ONE matrix -- the two
    // normalisations between them (ray_tracer.cu:170, 182) only scale, so the direction is Rot1 Rot (bs + st l) up to its length
    float f_bs[3], f_st[3], f_m[9], f_pad;
};
static_assert(sizeof(RtsLaunchConsts) % 8 == 0, "launch constants are copied to LDS dword by dword");

// lanes that share a ray in a unit of the cooperative kernel that walks the octant versions (rts_trace.hip: rts_walk_coop): 32 = two rays per wave (measured best: profiles/r05v_coop_group.log), 16 = four, 64 = one
#ifndef RTS_COOP_GROUP
#define RTS_COOP_GROUP 32
#endif
struct RtsTraceArgs {
    const RtsLaunchConsts* lc;      // device copy of the launch constants
    uint64_t ray_first;
    uint32_t n_rays, W;
    uint32_t max_refl, smooth;
    uint32_t n_prims, n_targets, n_rx, keep_all;
    uint32_t max_refr, rows;        // 0 or 2; output rows per launch index (1 or max_refl + 3)
    RtsChildState* child;           // [2][grid threads] (refraction only)
    // scene
    const RtsNode4* nodes4;
    const RtsNode4* nodes4v;        // != null: the eight OCTANT VERSIONS of every node record, [node][octant] (k_node_versions, rts_api.hip): the ordinary kernel walks these
    const RtsLeafTri* leaves;
    const uint32_t* tri_nidx;       // [n_prims][3] indices into normals
    const double* normals;          // world-space normals [.][3]
    const RtsTargetDev* targets;
    const RtsRxDev* rx;
    // outputs
    RtsEndRecord* recv_records;     // appended (unordered), capacity n_rays
    RtsEndRecord* all_records;      // [n_rays] (keep_all)
    unsigned long long* counters;   // [0] recv count (append atomic) [1] segments [2] shaded [3] node visits [4] tri tests [5] spills [6] hard overflow
    unsigned long long* block_counters;   // [grid][8] per-block partial sums of counters 1..6 (k_sum_counters)
    float* dir_hist;                // [max_refl][3][n_rays] reflected directions (f32)
    int32_t* hit_prim;              // [n_rays][max_refl+1] (keep_all)
    float* hit_t;                   // [n_rays][max_refl+1] (keep_all)
    int32_t* stack_ovf;             // [RTS_STACK_OVF][grid threads]
    uint32_t total_threads;         // threads of the ordinary kernel's grid
    uint32_t slab_threads;          // row length of the per-thread slabs (stack_ovf, child): total_threads + the cooperative kernel's threads
    const uint32_t* tile_order;     // [wave tiles] tile ids in descending order of the cost last seen by the handle (null: identity)
    uint32_t* tile_cost;            // [wave tiles] out: duration of the tile (units of 8/3 ticks of the 100 MHz counter = 64 shader clocks at 2.4 GHz, + 1)
    uint32_t* tile_ctr;             // the zero block: draw counters (element s * RTS_TILE_CTR_STRIDE; the cooperative kernel's at RTS_OFF_CTR_COOP), zero at launch
    uint32_t coop_spread;           // 1, 2, 4 or 8: XCD lists a head tile's 64 cooperative units are dealt to (rts_trace.hip)
    const uint32_t* xcd_seg;        // != null: XCD-affine sub-orders -- [RTS_XCD + 1] first position of each band's segment in tile_order (the last entry: its end)
    const uint32_t* tile_head;      // [1] number of tiles at the head of tile_order that are traced as 64 cooperative units (null: none)
    const uint32_t* tile_head_all;  // the same word whether or not this launch has a cooperative kernel (read back with the counters)
    const uint32_t* tile_live;      // != null: [1 + tiles at the front of tile_order that cost more than a dead tile last time] (0: unknown); the order behind them is drawn 64 tiles at a time, one LANE per tile (k_trace: dead-tile batches)
    uint32_t rx_window_screen;      // 1: the pre-filter also asks whether a crossing of a capture sphere can lie in the receiver's angular window (RTS_RX_WINDOW_SCREEN=0: only whether the sphere is reached)
    uint32_t coop_versions;         // 1: the cooperative kernel walks the octant versions too (product and counting builds of the plain chain; RTS_COOP_VERSIONS=0: the plain records and the sorted children)
    uint32_t batch_dead;            // 0: never batch; 1: batch the order's dead region (needs tile_live); 2: every position goes through the tile-level test first (RTS_DEAD_BATCH=all: tests)
    uint32_t* done_ctr; uint32_t n_blocks_all; unsigned long long* host_cnt;      // the last block of the launch (ticket from done_ctr, zero at launch) sums the block counters and writes them home
    uint32_t async_idle0, async_idle1, async_age;   // asynchronous bounces (rts_trace_unit_async): idle-lane limit of a walk phase for young / old tiles (0: lock-step kernel), age in cost units
    uint32_t coop_walk_steps_lo;    // ... bit 30 of the record: >= this many (LONGISH WALKS; the head rule asks more of such a tile's cost, rts_post.hip)
    uint32_t coop_min_cost, coop_walk_steps;   // a tile is flagged LONG WALKS (bit 31 of its cost record) if it took >= coop_min_cost units and >= coop_walk_steps walk iterations per bounce round
    unsigned long long* timeline;   // debug (RTS_TIMELINE, counting build): [grid][2] block start/end ticks, then [tiles] tile durations (100 MHz)
    uint32_t pre_filter;            // 1: primary rays go through the f32 pre-filter (needs the mask when there is geometry)
    const uint32_t* pmask;          // primary-ray mask: RTS_MASK_N^2 bits, then one word != 0 if the mask is void for this pulse (null: none)
    uint32_t stack_lds;             // LDS stack entries in use (RTS_STACK_LDS; smaller only to exercise the spill path in tests)
};

// ----------------------------------------------------------------------------- host context
struct RtsMeshHost {
    uint32_t n_tris, n_verts, n_normals;
    uint32_t tri_base, vert_base, normal_base;
    double refl_coeff, refr_index;
    bool perface;
};

// Pinned host staging (hipHostMalloc): every small per-pulse upload/readback goes through it, so the
// stream never has to be drained just to recycle a pageable temporary.
#define RTS_PIN_GROUPS 4096
struct RtsPinned {
    // [lc | motion | td]: the per-pulse parameters, uploaded with ONE copy into RtsContext::d_params (same layout): the launch
    // constants alone when no target moved, else up to the last target's placement
    RtsLaunchConsts lc;
    RtsTargetMotion motion[256];
    RtsTargetDev td[256];
    unsigned long long cnt[16];
    uint32_t G, pad;
    double rcs[256];
    double gsum[5 * RTS_PIN_GROUPS]; uint64_t gkey[RTS_PIN_GROUPS]; uint64_t grow[RTS_PIN_GROUPS]; uint32_t gmin[RTS_PIN_GROUPS];
};

template <typename T> struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        // hipFree is a device-wide synchronisation (it stalls the other handles' launches): grow geometrically, and give
        // the many small buffers sized by a pulse's received-ray count room to begin with
        size_t want = n + n / 8 + 16;
        if (want < 2 * cap) want = 2 * cap;
        if (sizeof(T) <= 144 && want < 65536) want = 65536;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Handles joined by rts_link_handles share one trace stream: their trace kernels execute one after the other, in the
// order the pulses were begun; the rest of each pulse (scene placement before, ordering / finalise /
// aggregation after) runs on the handle's own (high-priority) stream and overlaps with the other handles' trace kernels.
struct RtsGate { hipStream_t tstream = nullptr; std::atomic<int> refs{0}; int device = 0; };      // (handles may be driven from different threads)

// The immutable part of a scene -- meshes in their own frames, the static target-space hierarchy and its leaf order -- lives
// ONCE per device and is shared (reference counted) by every handle that was given it with rts_share_scene: handles that keep
// several pulses in flight place, trace and post-process into their own per-pulse buffers but read the same nodes.
struct RtsScene {
    std::atomic<int> refs{1}; int device = 0;
    std::vector<RtsMeshHost> meshes;
    uint32_t n_prims = 0, n_verts = 0, n_normals = 0;
    DevBuf<uint32_t> d_tri_vidx, d_tri_nidx, d_vert_targ, d_norm_targ, d_prim_targ;
    DevBuf<double> d_verts_local, d_normals_local;
    std::vector<RtsBlasInfo> blas; uint32_t n_nodes = 0, n_leaves = 0;
    DevBuf<RtsNode4> d_nodes4; DevBuf<uint32_t> d_leaf_prim;
    DevBuf<RtsNode4> d_nodes4v;     // [n_nodes][8] octant versions of d_nodes4 (empty: none -- RTS_NODE_VERSIONS=0, or the scene too large for them)
    double build_ms = 0; uint32_t builder = 0;     // how long the hierarchy build took, and where it ran (0 host SAH, 1 device LBVH)
    size_t device_bytes() const {
        return d_tri_vidx.cap * 4 + d_tri_nidx.cap * 4 + d_vert_targ.cap * 4 + d_norm_targ.cap * 4 + d_prim_targ.cap * 4 + d_verts_local.cap * 8 +
               d_normals_local.cap * 8 + (d_nodes4.cap + d_nodes4v.cap) * sizeof(RtsNode4) + d_leaf_prim.cap * 4;
    }
    void release() { d_tri_vidx.release(); d_tri_nidx.release(); d_vert_targ.release(); d_norm_targ.release(); d_prim_targ.release();
                     d_verts_local.release(); d_normals_local.release(); d_nodes4.release(); d_nodes4v.release(); d_leaf_prim.release(); }
};

// The tile-cost HISTORY (what every wave tile of the W^3 lattice cost the launch that traced it last: the cost order, the cooperative kernel's
// head and the dead part of the order are built from it, rts_post.hip) belongs to the handles that SHARE A SCENE on a device (round 5): they trace
// consecutive pulses of one interval, so what one of them measured is the best estimate the others have -- three handles in flight learn three times
// as fast, a handle's first launch finds the schedule its neighbours built, and a 20-pulse run is no longer mostly schedules settling.  Entries are
// single words replaced whole: a neighbour's merge racing with this handle's order build reads an old record or a new one, both valid estimates.
// (RTS_SHARE_HISTORY=0: every handle its own.)
struct RtsTileHist { std::atomic<int> refs{1}; DevBuf<uint32_t> d; uint32_t n = 0; bool any = false; uint32_t head_hint = 0; bool head_hint_valid = false; };

#define RTS_SMALL_CAP32 4096u         // received rays the one-block ordering kernels take with 32-bit sort keys (rts_post.hip) ...
#define RTS_SMALL_CAP64 2048u         // ... and with 64-bit keys; a speculatively enqueued post-processing chain is sized for the smaller of its two sorts
struct RtsSpecParams { std::vector<double> rcs; double wl = 0, gt = 0, gr = 0, carrier = 0, cspeed = 0; int32_t cube_pulse = -1; uint64_t base = 0;
                       int mode = 0; };     // 0: the uniform chain (rts_trace_pulse_end_uniform); 1: order + expand + the received set to the host mirror (rts_received_prefetch)
// Host mirror of a pulse's received set and of its aggregation outputs (rts_received_prefetch, the C++ adapter's path): ONE pinned
// allocation that KERNELS write (k_mirror_rows, k_mirror_agg: only the rows that exist cross the bus, no copy calls) and read
// (k_set_values: the power / Doppler the simulator's callbacks produced).  cap rows of: PerRayData | path row | RCS-angle row | slot |
// aggregated power, doppler, delay, phase, pathMatch | values in: power, doppler.
struct RtsHostMirror {
    char* host = nullptr; char* dev = nullptr; size_t bytes = 0; uint32_t cap = 0, D = 0;
    size_t o_rays = 0, o_paths = 0, o_angles = 0, o_slots = 0, o_apower = 0, o_adoppler = 0, o_adelay = 0, o_aphase = 0, o_apm = 0, o_vpower = 0, o_vdoppler = 0;
    bool recv_valid = false, agg_valid = false;      // the mirror holds the last pulse's received set / aggregation outputs (once the stream has drained)
    bool want = false;                               // this pulse's post-processing feeds the mirror (set by rts_received_prefetch, cleared by the next rts_trace_pulse_begin)
};
// rts_aggregate enqueues; the table is read (stream wait + pinned block -> RtsGroup records) by the first call that needs it
struct RtsAggPending { bool valid = false, wide = false, rows = false; uint32_t R = 0, D = 0, B = 0, shift = 0, spec = 0; uint64_t base = 0; double* gsum = nullptr; };

struct RtsContext {
    RtsParams params;
    uint32_t depth;                 // D = max_refr + max_refl
    int device;
    hipStream_t stream = nullptr;       // scene placement, ordering, finalise, aggregation (high priority: short kernels)
    hipStream_t tstream = nullptr;      // trace kernels (the link group's, see RtsGate)
    hipStream_t cstream = nullptr; hipEvent_t ev_coop[2];      // the cooperative trace kernel of a launch runs beside the ordinary one; stream created on first use (rts_trace.hip)
    uint32_t coop_grid_max = 1024;      // most blocks of the cooperative kernel (RTS_COOP_GRID)
    uint32_t coop_spread = 1;           // XCD lists a head tile's units are dealt to (RTS_COOP_SPREAD = 1 / 2 / 4 / 8)
    uint32_t n_head_hint = 0;           // head count of the handle's previous order build (came home with that launch's counters)
    uint32_t last_coop_grid = 0;        // blocks of the cooperative kernel in the handle's last launch (0: none was launched)
    hipEvent_t ev[9];
    // scene: the shared static part, and this handle's placement of it
    RtsScene* scene = nullptr;          // never null after rts_create
    DevBuf<double> d_verts_world, d_normals_world;
    std::vector<RtsTargetMotion> motion; bool motion_valid = false; bool bvh_valid = false;   // bvh_valid: scene placed for `motion`
    DevBuf<char> d_params;              // device image of RtsPinned's [lc | motion | td]
    RtsLaunchConsts* p_lc = nullptr; RtsTargetMotion* p_motion = nullptr; RtsTargetDev* p_targets = nullptr;     // ... and its parts
    // hierarchy: static nodes + leaf order (set_scene), leaf records refreshed per pulse
    uint32_t stack_lds = RTS_STACK_LDS;
    int grid_mult = 4, grid_spare = 160; bool grid_spare_forced = false; bool tile_lpt = true; double ew_rel = 1.7763568394002505e-15;   // per-handle knobs (rts_create reads RTS_GRID_MULT / RTS_GRID_SPARE / RTS_TILE_LPT / RTS_EW_REL)
    DevBuf<RtsLeafTri> d_leaves; DevBuf<char> d_sort_tmp;
    // receivers
    DevBuf<RtsRxDev> d_rx; uint32_t n_rx = 0;
    // per pulse
    uint64_t ray_first = 0; uint32_t n_rays = 0;
    DevBuf<RtsEndRecord> d_recv, d_all; DevBuf<unsigned long long> d_block_counters, d_timeline; unsigned long long* p_counters = nullptr;      // (the 16 counters live behind the draw counters: one fill zeroes both)
    DevBuf<uint32_t> d_tile_cost, d_tile_key, d_tile_key_sorted, d_tile_id, d_tile_order, d_tile_ctr;
    RtsTileHist* hist = nullptr; bool share_history = true;      // never null after rts_create; shared with the handles of the same scene (rts_share_scene) unless RTS_SHARE_HISTORY=0
    uint32_t coop_floor = 7500;         // ... more than this many cost units (shader clocks >> 6; 7 500 = 0.2 ms of one wave) (RTS_COOP_FLOOR)
    uint32_t coop_walk_steps_lo = 400; double coop_mid = 1.5;      // LONGISH WALKS (RTS_COOP_STEPS_LO) go to the head only if the tile cost more than coop_mid x the balanced time (RTS_COOP_MID; 0: never).  3 in round 4;
                                        // re-scanned in round 5, when the dead work had left the balanced time and the cooperative records had stopped flip-flopping (rts_record_to_keep): 1.5 takes a lone
                                        // BASELINE configs[3] launch from 7.4 to 4.75 ms and the pipelined pulse from 5.23 to 5.06, configs[2] / [4] / [1] do not move (profiles/r05c_coop_mid_scan.log)
    uint32_t coop_walk_steps = 1000;     // ... and whose bounce rounds took at least this many walk iterations each, on average (RTS_COOP_STEPS; 0: every tile above the floor is flagged) --
                                        // counted by the kernel, so neither other pulses sharing the GPU nor a launch that is all tail move it
    double coop_big_part = 1.5, coop_big_now = 0.0;      // coop_big for a launch that is a PART of a pulse (interleaved or dealt tiles) on a GPU no other pulse shares (RTS_COOP_BIG_PART; 0: off) -- one
                                        // pulse split over N GPUs for latency: such a launch is all tail, 1-2 ms tiles of moderately long walks bound it, and the ~5 x work of
                                        // cooperative units is free on an idle chip: one GPU's eighth of a BASELINE configs[3] pulse 1.1-2.3 -> 1.1-1.5 ms (profiles/r04_c4_deal*.log).  Whole pulses:
                                        // off (configs[4] one pulse at a time loses 3-5 % with it, the others do not move, profiles/r04_coop_big_inflight1.log); coop_big_now: the launch's value
    double coop_big = 0.0;              // ... or ANY tile costing more than this multiple of the balanced time, whatever its shape (RTS_COOP_BIG; 0 = off, the default:
                                        // measured on C3 at 0.8 / 1.0 / 1.3 -- the slowest tile of a launch is rarely the slowest of the previous one once the target moves,
                                        // the launch's duration did not change (0.70-0.77 ms, peaks of 1.0 ms as before) and the handle's first such launch takes 10 ms)
    uint32_t async_idle0 = 0, async_idle1 = 8, async_age = 7500;   // RTS_ASYNC_IDLE0 / _IDLE1 / _AGE (rts_trace_unit_async; idle0 = 0: the lock-step kernel)
    double coop_frac = 0.5;            // a tile costing more than this fraction of the launch's balanced time is traced as cooperative units (RTS_COOP_FRAC; 0: never)
    bool tile_cost_pending = false; uint64_t tile_cost_sig[4] = {0, 0, 0, 0};   // this handle's last launch left cost records that are not merged into the history yet (rts_post.hip)
    // tiles DEALT to this handle (rts_set_tile_list: ray sharding balanced by last-seen cost instead of interleaved parts): ascending tile
    // numbers in units of il_list_tile launch indices; a pulse with interleave_parts == RTS_INTERLEAVE_LIST traces them
    DevBuf<uint32_t> d_il_list; uint32_t il_list_n = 0, il_list_tile = 0, il_list_gen = 0; uint32_t il_list_last = 0;
    uint64_t tile_last_sig[4] = {0, 0, 0, 0}; bool tile_last_valid = false; DevBuf<uint32_t> d_rec_tmp;      // shape of the last launch that recorded costs (rts_tile_records_get)
    DevBuf<float> d_dir_hist; DevBuf<uint32_t> d_pmask; bool use_pmask = true, pre_dense = false;
    DevBuf<int32_t> d_hit_prim; DevBuf<float> d_hit_t; DevBuf<int32_t> d_stack_ovf; DevBuf<RtsChildState> d_child;
    DevBuf<uint64_t> d_rk64, d_rk64_sorted;
    RtsTraceArgs last_args; RtsLaunchConsts last_lc;
    // received set (ordered, expanded)
    uint64_t n_recv = 0;
    DevBuf<uint32_t> d_rk, d_rk_sorted, d_ri, d_ri_sorted;
    DevBuf<PerRayData> d_rx_rays; DevBuf<int32_t> d_rx_paths; DevBuf<double> d_rx_angles; DevBuf<uint64_t> d_rx_slots;
    DevBuf<PerRayData> d_all_rays; DevBuf<int32_t> d_all_paths; DevBuf<double> d_all_angles;
    // aggregation
    DevBuf<uint64_t> d_akeys, d_akeys_sorted; DevBuf<uint32_t> d_aidx, d_aidx_sorted; DevBuf<uint32_t> d_ghead, d_gid;
    DevBuf<double> d_gsum; DevBuf<uint32_t> d_gmin; DevBuf<uint64_t> d_gkey; DevBuf<uint32_t> d_gcount; DevBuf<uint64_t> d_grow; DevBuf<int32_t> d_gpath;
    DevBuf<double> d_delay, d_phase; DevBuf<int32_t> d_pathmatch; DevBuf<double> d_rcs;
    std::vector<RtsGroup> groups; bool agg_valid = false; uint64_t recv_index_base = 0;
    bool rx_window_screen = true;       // RTS_RX_WINDOW_SCREEN
    bool timeline_blocks = false; double tl_summary[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // RTS_TIMELINE_BLOCKS (rts_get_block_timeline)
    uint32_t tl_blocks = 0;             // debug (RTS_TIMELINE_BLOCKS): blocks whose start / end ticks this launch recorded
    bool coop_versions = true;          // RTS_COOP_VERSIONS
    int batch_dead = 1;                 // dead-tile batches of the trace kernel (RTS_DEAD_BATCH = 0 / 1 / all)
    bool node_versions = true;          // the ordinary trace kernel walks the octant versions of the node records when the scene has them (RTS_NODE_VERSIONS=0: the role fetch + sorting network)
    bool debug_coop = false;            // RTS_DEBUG_COOP: one line per launch on stderr (grids, head hint, thresholds)
    bool sum_in_kernel = false;         // the launch's last block sums the block counters (RTS_SUM_IN_KERNEL=1) instead of k_sum_counters -- measured SLOWER, off: the ticket's
                                        // release / acquire fences (one per block) write back and invalidate L2, and the post-processing behind the trace took 0.40 instead of 0.34 ms
    bool spin_wait = true;              // the pulse's two host waits poll the stream instead of blocking (rts_stream_wait; RTS_SPIN_WAIT=0)
    bool order_fused = true, order_sum_valid = false;        // the tile order in two launches when the previous launch had this launch's shape (RTS_ORDER_FUSED=0: four)
    bool place_fused = true, verts_world_valid = false;      // the per-pulse scene update in one launch (RTS_PLACE_FUSED=0: k_place + k_leaves); d_verts_world holds the current placement
    bool tile_bucket_order = true;      // tile order by counting bins instead of a radix sort (RTS_TILE_SORT=radix: the sort)
    int xcd_affine = 0; bool xcd_affine_now = false; uint32_t xcd_bnd_tiles = 0; DevBuf<uint32_t> d_xcd;      // XCD-affine sub-orders of the ORDINARY kernel (RTS_XCD_AFFINE = 0, the default / 1 / auto; rts_post.hip: rts_tile_order_build) -- measured slower, DESIGN.md section 5
    uint32_t post_prio = 3;             // s_setprio of k_post_all's waves (RTS_POST_PRIO = 0 .. 3)
    uint64_t post_one_max = 1024;       // ... when the handle's previous pulse received at most this many rays (RTS_POST_ONE_MAX); above it the seven launches are faster
    bool post_one = true;               // rts_trace_pulse_end_uniform: ONE kernel for order + expand + finalise + cube + aggregation of a small received set (RTS_POST_ONE=0: seven)
    bool post_small = true;             // received sets of up to 4096 rays are ordered / finished by single blocks (RTS_POST_SMALL=0: the general chain)
    hipStream_t tstream_now = nullptr; bool trace_own_stream = true;      // the stream this pulse's trace kernel went to (rts_trace_pulse_begin; RTS_TRACE_OWN_STREAM=0: always the trace stream)
    hipEvent_t ev_spec = nullptr; uint32_t spec_cap = RTS_SMALL_CAP64; bool spec_on_trace_stream = false;      // (RTS_SPEC_STREAM=trace: the speculative chain behind the trace kernel on ITS stream)
    RtsSpecParams spec; bool spec_pending = false, spec_enabled = true;      // rts_trace_pulse_end_uniform: parameters of the chain; a chain enqueued on the device-side count awaits its resolution (RTS_SPECULATE=0: never)
    const unsigned long long* recv_dev = nullptr;                           // != nullptr while such a chain is being enqueued: its kernels take the received count from here
    uint64_t recv_hint = 0; bool recv_hint_valid = false;                    // received rays of the handle's previous pulse
    RtsAggPending agg_pending;          // the group table of the last rts_aggregate is still on its way (rts_aggregate_fetch reads it)
    RtsHostMirror mirror;               // rts_received_prefetch / rts_received_view / rts_finalise_values / rts_aggregated_view
    std::vector<PerRayData> v_rays; std::vector<int32_t> v_paths; std::vector<double> v_angles, v_apower, v_adoppler, v_adelay, v_aphase; std::vector<uint64_t> v_slots; std::vector<int32_t> v_apm;      // the views' fallback storage (sets beyond the mirror's capacity)
    std::vector<PerRayData> v_agg_rays; unsigned v_recv_have = 0;      // rts_aggregated_view's own scratch (never the received view's storage); bits: which of v_rays / v_paths / v_angles / v_slots hold THIS pulse's set already
    RtsRxDev* pin_rx = nullptr; uint32_t pin_rx_cap = 0; std::vector<RtsRxDev> rx_host;      // receivers: last values set (an unchanged set is not uploaded again) and the pinned staging of the asynchronous upload
    RtsCubeParams cube_params; double* cube = nullptr; DevBuf<double> d_cube_own; bool cube_set = false;
    DevBuf<double> d_doppler_own; double* doppler = nullptr; uint32_t doppler_n = 0;       // slow-time transform of the cube (rts_cube_doppler)
    bool agg_delay_in = true;           // rts_aggregate_device: the delay / phase arrays carry initial sums (rs::kernel_wrapper's in-out arguments); false: they start at zero
    int64_t agg_base_local = 0;         // pathMatch value of received ray i after rts_aggregate = agg_base_local + i
    RtsPinned* pin = nullptr; RtsPinned* pin_dev = nullptr;      // pinned host staging and its address on the device: kernels write the small per-pulse read-backs (counters, group table) straight into it
    bool rcs_uploaded = false; DevBuf<double> d_rcsval; int n_cu = 0; bool stats_pending = false; bool agg_timed = false, fin_timed = false;
    RtsStats stats;
    double lap_s[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t lap_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // RTS_LAP=1: host time per section of rts_trace_pulse_begin
    RtsGate* gate = nullptr; bool pulse_open = false;   // gate: never null after rts_create
};

// implemented in the .hip units
int rts_sah_build(const double* verts, const uint32_t* tris, uint32_t n_tris, double split_budget, std::vector<RtsNode4>& nodes, std::vector<uint32_t>& leaf_prim, RtsBlasInfo& out);
int rts_lbvh_build_device(RtsContext* c, RtsScene* ns, const std::vector<uint32_t>& vidx, const std::vector<RtsMeshHost>& mh, double split_budget);
int rts_scene_place(RtsContext* c, const RtsLaunchConsts& lc, bool place, uint32_t* pmask);      // placement kernels (place) + the primary-ray mask (lc.mask.n != 0)
int rts_tile_costs_flush(RtsContext* c);      // merge the cost records of the last launch into the history now (before its launch shape goes away)
int rts_tile_records_masked(RtsContext* c, uint32_t* d_out, uint32_t n);      // the history's records of the tiles the last launch traced, 0 elsewhere
int rts_tile_order_build(RtsContext* c, const uint64_t* prev_sig, bool prev_valid, const uint64_t* cur_sig, uint32_t n_tiles_cur, uint32_t resident_waves);
int rts_trace_launch(RtsContext* c, const RtsTraceArgs& a, bool count_traversal, unsigned coop_grid);
void rts_trace_preload();
int rts_post_order_and_expand(RtsContext* c);
int rts_mirror_reserve(RtsContext* c, uint32_t rows);
int rts_post_mirror_received(RtsContext* c);       // the ordered, expanded received set -> the host mirror (kernel stores, count from c->recv_dev when set)
int rts_post_mirror_aggregated(RtsContext* c);     // per-ray aggregation outputs -> the host mirror
int rts_post_set_values(RtsContext* c, const double* power, const double* doppler);      // power / Doppler of the received rays <- device-readable arrays (the mirror's values-in area)
int rts_post_expand_all(RtsContext* c);
int rts_post_all_small(RtsContext* c, uint32_t cap, const RtsSpecParams& sp, bool want_groups);      // the whole uniform post-processing of a small received set in ONE one-block kernel (count from the device)
int rts_cube_accumulate_device(RtsContext* c, uint32_t pulse_index, double cspeed, double carrier);
int rts_cube_accumulate_paths_device(RtsContext* c, uint32_t pulse_index, int64_t base);
int rts_cube_doppler_device(RtsContext* c, uint32_t n_fft, double* out);
int rts_post_finalise(RtsContext* c, const double* rcs_host, double wl, double gt, double gr, double carrier, double cspeed);
hipError_t rts_stream_wait(RtsContext* c, hipStream_t st);
int rts_aggregate_fetch(RtsContext* c, std::vector<RtsGroup>* groups);      // second half of rts_aggregate_device when groups == &c->groups: no-op when nothing is pending
int rts_aggregate_device(RtsContext* c, int32_t max_path, int32_t max_rx, const int32_t* d_paths, uint64_t R, uint32_t D,
                         double cspeed, double carrier, uint64_t base, PerRayData* d_rays, double* d_delay,
                         double* d_phase, int32_t* d_pm, std::vector<RtsGroup>* groups, double* d_npath,
                         double* d_power_sum, double* d_doppler_sum, int32_t pm_init, const uint64_t* d_rows);
void rts_set_error(const char* fmt, ...);
// RTS_DEBUG_SYNC=1: synchronise after every stage and name it on stderr, so that a device
// fault is attributed to the kernel that caused it
int rts_debug_stage(RtsContext* c, const char* name);
#define RTS_STAGE(c, name) do { int rc_ = rts_debug_stage((c), (name)); if (rc_ != RTS_OK) return rc_; } while (0)

#define RTS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    rts_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return RTS_ERR_HIP; } } while (0)
