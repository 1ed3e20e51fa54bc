// rts_trace_pt.hip -- persistent-wave trace kernel with lane refill (maxRefr == 0, received rays only).
//
// Same per-ray arithmetic as k_trace (rts_trace.hip) -- the ray operations are shared through
// rts_ray_ops.h and rts_raygen.h -- but a different schedule.  In k_trace a lane owns one launch index
// until that ray has finished all its bounces, and the wave advances to its next 64 launch indices only
// when its slowest lane is done: in a radar beam most rays miss, so the waves that contain hits run with
// most lanes idle.  Here a lane is a worker:
//   * every lane is in one of three phases: TRAVERSING a segment, WAITING for service (its traversal is
//     finished), or IDLE (no ray);
//   * the wave runs short bursts of traversal steps; when enough lanes are waiting (or nobody traverses)
//     it executes ONE service pass for all of them together -- miss program + write-back for the rays that
//     left the scene, shading + next segment for the rays that hit, and new launch indices for the lanes
//     that became free, taken as consecutive indices from the wave's own strided sequence of 64-index
//     tiles (ballot / mbcnt compaction, no atomics);
//   * shading, miss and ray generation are f64-heavy blocks: batching them per service pass keeps them
//     from being executed once per straggler.
#include "rts_internal.h"
#include "rts_raygen.h"
#include "rts_ray_ops.h"

#define PT_IDLE 0
#define PT_TRAVERSING 1
#define PT_WAITING 2
#ifndef PT_SERVICE_LANES
#define PT_SERVICE_LANES 20        // run a service pass once this many lanes are waiting or idle-with-work-left
#endif
#ifndef PT_BURST
#define PT_BURST 6                 // traversal steps between two looks at the wave state
#endif

template <bool COUNT>
__global__ void __launch_bounds__(RTS_BLOCK) k_trace_pt(const RtsTraceArgs a)
{
    __shared__ int32_t s_stack[RTS_STACK_LDS * RTS_BLOCK];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t gtid = blockIdx.x * RTS_BLOCK + tid;
    const RtsLaunchConsts& lc = *a.lc;
    const dvec3 origin = mk3(lc.ox, lc.oy, lc.oz);
    unsigned long long n_seg = 0, n_shaded = 0, n_nodes = 0, n_tris = 0, n_spill = 0;
    bool hard_overflow = false;

    // ray supply of this wave: tiles wave, wave + n_waves, ... of 64 consecutive launch indices
    const uint32_t n_waves = a.total_threads >> 6;
    uint32_t tile = __builtin_amdgcn_readfirstlane(gtid >> 6), tile_pos = 0;
    const uint32_t n_tiles = (a.n_rays + 63u) >> 6;

    // per-lane ray + traversal state
    RtsRay r; r.dir = r.prev = r.first = mk3(0, 0, 0); r.rayLength = r.power = r.doppler = 0; r.reflDepth = 0; r.received = -1; r.end = false; r.path_lo = r.path_hi = 0;
    uint32_t slot = 0; bool primary = true;
    uint32_t phase = PT_IDLE; bool want_ray = true;
    float oNx = 0, oNy = 0, oNz = 0, oFx = 0, oFy = 0, oFz = 0, iNx = 0, iNy = 0, iNz = 0, iFx = 0, iFy = 0, iFz = 0;
    bool spx = true, spy = true, spz = true;
    float tmin = SCENE_EPS, t_prune = RTS_DEFAULT_TMAX, best_t = RTS_DEFAULT_TMAX;
    int best_leaf = -1; uint32_t best_prim = 0xffffffffu;
    int node = 0, sp = 0; uint32_t steps = 0;
    const int SENTINEL = 0x7fffffff;

    auto begin_segment = [&]() {       // rtTrace of the lane's current ray: set up the conservative f32 slab constants (see rts_trace.hip)
        n_seg++;
        tmin = primary ? SCENE_EPS : SCENE_EPS_R;
        best_t = RTS_DEFAULT_TMAX; best_leaf = -1; best_prim = 0xffffffffu; t_prune = RTS_DEFAULT_TMAX;
        const float Eo = fmaxf(fmaxf(fabsf((float)r.prev.x), fabsf((float)r.prev.y)), fabsf((float)r.prev.z)) * 3.0e-7f + 1.0e-30f;
        const float ivx = (float)(1.0 / r.dir.x), ivy = (float)(1.0 / r.dir.y), ivz = (float)(1.0 / r.dir.z);
        spx = !(ivx < 0.0f); spy = !(ivy < 0.0f); spz = !(ivz < 0.0f);
        oNx = (float)r.prev.x + (spx ? Eo : -Eo); oFx = (float)r.prev.x - (spx ? Eo : -Eo);
        oNy = (float)r.prev.y + (spy ? Eo : -Eo); oFy = (float)r.prev.y - (spy ? Eo : -Eo);
        oNz = (float)r.prev.z + (spz ? Eo : -Eo); oFz = (float)r.prev.z - (spz ? Eo : -Eo);
        iNx = ivx * 0.9999996f; iFx = ivx * 1.0000004f;
        iNy = ivy * 0.9999996f; iFy = ivy * 1.0000004f;
        iNz = ivz * 0.9999996f; iFz = ivz * 1.0000004f;
        sp = 0; steps = 0;
        node = (a.n_prims > 0) ? 0 : SENTINEL;
        phase = (node == SENTINEL) ? PT_WAITING : PT_TRAVERSING;
    };

    for (;;) {
        // ---------------------------------------------------------------- look at the wave
        const unsigned long long m_trav = __ballot(phase == PT_TRAVERSING);
        const unsigned long long m_wait = __ballot(phase == PT_WAITING);
        const bool supply_left = tile < n_tiles;                                   // wave-uniform
        const unsigned long long m_free = __ballot(phase == PT_IDLE && want_ray);
        const int n_service = __popcll(m_wait) + (supply_left ? __popcll(m_free) : 0);
        if (m_trav == 0 && n_service == 0) break;                                  // nothing traversing, nothing to serve, no rays left
        if (m_trav == 0 || n_service >= PT_SERVICE_LANES) {
            // ------------------------------------------------------------ service pass
            if (phase == PT_WAITING) {
                bool finished = true;
                if (best_leaf < 0) {
                    rts_miss_program(r, a.rx, a.n_rx, origin);                     // ray_tracer.cu:260-478
                } else if ((r.end == false) && (r.reflDepth < a.max_refl)) {       // closest_hit gate, normal_shader.cu:134
                    n_shaded++;
                    const RtsLeafTri L = a.leaves[best_leaf];
                    const RtsTargetDev T = a.targets[L.targ];
                    const fvec3 nd = rts_shade_reflect(r, a, L, T, best_t, tmin, primary, origin);
                    float* dh = a.dir_hist + (size_t)(r.reflDepth - 1) * 3 * a.n_rays;   // direction history (RCS angles of received rays)
                    dh[slot] = nd.x; dh[(size_t)a.n_rays + slot] = nd.y; dh[2*(size_t)a.n_rays + slot] = nd.z;
                    primary = false;
                    finished = false;
                }                                                                  // else: absorbed, payload untouched
                if (finished) {
                    if (r.received >= 0) {                                         // write-back (ray_tracer.cu:246-253)
                        RtsEndRecord e;
                        e.rayLength = r.rayLength; e.power = r.power; e.doppler = r.doppler;
                        e.prevx = r.prev.x; e.prevy = r.prev.y; e.prevz = r.prev.z;
                        e.firstx = r.first.x; e.firsty = r.first.y; e.firstz = r.first.z;
                        e.path_lo = r.path_lo; e.path_hi = r.path_hi; e.slot = slot; e.received = r.received; e.reflDepth = r.reflDepth; e.pad = 0;
                        const unsigned long long idx = atomicAdd(&a.counters[0], 1ULL);
                        a.recv_records[idx] = e;
                    }
                    phase = PT_IDLE; want_ray = true;
                }
            }
            // next segment of the rays that were shaded (all lanes that are still WAITING here have a live ray)
            if (phase == PT_WAITING) begin_segment();
            // refill: idle lanes take consecutive launch indices from the wave's tiles
            unsigned long long m_want = __ballot(phase == PT_IDLE && want_ray);
            while (m_want != 0 && tile < n_tiles) {                                // wave-uniform loop
                const uint32_t base = tile * 64u + tile_pos;
                const uint32_t avail = min(64u - tile_pos, a.n_rays - base);
                const uint32_t rank = (uint32_t)__popcll(m_want & ((1ULL << lane) - 1ULL));
                const bool mine = ((m_want >> lane) & 1ULL) && rank < avail;
                if (mine) {
                    slot = base + rank;
                    r.dir = rts_primary_dir(lc, slot);                             // ray_generation, ray_tracer.cu:144-224
                    r.prev = origin; r.first = mk3(0.0, 0.0, 0.0);
                    r.rayLength = 0; r.power = 0; r.doppler = 0; r.reflDepth = 0; r.received = -1; r.end = false; r.path_lo = 0; r.path_hi = 0;
                    primary = true;
                    begin_segment();
                }
                const uint32_t took = min((uint32_t)__popcll(m_want), avail);
                tile_pos += took;
                if (tile_pos >= 64u || base + took >= a.n_rays) { tile += n_waves; tile_pos = 0; }
                m_want = __ballot(phase == PT_IDLE && want_ray);
            }
            if (!(tile < n_tiles) && phase == PT_IDLE) want_ray = false;           // supply exhausted: this lane retires
            continue;
        }
        // ---------------------------------------------------------------- traversal burst
#pragma unroll 1
        for (int it = 0; it < PT_BURST; it++) {
            if (phase != PT_TRAVERSING) continue;
            if (++steps > (1u << 24)) { hard_overflow = true; node = SENTINEL; phase = PT_WAITING; continue; }   // malformed tree guard
            if (node >= 0) {
                const float4* np = reinterpret_cast<const float4*>(a.nodes + node);
                const float4 q0 = np[0], q1 = np[1], q2 = np[2];
                const int4 q3 = reinterpret_cast<const int4*>(np)[3];
                if (COUNT) n_nodes++;
                float tn0, tf0, tn1, tf1;
                tn0 = ((spx ? q0.x : q0.w) - oNx) * iNx; tf0 = ((spx ? q0.w : q0.x) - oFx) * iFx;
                tn0 = fmaxf(tn0, ((spy ? q0.y : q1.x) - oNy) * iNy); tf0 = fminf(tf0, ((spy ? q1.x : q0.y) - oFy) * iFy);
                tn0 = fmaxf(tn0, ((spz ? q0.z : q1.y) - oNz) * iNz); tf0 = fminf(tf0, ((spz ? q1.y : q0.z) - oFz) * iFz);
                tn1 = ((spx ? q1.z : q2.y) - oNx) * iNx; tf1 = ((spx ? q2.y : q1.z) - oFx) * iFx;
                tn1 = fmaxf(tn1, ((spy ? q1.w : q2.z) - oNy) * iNy); tf1 = fminf(tf1, ((spy ? q2.z : q1.w) - oFy) * iFy);
                tn1 = fmaxf(tn1, ((spz ? q2.x : q2.w) - oNz) * iNz); tf1 = fminf(tf1, ((spz ? q2.w : q2.x) - oFz) * iFz);
                const bool h0 = fmaxf(tn0, 0.0f) <= fminf(tf0, t_prune);
                const bool h1 = fmaxf(tn1, 0.0f) <= fminf(tf1, t_prune);
                if (h0 && h1) {
                    const bool swap = tn1 < tn0;
                    const int nearc = swap ? q3.y : q3.x, farc = swap ? q3.x : q3.y;
                    if (sp < RTS_STACK_LDS) s_stack[sp * RTS_BLOCK + tid] = farc;
                    else if (sp < RTS_STACK_LDS + RTS_STACK_OVF) { a.stack_ovf[(size_t)(sp - RTS_STACK_LDS) * a.total_threads + gtid] = farc; n_spill++; }
                    else hard_overflow = true;
                    if (sp < RTS_STACK_LDS + RTS_STACK_OVF) sp++;
                    node = nearc;
                } else if (h0) node = q3.x;
                else if (h1) node = q3.y;
                else {
                    if (sp == 0) node = SENTINEL;
                    else { sp--; node = (sp < RTS_STACK_LDS) ? s_stack[sp * RTS_BLOCK + tid] : a.stack_ovf[(size_t)(sp - RTS_STACK_LDS) * a.total_threads + gtid]; }
                }
            } else {
                const int leaf = ~node;
                const RtsLeafTri L = a.leaves[leaf];
                if (COUNT) n_tris++;
                const TriHit h = tri_test(L, r.prev, r.dir, tmin, RTS_DEFAULT_TMAX);
                if (h.ok) {
                    const float tf = (float)h.t;                                   // rtPotentialIntersection takes float, triangle_mesh.cu:167
                    if ((tf > tmin) && (tf < best_t || (tf == best_t && L.prim < best_prim))) {
                        best_t = tf; best_leaf = leaf; best_prim = L.prim;
                        t_prune = f32_next_up_pos(tf);
                    }
                }
                if (sp == 0) node = SENTINEL;
                else { sp--; node = (sp < RTS_STACK_LDS) ? s_stack[sp * RTS_BLOCK + tid] : a.stack_ovf[(size_t)(sp - RTS_STACK_LDS) * a.total_threads + gtid]; }
            }
            if (node == SENTINEL) phase = PT_WAITING;
        }
    }

    // ------------------------------------------------------------------ counters: wave reduce, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        n_seg += __shfl_down(n_seg, off); n_shaded += __shfl_down(n_shaded, off);
        if (COUNT) { n_nodes += __shfl_down(n_nodes, off); n_tris += __shfl_down(n_tris, off); }
        n_spill += __shfl_down(n_spill, off);
    }
    if (lane == 0) {
        atomicAdd(&a.counters[1], n_seg); atomicAdd(&a.counters[2], n_shaded);
        if (COUNT) { atomicAdd(&a.counters[3], n_nodes); atomicAdd(&a.counters[4], n_tris); }
        if (n_spill) atomicAdd(&a.counters[5], n_spill);
    }
    if (hard_overflow) atomicAdd(&a.counters[6], 1ULL);
}

int rts_trace_launch_pt(RtsContext* c, const RtsTraceArgs& a, bool count_traversal)
{
    if (a.n_rays == 0) return RTS_OK;
    const unsigned grid = a.total_threads / RTS_BLOCK;
    if (count_traversal) k_trace_pt<true><<<grid, RTS_BLOCK, 0, c->tstream>>>(a);
    else k_trace_pt<false><<<grid, RTS_BLOCK, 0, c->tstream>>>(a);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}
