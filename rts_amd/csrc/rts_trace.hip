// rts_trace.hip -- the hot kernel: ray generation, BVH4 traversal (target space) with the f64 triangle
// test, reflection shading and receiver capture, for one launch (one pulse).
//
// Replaces the OptiX programs of the reference:
//   ray_generation  ray_tracer.cu:144-255      miss         ray_tracer.cu:260-478
//   intersect       triangle_mesh.cu:121-200   closest_hit  normal_shader.cu:128-340
// and OptiX's closed-source "Bvh" traverser (ray_tracer.cpp:1127-1128).
//
// Mixed-precision contract (reference quirks 1-3):
//   - the triangle test runs in f64 on the payload's origin/direction (prevHitPoint,
//     rayDirection), t is compared in f64 against the f32 constants and then narrowed to f32;
//   - the hit point is prev + (double)t_f32 * dir, rayLength += t_f32;
//   - the bounce direction is optixu reflect() in f32 on the f32 ray direction, widened
//     to f64 WITHOUT renormalisation.
// Closest hit = smallest f32 t; equal f32 t resolved to the lowest global primitive id
// (OptiX keeps whichever it met first; order there is unknowable).
//
// Structure: one lane per launch index, persistent waves drawing 64-index tiles from a queue; each lane iterates its
// bounces (the reference recurses through rtTrace).  The traversal stack lives in LDS
// (entry-major, lane-minor: conflict free), spilling to a global slab when deeper.
#include "rts_internal.h"
#include "rts_raygen.h"
#include "rts_ray_ops.h"

// Conservative f32 slab test of a ray against padded f32 boxes (the traversal only has to be CONSERVATIVE with respect
// to the f64 triangle test; the boxes are padded f32 already, rts_sah.cpp).  Per axis the ray parameter at a box plane P is
//     t(P) = (P - o) / d  =  fma(P, i, c),   i = 1/d,  c = -o i
// -- ONE instruction per plane, with per-ray constants (i, c) that already carry the whole error budget, chosen by the role
// the plane plays for this ray (d > 0: the box's low plane is the entry plane, its high plane the exit plane; d < 0 the
// other way round): the record fetch brings each lane ITS entry and exit planes (RtsSlabRay::sg below), the per-node code
// evaluates near = fma(N, iN, cN), far = fma(F, iF, cF) per axis and takes max3 / min3 over the axes.  Error budget (f32, eps = 2^-24):
//   o32 = fl32(o): |o - o32| <= eps |o|; c = fl(o' i') with o' = o32 -+ E: its rounding, eps |o' i'|, is a shift of the
//   origin by eps |o'|; the f64 hit point that has to lie inside the box is off the ideal target-space point by `extra`
//   (world-scale rounding of the placed vertices, RtsTargetDev::ew).  E = 4.5e-7 max|o| + extra covers all three (2.7 eps
//   would do) and the rounding of o32 -+ E itself.
//   The roundings of fl32(d), of 1/d, of i' = i (1 -+ 6e-7) and of the fma result are <= 4 eps = 2.4e-7 relative to t: the
//   entry reciprocal is scaled by (1 - 6e-7), the exit one by (1 + 6e-7).  (Where that shrinks a NEGATIVE entry value
//   towards zero the clamp max(., 0) of the caller makes it irrelevant; an exit value that comes out negative belongs to
//   a box behind the origin, E included.)
//   d = 0 (or |1/d| beyond 1e30, where o i could overflow): the axis is dropped (entry -inf, exit +inf) -- conservative.
// Constants by ROLE, not by plane (round 3): N = the plane the ray enters the slab through (the box's LOW plane when d > 0, its
// HIGH plane when d < 0), F = the plane it leaves through.  The SIGN of iN? says which: the N plane of an axis sits 0 or 48 bytes
// behind the node record's low plane of that axis (hi? = lo? + 48, RtsNode4; a dropped axis has iN = +0) -- the record fetch adds
// that to the address of the lane's N-plane load and 48 minus that to the F-plane load's, so the registers come back holding near / far planes and the node
// code needs neither sign selects nor the min / max pair per axis that sorted the two parameters (24 of the node step's 41
// min / max instructions, each twice the issue cost of an fma: profiles/r03_valu_calib.json).  Same values as before wherever
// entry <= exit; an axis whose parameters come out crossed (a flat box far behind the origin) now reads as the empty
// interval it is -- a miss either way (exit < 0).
struct RtsSlabRay { float iNx, cNx, iFx, cFx, iNy, cNy, iFy, cFy, iNz, cNz, iFz, cFz; };
__device__ __forceinline__ void rts_slab_axis(float o, float d, float E, float& iN_, float& cN_, float& iF_, float& cF_)
{
    const float iv = 1.0f / d;
    const bool pos = !(iv < 0.0f);
    const float oN = o + (pos ? E : -E), oF = o - (pos ? E : -E);          // entry origin pushed towards the entry plane, exit origin away
    const float iN = iv * 0.9999994f, iF = iv * 1.0000006f;
    const float cN = -(oN * iN), cF = -(oF * iF);
    const bool drop = !(fabsf(iv) < 1.0e30f);                               // d == 0, denormal d, NaN
    const float NINF = -__builtin_inff(), PINF = __builtin_inff();
    iN_ = drop ? 0.0f : iN; cN_ = drop ? NINF : cN;
    iF_ = drop ? 0.0f : iF; cF_ = drop ? PINF : cF;
}
__device__ __forceinline__ RtsSlabRay rts_slab_setup(const dvec3& o, const dvec3& d, float extra)
{
    RtsSlabRay r;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    const float E = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) * 4.5e-7f + extra + 1.0e-30f;
    rts_slab_axis(ox, (float)d.x, E, r.iNx, r.cNx, r.iFx, r.cFx);
    rts_slab_axis(oy, (float)d.y, E, r.iNy, r.cNy, r.iFy, r.cFy);
    rts_slab_axis(oz, (float)d.z, E, r.iNz, r.cNz, r.iFz, r.cFz);
    return r;
}

// The record fetch of a traversal step as explicit instructions.  Written as plain loads ahead of the node / leaf branches
// the compiler sinks them into the branches again -- narrowed to the dwords each branch uses (17 global_load_dword) and, being
// in an if / else, issued one branch after the other: two dependent memory round trips per step.  Here: five dwordx4 loads
// from one per-lane base for every lane, two more for the lanes at nodes (lanes at leaves need 80 bytes, lanes at nodes 112;
// EXEC is narrowed to the latter inside the block), then the wait -- ONE asm statement:
// the backend does not track vector-memory loads issued from inline asm and the hardware has no interlock on a VGPR with a
// load in flight, so nothing the compiler might place (a copy, a spill of q0..q6) may come between the loads and the wait.
typedef unsigned int rts_u32x4 __attribute__((ext_vector_type(4)));
// byte address of an LDS word for ds_* instructions written by hand
typedef __attribute__((address_space(3))) int32_t rts_lds_i32;
__device__ __forceinline__ uint32_t rts_lds_addr(int32_t* p) { return (uint32_t)(uintptr_t)(rts_lds_i32*)p; }
// Loads: q0..q2 <- N planes x, y, z (record + 0 / 16 / 32 + offset), q3..q5 <- F planes (record + 0 / 16 / 32 + 48 - offset),
// q6 <- the child ids (record + 96); q5, q6 for the lanes at nodes only.  A lane at a leaf reads its record straight through
// (offsets 0: nm = 0).
__device__ __forceinline__ void rts_fetch_record(const void* p, int node, uint32_t nm, float iNx, float iNy, float iNz, rts_u32x4& q0, rts_u32x4& q1, rts_u32x4& q2, rts_u32x4& q3,
                                                 rts_u32x4& q4, rts_u32x4& q5, rts_u32x4& q6)
{
    // nm = 48 for a lane at a node, 0 at a leaf.  Per axis: off = (sign of iN ? 48 : 0) & nm; N-plane address = record + off,
    // F-plane address = record + (off ^ 48) -- formed INSIDE the block, one axis after the other, in one 64-bit temporary (a
    // load's address registers are read when it issues; its successor may overwrite them): as six 64-bit operands the
    // addresses cost the kernel twelve registers at its tightest point, and spills in the loops around the walk.
    // (the lanes at nodes -- node >= 0 -- are found and EXEC is parked in VCC, which the allocator never hands out as a
    // general pair: the kernel has no scalar register to spare; the carry-out of the address additions goes to VCC as well,
    // before that)
    unsigned long long t; uint32_t o;
    asm volatile("v_ashrrev_i32 %8, 31, %12\n\t"
                 "v_and_b32 %8, %8, %11\n\t"
                 "v_mad_u64_u32 %7, vcc, %8, 1, %9\n\t"
                 "global_load_dwordx4 %0, %7, off\n\t"
                 "v_xor_b32 %8, 48, %8\n\t"
                 "v_mad_u64_u32 %7, vcc, %8, 1, %9\n\t"
                 "global_load_dwordx4 %3, %7, off\n\t"
                 "v_ashrrev_i32 %8, 31, %13\n\t"
                 "v_and_b32 %8, %8, %11\n\t"
                 "v_mad_u64_u32 %7, vcc, %8, 1, %9\n\t"
                 "global_load_dwordx4 %1, %7, off offset:16\n\t"
                 "v_xor_b32 %8, 48, %8\n\t"
                 "v_mad_u64_u32 %7, vcc, %8, 1, %9\n\t"
                 "global_load_dwordx4 %4, %7, off offset:16\n\t"
                 "v_ashrrev_i32 %8, 31, %14\n\t"
                 "v_and_b32 %8, %8, %11\n\t"
                 "v_mad_u64_u32 %7, vcc, %8, 1, %9\n\t"
                 "global_load_dwordx4 %2, %7, off offset:32\n\t"
                 "v_xor_b32 %8, 48, %8\n\t"
                 "v_mad_u64_u32 %7, vcc, %8, 1, %9\n\t"
                 "v_cmp_lt_i32 vcc, -1, %10\n\t"
                 "s_and_saveexec_b64 vcc, vcc\n\t"
                 "global_load_dwordx4 %5, %7, off offset:32\n\t"
                 "global_load_dwordx4 %6, %9, off offset:96\n\t"
                 "s_mov_b64 exec, vcc\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(t), "=&v"(o)     // (q5, q6: written for the lanes at nodes only, read by them only)
                 : "v"(p), "v"(node), "v"(nm), "v"(iNx), "v"(iNy), "v"(iNz) : "memory", "scc", "vcc");
}

// The record read straight through (offsets 0 .. 96): the cooperative kernel's fetch.  Its waves are few and wait for memory,
// not for issue slots; the address arithmetic between the loads of the fetch above spreads them over ~15 instructions, and on
// BASELINE configs[3] -- 440 MB of records streaming through the L1 -- lines were evicted between the first and the last load
// of a record: +15 % L2 requests, the cooperative kernel 2.5 -> 2.95 ms (profiles/r03c_c4_ab.log).
__device__ __forceinline__ void rts_fetch_record_plain(const void* p, int node, rts_u32x4& q0, rts_u32x4& q1, rts_u32x4& q2, rts_u32x4& q3,
                                                       rts_u32x4& q4, rts_u32x4& q5, rts_u32x4& q6)
{
    asm volatile("global_load_dwordx4 %0, %7, off\n\t"
                 "global_load_dwordx4 %1, %7, off offset:16\n\t"
                 "global_load_dwordx4 %2, %7, off offset:32\n\t"
                 "global_load_dwordx4 %3, %7, off offset:48\n\t"
                 "global_load_dwordx4 %4, %7, off offset:64\n\t"
                 "v_cmp_lt_i32 vcc, -1, %8\n\t"
                 "s_and_saveexec_b64 vcc, vcc\n\t"
                 "global_load_dwordx4 %5, %7, off offset:80\n\t"
                 "global_load_dwordx4 %6, %7, off offset:96\n\t"
                 "s_mov_b64 exec, vcc\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6)
                 : "v"(p), "v"(node) : "memory", "scc", "vcc");
}
// the constants of a ray by PLANE for that fetch: N* <- the low plane's pair, F* <- the high plane's
__device__ __forceinline__ RtsSlabRay rts_slab_by_plane(const RtsSlabRay& r)
{
    RtsSlabRay p = r;
    if ((int32_t)__float_as_uint(r.iNx) < 0) { p.iNx = r.iFx; p.cNx = r.cFx; p.iFx = r.iNx; p.cFx = r.cNx; }
    if ((int32_t)__float_as_uint(r.iNy) < 0) { p.iNy = r.iFy; p.cNy = r.cFy; p.iFy = r.iNy; p.cFy = r.cNy; }
    if ((int32_t)__float_as_uint(r.iNz) < 0) { p.iNz = r.iFz; p.cNz = r.cFz; p.iFz = r.iNz; p.cFz = r.cNz; }
    return p;
}

// The stack entry below the top when it may live in the global spill slab (rare; kept out of line so that the common LDS
// read stays a ds_read).
__device__ __attribute__((noinline)) int rts_stack_below_spilled(int from_lds, int sp, int lds_cap, const int32_t* ovf, uint32_t total_threads, uint32_t gtid)
{
    return (sp - 1 < lds_cap) ? from_lds : ovf[(size_t)(sp - 1 - lds_cap) * total_threads + gtid];
}

// One step of the walk for the lanes that hold a node or a leaf (`node` != the sentinel): fetch the record, test the four
// child boxes / the triangle, update the stack and the closest hit.
#define RTS_STACK_SENTINEL 0x7fffffff
#define RTS_SEG_ONE 0x00400001u      // one traced segment in the per-lane LDS counter: launch total (bits 0-21) and current tile (bits 22-31)
// MODE (how a node record reaches the lane, and in which order its children are visited):
//   RTS_WALK_ROLES     the role fetch above (entry / exit planes by per-axis address offsets), children sorted by entry distance;
//   RTS_WALK_PLANES    the record read straight through (low / high planes; lr by plane, rts_slab_by_plane), children sorted -- the cooperative kernel's;
//   RTS_WALK_VERSIONS  the record of the ray's OCTANT (a.nodes4v: eight versions per node, written once per scene by k_node_versions,
//                      rts_api.hip) read straight through: its planes are already entry / exit planes for every ray of that octant and its
//                      children are already in front-to-back order along the octant's diagonal -- no address arithmetic per axis, no
//                      sorting network (5 compares + 20 selects of the node step), no +inf selects.  The ORDER only decides what gets
//                      pruned, never the result: every child whose slab interval is open is visited in either scheme, and the closest hit
//                      is the minimum over (f32 t, primitive id) whatever the order.
#define RTS_WALK_ROLES 0
#define RTS_WALK_PLANES 1
#define RTS_WALK_VERSIONS 2
template <bool COUNT, int MODE = RTS_WALK_ROLES>
__device__ __forceinline__ void rts_walk_step(const RtsTraceArgs& a, int32_t* s_stack, uint32_t tid, uint32_t gtid, int lds_cap, uint32_t* n_spill_lds,
                                              int& node, int& sp, const RtsSlabRay& lr, const dvec3& prev, const dvec3& dir, float tmin,
                                              float& best_t, int& best_leaf, uint32_t& best_prim, float& t_prune,
                                              uint32_t& n_nodes, uint32_t& n_tris, bool& hard_overflow)
{
    constexpr bool ROLES = MODE != RTS_WALK_PLANES;      // lr holds the constants by role (entry / exit)
    constexpr bool VERS = MODE == RTS_WALK_VERSIONS;
    const bool deep = __any(sp + 4 > lds_cap);                  // wave-uniform: some lane is about to leave the LDS part
    int below = s_stack[min(sp - 1, lds_cap - 1) * RTS_BLOCK + tid];      // (always an LDS read: a second, global source here made the compiler fold both into one FLAT load)
    if (deep) below = rts_stack_below_spilled(below, sp, lds_cap, a.stack_ovf, a.slab_threads, gtid);
    // One fetch for both kinds of step: a lane at a node needs its 112-byte record (six planes + child ids),
    // a lane at a leaf its 80-byte record -- the same five (seven) dwordx4 loads from a per-lane base,
    // issued together at the top of the step, so a wave whose lanes are at nodes AND at leaves waits for
    // ONE memory round trip (as if / else bodies the leaf loads could only be issued after the node body).
    const bool at_node = node >= 0;
    // (VERS: a node id IS the index of its record in a.nodes4v -- 8 node + octant; the children a version names carry that version's octant)
    const void* rp = at_node ? static_cast<const void*>(VERS ? a.nodes4v + node : a.nodes4 + node) : static_cast<const void*>(a.leaves + ~node);
    rts_u32x4 q0, q1, q2, q3, q4, q5, q6;
    if (MODE == RTS_WALK_ROLES) rts_fetch_record(rp, node, at_node ? 48u : 0u, lr.iNx, lr.iNy, lr.iNz, q0, q1, q2, q3, q4, q5, q6);
    else rts_fetch_record_plain(rp, node, q0, q1, q2, q3, q4, q5, q6);
    if (at_node) {
        // BVH4 node: six dwordx4 planes (lo/hi x,y,z of the four children) + the four child ids
        // (by value through __uint_as_float: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0)
#define RTS_F4(q) make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w))
        const float4 NX = RTS_F4(q0), NY = RTS_F4(q1), NZ = RTS_F4(q2), FX = RTS_F4(q3), FY = RTS_F4(q4), FZ = RTS_F4(q5);     // near / far planes of the four children
#undef RTS_F4
        const int4 CH = make_int4((int)q6.x, (int)q6.y, (int)q6.z, (int)q6.w);
        if (COUNT) n_nodes++;
        const float INF = __builtin_inff();
#define RTS_CHILD(k, dk) float dk, tn##dk, tf##dk; { \
            const float nx = __builtin_fmaf(NX.k, lr.iNx, lr.cNx), fx = __builtin_fmaf(FX.k, lr.iFx, lr.cFx); \
            const float ny = __builtin_fmaf(NY.k, lr.iNy, lr.cNy), fy = __builtin_fmaf(FY.k, lr.iFy, lr.cFy); \
            const float nz = __builtin_fmaf(NZ.k, lr.iNz, lr.cNz), fz = __builtin_fmaf(FZ.k, lr.iFz, lr.cFz); \
            const float tn = ROLES ? fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f) : fmaxf(fmaxf(fminf(nx, fx), fminf(ny, fy)), fmaxf(fminf(nz, fz), 0.0f)); \
            const float tf = ROLES ? fminf(fminf(fminf(fx, fy), fz), t_prune) : fminf(fminf(fmaxf(nx, fx), fmaxf(ny, fy)), fminf(fmaxf(nz, fz), t_prune)); \
            tn##dk = tn; tf##dk = tf; dk = VERS ? 0.0f : ((tn <= tf) ? tn : INF); }
        RTS_CHILD(x, d0) RTS_CHILD(y, d1) RTS_CHILD(z, d2) RTS_CHILD(w, d3)
#undef RTS_CHILD
        int c0 = CH.x, c1 = CH.y, c2 = CH.z, c3 = CH.w;
        if (VERS) {
            // children 0 .. 3 are in visiting order already: push the open ones last-first, go on with the first open one (which
            // need not be child 0: the open children are no prefix of an order that was fixed before the ray was known)
            if (!deep) {
                // per child: store it at the top, then  open = tn <= tf (VCC);  top = open ? child : top;  sp += open (carry in) -- four
                // vector instructions and one LDS store, the compare's mask never leaves VCC (as C++ the compiler forms all four masks
                // first: eight more scalar registers live through the step, and 0 / 1 selects + adds for the stack pointer).  Child 0:
                // node = open ? child 0 : top;  sp += open - 1 (closed: the top of the stack -- the entry just pushed, or `below`).
                // (the LDS stores of one wave complete in order: the next step's read of `below` sees them)
                uint32_t ad_; int top_, nd_;
                asm volatile("v_lshl_add_u32 %[ad], %[sp], 10, %[base]\n\t"
                             "ds_write_b32 %[ad], %[c3]\n\t"
                             "v_cmp_le_f32 vcc, %[n3], %[f3]\n\t"
                             "v_cndmask_b32 %[top], %[below], %[c3], vcc\n\t"
                             "v_addc_co_u32 %[sp], vcc, 0, %[sp], vcc\n\t"
                             "v_lshl_add_u32 %[ad], %[sp], 10, %[base]\n\t"
                             "ds_write_b32 %[ad], %[c2]\n\t"
                             "v_cmp_le_f32 vcc, %[n2], %[f2]\n\t"
                             "v_cndmask_b32 %[top], %[top], %[c2], vcc\n\t"
                             "v_addc_co_u32 %[sp], vcc, 0, %[sp], vcc\n\t"
                             "v_lshl_add_u32 %[ad], %[sp], 10, %[base]\n\t"
                             "ds_write_b32 %[ad], %[c1]\n\t"
                             "v_cmp_le_f32 vcc, %[n1], %[f1]\n\t"
                             "v_cndmask_b32 %[top], %[top], %[c1], vcc\n\t"
                             "v_addc_co_u32 %[sp], vcc, 0, %[sp], vcc\n\t"
                             "v_cmp_le_f32 vcc, %[n0], %[f0]\n\t"
                             "v_cndmask_b32 %[nd], %[top], %[c0], vcc\n\t"
                             "v_addc_co_u32 %[sp], vcc, -1, %[sp], vcc"
                             : [sp] "+v"(sp), [top] "=&v"(top_), [ad] "=&v"(ad_), [nd] "=v"(nd_)
                             : [base] "v"(rts_lds_addr(s_stack + tid)), [below] "v"(below), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3),
                               [n0] "v"(tnd0), [f0] "v"(tfd0), [n1] "v"(tnd1), [f1] "v"(tfd1), [n2] "v"(tnd2), [f2] "v"(tfd2), [n3] "v"(tnd3), [f3] "v"(tfd3)
                             : "vcc", "memory");
                node = nd_;
            } else {
                // (the compares at their uses: formed ahead of the branch, the four masks were eight scalar registers live through the step)
#define RTS_PUSH(cv) { if (sp < lds_cap) s_stack[sp * RTS_BLOCK + tid] = (cv); \
                       else if (sp < lds_cap + RTS_STACK_OVF) { a.stack_ovf[(size_t)(sp - lds_cap) * a.slab_threads + gtid] = (cv); atomicAdd(n_spill_lds, 1u); } \
                       else hard_overflow = true; \
                       if (sp < lds_cap + RTS_STACK_OVF) sp++; }
                int top = below;
                if (tnd3 <= tfd3) { RTS_PUSH(c3) top = c3; }
                if (tnd2 <= tfd2) { RTS_PUSH(c2) top = c2; }
                if (tnd1 <= tfd1) { RTS_PUSH(c1) top = c1; }
#undef RTS_PUSH
                if (tnd0 <= tfd0) node = c0;
                else { node = top; sp--; }
            }
            return;
        }
        // sort the four (distance, child) pairs ascending (5 compare-exchanges); misses carry +inf
        // (measured, not kept: the compare into an SGPR pair and four VOP3 selects written by hand, because a probe --
        // tools/cndmask_probe.hip -- shows v_cndmask_b32 reading VCC at ~22 cycles each when several follow one compare, against
        // 4.2 with an SGPR-pair mask.  Bit-identical, and no change at all in the kernel: C3 0.757 / 0.764 ms, control 2.80 / 2.83 ms.)
#define RTS_CSWAP(da, ca, db, cb) { const bool sw = db < da; const float td = sw ? db : da; const int tc = sw ? cb : ca; db = sw ? da : db; cb = sw ? ca : cb; da = td; ca = tc; }
        RTS_CSWAP(d0, c0, d1, c1) RTS_CSWAP(d2, c2, d3, c3) RTS_CSWAP(d0, c0, d2, c2) RTS_CSWAP(d1, c1, d3, c3) RTS_CSWAP(d1, c1, d2, c2)
#undef RTS_CSWAP
        // continue with the nearest, push the others farthest first
        if (!deep) {
            s_stack[sp * RTS_BLOCK + tid] = c3; sp += (d3 < INF) ? 1 : 0;
            s_stack[sp * RTS_BLOCK + tid] = c2; sp += (d2 < INF) ? 1 : 0;
            s_stack[sp * RTS_BLOCK + tid] = c1; sp += (d1 < INF) ? 1 : 0;
            const bool go = d0 < INF;                         // nothing hit: nothing was pushed either, `below` is the top
            node = go ? c0 : below; sp -= go ? 0 : 1;
        } else {
#define RTS_PUSH(cv) { if (sp < lds_cap) s_stack[sp * RTS_BLOCK + tid] = (cv); \
                       else if (sp < lds_cap + RTS_STACK_OVF) { a.stack_ovf[(size_t)(sp - lds_cap) * a.slab_threads + gtid] = (cv); atomicAdd(n_spill_lds, 1u); } \
                       else hard_overflow = true; \
                       if (sp < lds_cap + RTS_STACK_OVF) sp++; }
            if (d3 < INF) RTS_PUSH(c3)
            if (d2 < INF) RTS_PUSH(c2)
            if (d1 < INF) RTS_PUSH(c1)
#undef RTS_PUSH
            if (d0 < INF) node = c0;
            else { node = below; sp--; }
        }
    } else {
        const int leaf = ~node;
        RtsLeafTri L;
#define RTS_F64(lo, hi) __hiloint2double((int)(hi), (int)(lo))
        L.p0x = RTS_F64(q0.x, q0.y); L.p0y = RTS_F64(q0.z, q0.w); L.p0z = RTS_F64(q1.x, q1.y); L.p1x = RTS_F64(q1.z, q1.w);
        L.p1y = RTS_F64(q2.x, q2.y); L.p1z = RTS_F64(q2.z, q2.w); L.p2x = RTS_F64(q3.x, q3.y); L.p2y = RTS_F64(q3.z, q3.w);
        L.p2z = RTS_F64(q4.x, q4.y); L.prim = q4.z; L.targ = q4.w;
#undef RTS_F64
        if (COUNT) n_tris++;
        const TriHit h = tri_test(L, prev, dir, tmin, RTS_DEFAULT_TMAX);
        if (h.ok) {
            const float tf = (float)h.t;                      // rtPotentialIntersection takes float, triangle_mesh.cu:167
            if ((tf > tmin) && (tf < best_t || (tf == best_t && L.prim < best_prim))) {
                best_t = tf; best_leaf = leaf; best_prim = L.prim;
                t_prune = f32_next_up_pos(tf);                 // keep equal-t candidates reachable
            }
        }
        node = below; sp--;
    }
}

// The walk of ONE ray by all 64 lanes of a wave (COOP units).  Every lane runs the ordinary depth-first step (rts_walk_step)
// on its own LDS stack; a lane without work takes the BOTTOM entry -- the largest open subtree -- of a lane that has one,
// through a 64-entry exchange row in LDS (giver k writes row[k], taker k reads it; ranks by v_mbcnt over the two ballots).
// A given entry is overwritten with the sentinel, so its owner's stack simply ends one entry higher.  The prune bound is
// shared: whenever a lane finds a closer hit the wave takes the minimum of the lanes' bounds.  The winner is the
// lexicographic minimum of (f32 t bits, global primitive id) over the lanes -- the tie rule of the per-lane walk -- and is
// broadcast, so every lane leaves with the same closest hit.  Exit: no lane holds a node (a lane only goes idle with an
// empty stack, and entries are handed over in the iteration they are taken in, so nothing is pending then).
__device__ __forceinline__ float rts_wave_min_f32(float v) { for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ unsigned long long rts_wave_min_u64(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o); v = w < v ? w : v; } return v; }
// GROUPS (round 5, late): CG = 64 is the walk above -- one ray per wave.  CG = 16: the wave walks FOUR rays, lanes 16 g .. 16 g + 15 the g-th (each lane holds its group's ray,
// redundantly within the group): ballots are cut to the group's 16 bits, ranks counted inside them, the exchange row is the group's quarter of the wave's, the prune bound and the
// winner are reduced by butterflies that stay inside the group (xor 8, 4, 2, 1).  The wave leaves when no group holds a node.  What it buys: the arithmetic outside the walk --
// ray generation, capture, shading: ~1 500 instructions per segment that a one-ray unit issues for ONE ray -- is issued once for four.
template <bool COUNT, bool VERS = false, uint32_t CG = 64u>
__device__ __forceinline__ void rts_walk_coop(const RtsTraceArgs& a, int32_t* s_stack, int32_t* s_exch, uint32_t tid, uint32_t gtid, uint32_t lane, int lds_cap, uint32_t* n_spill_lds,
                                              int root, const RtsSlabRay& lr, const dvec3& prev, const dvec3& dir, float tmin,
                                              float& best_t, int& best_leaf, uint32_t& best_prim, float& t_prune, uint32_t& n_nodes, uint32_t& n_tris, bool& hard_overflow)
{
    const int SENTINEL = RTS_STACK_SENTINEL;
    const uint32_t g0 = lane & ~(CG - 1u);                                        // first lane of this lane's group
    const unsigned long long gmask = CG == 64u ? ~0ULL : (((1ULL << (CG & 63u)) - 1ULL) << g0);
    const unsigned long long below_me = (1ULL << lane) - 1ULL;
    volatile int32_t* row = s_exch + (tid & ~63u) + g0;
    s_stack[tid] = SENTINEL;
    int sp = 1, bot = 1;                                  // pending entries of this lane: [bot, sp)
    int node = (lane & (CG - 1u)) == 0u ? root : SENTINEL;
    uint32_t steps = 0;
    const RtsSlabRay lp = VERS ? lr : rts_slab_by_plane(lr);          // (this kernel reads its records straight through: rts_fetch_record_plain; the octant versions' planes are entry / exit planes already)
    for (;;) {
        const bool busy = node != SENTINEL;
        const unsigned long long busy_m = __ballot(busy);
        if (busy_m == 0ULL) break;
        if (++steps > (1u << 24)) { hard_overflow = true; break; }                  // malformed tree guard: every wave must drain
        const unsigned long long idle_m = __ballot(true) & ~busy_m;                 // (the lanes of the wave that are in the unit at all: a last, partial tile)
        if (idle_m != 0ULL) {
            const bool can = busy && sp > bot && bot < lds_cap;                     // something pending, and in the LDS part of the stack
            const unsigned long long can_m = __ballot(can);
            if (can_m != 0ULL) {
                const unsigned long long idle_g = idle_m & gmask, can_g = can_m & gmask;
                const uint32_t n_give = min((uint32_t)__popcll(idle_g), (uint32_t)__popcll(can_g));
                const uint32_t r_can = (uint32_t)__popcll(can_g & below_me), r_idle = (uint32_t)__popcll(idle_g & below_me);
                if (can && r_can < n_give) { row[r_can] = s_stack[bot * RTS_BLOCK + tid]; s_stack[bot * RTS_BLOCK + tid] = SENTINEL; bot++; }
                __builtin_amdgcn_wave_barrier();
                if (!busy && r_idle < n_give) { node = row[r_idle]; sp = 1; bot = 1; }      // (entry 0 of the taker still holds the sentinel)
                __builtin_amdgcn_wave_barrier();
            }
        }
        bool improved = false;
        if (node != SENTINEL) {
            const float before = t_prune;
            rts_walk_step<COUNT, VERS ? RTS_WALK_VERSIONS : RTS_WALK_PLANES>(a, s_stack, tid, gtid, lds_cap, n_spill_lds, node, sp, lp, prev, dir, tmin, best_t, best_leaf, best_prim, t_prune, n_nodes, n_tris, hard_overflow);
            improved = t_prune != before;
        }
        if (__any(improved)) { for (int o = (int)CG / 2; o > 0; o >>= 1) t_prune = fminf(t_prune, __shfl_xor(t_prune, o)); }      // (the group's minimum)
    }
    // closest hit over the group's lanes: smallest f32 t (positive: its bits order like the value), then lowest global primitive id
    const unsigned long long key = best_leaf >= 0 ? (((unsigned long long)__float_as_uint(best_t) << 32) | best_prim) : ~0ULL;
    unsigned long long kmin = key;
    for (int o = (int)CG / 2; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(kmin, o); kmin = w < kmin ? w : kmin; }
    const unsigned long long has_m = __ballot(key == kmin) & gmask;                   // (every lane of the wave takes part in the ballot and the shuffle below)
    const int win = has_m ? __ffsll((long long)has_m) - 1 : (int)lane;
    const int leaf_w = __shfl(best_leaf, win);
    if (kmin != ~0ULL) {
        best_leaf = leaf_w; best_t = __uint_as_float((uint32_t)(kmin >> 32)); best_prim = (uint32_t)kmin;
        t_prune = f32_next_up_pos(best_t);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// One work unit of the trace kernel: launch index `slot` through its whole bounce chain (ray generation, closest hit per
// segment, shading, capture, write-back).  Two instantiations per kernel:
//   COOP = false : one lane per launch index -- the wave traces 64 consecutive launch indices (a tile), every lane with its
//                  own `slot` and its own walk;
//   COOP = true  : ALL 64 lanes of the wave trace the SAME launch index: the ray state is computed redundantly (uniformly)
//                  by every lane, the WALK is shared out between the lanes (rts_walk_coop), lane 0 alone counts and writes.
//                  For the tiles at the head of the cost order, whose rays walk thousands of dependent steps (grazing rays
//                  along a fuselage: ~4 000 steps per segment on BASELINE configs[3]): traced one lane per ray such a tile
//                  occupies ONE wave for 7-13 ms of a launch whose balanced time is 7.8 ms; as 64 cooperative units it is
//                  spread over 64 waves that finish each ray in a few hundred wave steps.
// The two share every expression of the payload arithmetic (same code, same operand order): results are bit-identical.
// rows of k_trace's s_rxp: the receivers' pre-filter constants (rts_rx_maybe)
#define RTS_RXP_QQW 3          // |q|^2 (1 + 1e-6)
#define RTS_RXP_R2W 4          // widened radius^2
#define RTS_RXP_QN 5           // |q|
#define RTS_RXP_QQ 6           // |q|^2
#define RTS_RXP_R2 7           // radius^2
#define RTS_RXP_E2 8           // error bound of s^2
#define RTS_RXP_CT 9           // cos, sin of the window's centre azimuth
#define RTS_RXP_ST 10
#define RTS_RXP_CH 11          // cos of its half span
#define RTS_RXP_SLO 12         // r sin(min phi), r sin(max phi)
#define RTS_RXP_SHI 13
#define RTS_RXP_OK 14          // 1: the window screen applies
#define RTS_RXP_RW 15          // sqrt(widened radius^2) (1 + 1e-6)
#define RTS_RXP_N 16
struct RtsUnitLds { int32_t* stack; int32_t* exch; double* first; unsigned long long* path; uint32_t* n; const RtsRxDev* rx; const float (*rxp)[RTS_RXP_N]; uint32_t* lane_scratch; uint32_t* walk; };      // walk: per wave [walk iterations of the tile, walks]
// The payload of one ray chain between its segments (PerRayData fields that the walk does not touch but the shading does;
// the first hit point, the path words and the counters live in LDS, RtsUnitLds).
struct RtsRay { dvec3 dir, prev; double rayLength, power, doppler, refx, refy; uint32_t reflDepth, refrDepth; int received; bool end, chain_start; };

// What happens to a ray after the walk of one segment: miss (receiver capture, Earth) or closest_hit (shading, the refracted
// child, the reflected direction).  Returns true when the chain goes on with another segment.  ONE body for the three ways a
// launch index is driven (lanes in lock step per bounce round; lanes advancing on their own, rts_trace_unit_async; one ray per
// wave, COOP): same expressions, same operand order, bit-identical results.
template <bool KEEP_ALL, bool REFR, bool COOP, uint32_t CG = 64u>
__device__ __forceinline__ bool rts_shade(const RtsTraceArgs& a, const RtsUnitLds& L_, const uint32_t tid, const uint32_t gtid, const uint32_t lane, const uint32_t slot,
                                          const uint32_t chain, const uint32_t D, const uint32_t max_refr, const dvec3& origin, const bool primary, const bool may_rx,
                                          const float best_t, const int best_leaf, const uint32_t best_prim, const float tmin, RtsRay& S, uint32_t& pending, uint32_t& refr_code0)
{
    double* const s_first = L_.first; unsigned long long* const s_path = L_.path; uint32_t* const s_n = L_.n; const RtsRxDev* const s_rx = L_.rx;
    dvec3& dir = S.dir; dvec3& prev = S.prev;
    double& rayLength = S.rayLength; double& power = S.power; double& doppler = S.doppler; double& refx = S.refx; double& refy = S.refy;
    uint32_t& reflDepth = S.reflDepth; uint32_t& refrDepth = S.refrDepth; int& received = S.received; bool& end = S.end; bool& chain_start = S.chain_start;
        if (KEEP_ALL && chain == 0 && (!COOP || (lane & (CG - 1u)) == 0u)) {
            const size_t hidx = (size_t)slot * (a.max_refl + 1) + reflDepth;
            a.hit_prim[hidx] = (best_leaf >= 0) ? (int32_t)best_prim : -1;
            a.hit_t[hidx] = (best_leaf >= 0) ? best_t : 0.0f;
        }

        if (best_leaf < 0) {
            // -------------------------------------------------------- miss, ray_tracer.cu:260-478
            if (end == false && (!primary || may_rx)) {
                for (uint32_t Rx_i = 0; Rx_i < a.n_rx; Rx_i++) {
                    const RtsRxDev rx = Rx_i < RTS_RX_LDS ? s_rx[Rx_i] : a.rx[Rx_i];
                    double t[2] = {0, 0};
                    const double A = (dir.x)*(dir.x) + (dir.y)*(dir.y) + (dir.z)*(dir.z);
                    const double B = 2*(((prev.x - rx.cx)*dir.x) + ((prev.y - rx.cy)*dir.y) + ((prev.z - rx.cz)*dir.z));
                    const double C = prev.x*prev.x + prev.y*prev.y + prev.z*prev.z + (rx.cx*rx.cx) + (rx.cy*rx.cy) + (rx.cz*rx.cz) -
                                     2*((rx.cx*prev.x) + (rx.cy*prev.y) + (rx.cz*prev.z)) - rx.radius*rx.radius;
                    double discriminant = B*B - 4*A*C;
                    if (discriminant > 0.f) {
                        discriminant = sqrt(discriminant);
                        t[0] = (-B - discriminant)/(2*A);
                        t[1] = (-B + discriminant)/(2*A);
                        unsigned int received_root = 2;
    #pragma unroll
                        for (int i = 0; i < 2; i++) {
                            if ((t[i] >= 0) && ((rayLength + t[i]) > SCENE_EPS) && ((rayLength + t[i]) > SCENE_EPS_R)) {
                                const dvec3 ep = mk3(prev.x + t[i]*dir.x, prev.y + t[i]*dir.y, prev.z + t[i]*dir.z);
                                // atan2f(float, float): arguments narrow to f32 first (:326-329)
                                double theta = rts_atan2f((float)(ep.y - rx.cy), (float)(ep.x - rx.cx));
                                double phi = rts_atan2f((float)(ep.z - rx.cz), (float)sqrt(((ep.y - rx.cy) * (ep.y - rx.cy)) + ((ep.x - rx.cx) * (ep.x - rx.cx))));
                                if ((phi < -RTS_PI/2)) { theta += RTS_PI; phi = -RTS_PI - phi; }
                                if ((phi > RTS_PI/2)) { theta += RTS_PI; phi = RTS_PI - phi; }
                                double maxTheta1 = rx.maxTheta, minTheta1 = rx.minTheta, maxTheta2 = maxTheta1, minTheta2 = minTheta1;
                                double maxPhi1 = rx.maxPhi, minPhi1 = rx.minPhi, maxPhi2 = maxPhi1, minPhi2 = minPhi1;
                                if ((minPhi1 < -RTS_PI/2)) { maxTheta2 += RTS_PI; minTheta2 += RTS_PI; maxPhi2 = -RTS_PI - minPhi1; minPhi2 = -RTS_PI/2; minPhi1 = -RTS_PI/2; }
                                if ((maxPhi1 > RTS_PI/2)) { maxTheta2 += RTS_PI; minTheta2 += RTS_PI; minPhi2 = RTS_PI - maxPhi1; maxPhi2 = RTS_PI/2; maxPhi1 = RTS_PI/2; }
                                if (((rts_angle_in_range(theta, minTheta1, maxTheta1)) && (rts_angle_in_range(phi, minPhi1, maxPhi1))) ||
                                    ((rts_angle_in_range(theta, minTheta2, maxTheta2)) && (rts_angle_in_range(phi, minPhi2, maxPhi2)))) {
                                    if (received_root == 2) received_root = i;
                                    else if (t[received_root] > t[i]) received_root = i;
                                }
                            }
                        }
                        if (received_root < 2) {
                            end = true;                                                    // :396
                            const double tr = received_root == 0 ? t[0] : t[1];
                            const dvec3 ep = mk3(prev.x + tr*dir.x, prev.y + tr*dir.y, prev.z + tr*dir.z);
                            if ((reflDepth == 0) && (refrDepth == 0)) {                    // direct transmission :410-417
                                const dvec3 RxRange = sub3(ep, origin);
                                if (len3(RxRange) >= SCENE_EPS) {
                                    power = 1/(4*RTS_PI*4*RTS_PI*(magsq3(RxRange)));
                                    doppler = 0;
                                    rayLength += tr;
                                    received = (int)Rx_i;
                                }
                            } else {                                                       // :419-425
                                const dvec3 RxRange = sub3(ep, prev);
                                if (len3(RxRange) >= SCENE_EPS_R) {
                                    power *= 1/((magsq3(RxRange))*4*RTS_PI*4*RTS_PI);
                                    rayLength += tr;
                                    received = (int)Rx_i;
                                }
                            }
                        }
                    }
                }
            }
            // Earth sphere :438-476 (both roots require rayLength > 0, :464).  It can only touch a ray that was NOT captured (end == false), and such a ray leaves no
            // record unless every ray is kept: the block -- a quadratic, a square root and two divisions in f64 per missing bounce segment -- is compiled into the
            // KEEP_ALL builds only (the tests' full-output comparisons, incl. every Earth branch of tests/test_capture_branches.py); what a simulator sees cannot differ.
            if (KEEP_ALL && end == false && rayLength > 0) {
                const double d_earthRadius = 6378136;
                const double A = (dir.x)*(dir.x) + (dir.y)*(dir.y) + (dir.z)*(dir.z);
                const double B = 2*(prev.x*dir.x + prev.y*dir.y + prev.z*dir.z);
                const double C = prev.x*prev.x + prev.y*prev.y + prev.z*prev.z - d_earthRadius*d_earthRadius;
                double discriminant = B*B - 4*A*C;
                if (discriminant > 0.f) {
                    discriminant = sqrt(discriminant);
                    const double t0 = (-B - discriminant)/(2*A), t1 = (-B + discriminant)/(2*A);
                    if ((t0 >= 0) && (rayLength > 0)) { end = true; rayLength += t0; }
                    if ((t1 >= 0) && (rayLength > 0)) { end = true; rayLength += t1; }
                }
            }
            return false;
        }

        // ------------------------------------------------------------ closest_hit, normal_shader.cu:128-340
        if (!((end == false) && ((refrDepth < max_refr) || (reflDepth < a.max_refl)))) return false;   // gate :134 ; absorbed hit leaves the payload untouched
        if (!COOP || (lane & (CG - 1u)) == 0u) atomicAdd(&s_n[RTS_BLOCK + tid], 1u);
        const RtsLeafTri L = a.leaves[best_leaf];
        const RtsTargetDev T = a.targets[L.targ];
        if (refrDepth != 1) {                                              // path column (:140-146)
            const uint32_t col = reflDepth + refrDepth;
            if (col < D) {
                const uint64_t code = (uint64_t)(L.targ + 1);
                unsigned long long* pw = &s_path[(col < 8 ? 0 : RTS_BLOCK) + tid];
                const uint32_t sh = 8 * (col & 7u);
                *pw = (*pw & ~(0xffULL << sh)) | (code << sh);
            }
        }
        const float hit_t = best_t;
        const dvec3 hitPoint = mk3(prev.x + (double)hit_t*dir.x, prev.y + (double)hit_t*dir.y, prev.z + (double)hit_t*dir.z);   // :149-152
        rayLength += hit_t;                                                // :153
        if ((reflDepth == 0) && (refrDepth == 0)) {                        // :159-166
            s_first[tid] = hitPoint.x; s_first[RTS_BLOCK + tid] = hitPoint.y; s_first[2 * RTS_BLOCK + tid] = hitPoint.z;
            const dvec3 TxRange = sub3(hitPoint, origin);
            if (len3(TxRange) >= SCENE_EPS) power = 1/((magsq3(TxRange))*4*RTS_PI);
            else end = true;
        } else {                                                           // :167-173
            const dvec3 TargRange = sub3(hitPoint, prev);
            if (len3(TargRange) >= SCENE_EPS_R) power *= 1/((magsq3(TargRange))*4*RTS_PI);
            else end = true;
        }
        // attribute normal (triangle_mesh.cu:169-194): recompute the accepted test, same bits
        const TriHit h = tri_test(L, prev, dir, tmin, RTS_DEFAULT_TMAX);
        prev = hitPoint;                                                   // :176
        dvec3 normal;
        if (a.smooth) {
            const uint32_t* ni = a.tri_nidx + 3*(size_t)L.prim;
            if (T.perface_normals) {
                const double* n = a.normals + 3*(size_t)ni[0];
                normal = mk3(n[0], n[1], n[2]);
            } else {
                const double* n0 = a.normals + 3*(size_t)ni[0]; const double* n1 = a.normals + 3*(size_t)ni[1]; const double* n2 = a.normals + 3*(size_t)ni[2];
                const double w = 1.0f - h.beta - h.gamma;
                normal = mk3(n1[0]*h.beta + n2[0]*h.gamma + n0[0]*w, n1[1]*h.beta + n2[1]*h.gamma + n0[1]*w, n1[2]*h.beta + n2[2]*h.gamma + n0[2]*w);
            }
            normal = unit3(normal);
        } else {
            normal = unit3(h.n);
        }
        // f32 direction of the current OptiX ray: primary = normalise_float3(rayDir_d3) (ray_tracer.cu:208);
        // bounce / refracted = the f32 reflect()/refract() result itself (normal_shader.cu:242,296-297)
        const fvec3 dirf = (chain == 0 && chain_start) ? unit3_to_f32(dir) : mk3f((float)dir.x, (float)dir.y, (float)dir.z);
        const fvec3 nf = unit3_to_f32(normal);

        // ---- refraction branch (:191-282): prd_refr = prd; prd_refr.refrIndex.x = prd_refr.refrIndex.y
        const double rrefx = refy;                                         // prd_refr.refrIndex.x
        if (REFR) {
            if ((fabs(T.reflCoeff) != 1.00000f) && (refrDepth < max_refr) && (reflDepth == 0)) {   // :198
                const double rrefy = (rrefx == 1) ? T.refrIndex : 1.0;     // :201-206
                const float ratio = (float)(rrefy / rrefx);                // :209
                fvec3 rd;
                if (refract3f(rd, dirf, nf, ratio)) {                      // :212
                    RtsChildState cs;
                    cs.prevx = prev.x; cs.prevy = prev.y; cs.prevz = prev.z; cs.firstx = s_first[tid]; cs.firsty = s_first[RTS_BLOCK + tid]; cs.firstz = s_first[2 * RTS_BLOCK + tid];
                    cs.rayLength = rayLength; cs.refx = rrefx; cs.refy = rrefy; cs.end = end ? 1u : 0u;
                    double cpower = power;
                    if ((reflDepth + 1) < (a.max_refl + 1)) cpower *= (1 - fabs(T.reflCoeff));   // :245-246
                    cs.power = cpower;
                    cs.refrDepth = refrDepth + 1;                           // :247
                    const dvec3 k0 = unit3(dir);                            // :251-256
                    const dvec3 nd3 = widen3(rd);
                    const dvec3 k1 = unit3(nd3);
                    cs.doppler = doppler + dot3(mk3(T.vx, T.vy, T.vz), sub3(k1, k0));
                    cs.dx = rd.x; cs.dy = rd.y; cs.dz = rd.z;
                    cs.refr_code = (chain == 0) ? (L.targ + 1) : (uint32_t)(s_path[tid] & 0xff);   // prefill code travels with the first refraction only
                    if (chain == 0) refr_code0 = L.targ + 1;
                    a.child[(size_t)chain * a.slab_threads + gtid] = cs;
                    pending |= 1u << (chain + 1);
                    // direction history plane 0 of the child chain: RCS angle of the refraction event (:259-265)
                    float* dh = a.dir_hist + (size_t)((chain + 1) * (a.max_refl + 1)) * 3 * a.n_rays;
                    if (!COOP || (lane & (CG - 1u)) == 0u) { dh[slot] = rd.x; dh[(size_t)a.n_rays + slot] = rd.y; dh[2*(size_t)a.n_rays + slot] = rd.z; }
                }
            }
        }
        reflDepth++;                                                       // :286
        refy = rrefx; refx = rrefx;                                        // :289-290
        chain_start = false;
        if (!(reflDepth < a.max_refl + 1)) return false;                          // :293 (can fail only inside a refracted chain)
        const fvec3 nd = reflect3f(dirf, nf);                              // :296
        power *= T.reflCoeff;                                              // :298
        // Doppler :302-314 -- doppler += V . (unit(new dir) - unit(old dir)): two f64 normalisations (a square root and three divisions each) per shaded hit
        // for a sum that only a RECEIVED ray ever shows.  The product builds without refraction leave it to the expansion of the received records
        // (rts_post.hip expand_row), which forms the same unit vectors from the direction history for the RCS angles anyway and adds the same terms in the
        // same order; the KEEP_ALL builds (every ray's record is output) and the refracting ones (the child inherits the running sum) keep it here.
        if (KEEP_ALL || REFR) {
            const dvec3 k0 = unit3(dir);                                   // :302
            const dvec3 k1 = unit3(widen3(nd));                            // :304
            doppler += dot3(mk3(T.vx, T.vy, T.vz), sub3(k1, k0));          // :314
        }
        dir = widen3(nd);                                                  // :303
        {   // direction history: the RCS angles of received rays are rebuilt from it (:320-326)
            const size_t plane = REFR ? (size_t)chain * (a.max_refl + 1) + reflDepth : (size_t)(reflDepth - 1);
            float* dh = a.dir_hist + plane * 3 * a.n_rays;
            if (!COOP || (lane & (CG - 1u)) == 0u) { dh[slot] = nd.x; dh[(size_t)a.n_rays + slot] = nd.y; dh[2*(size_t)a.n_rays + slot] = nd.z; }
        }
    return true;
}


// End of a chain: the record of a received ray (and of every ray in the KEEP_ALL builds), ray_tracer.cu:246-253, normal_shader.cu:272-279
template <bool KEEP_ALL, bool REFR, bool COOP, uint32_t CG = 64u>
__device__ __forceinline__ void rts_write_back(const RtsTraceArgs& a, const RtsUnitLds& L_, const uint32_t tid, const uint32_t lane, const uint32_t slot, const uint32_t chain,
                                               const RtsRay& S, const uint32_t pending, const uint32_t refr_code0)
{
    const double* const s_first = L_.first; const unsigned long long* const s_path = L_.path;
    const dvec3& prev = S.prev; const double rayLength = S.rayLength, power = S.power, doppler = S.doppler;
    const uint32_t reflDepth = S.reflDepth, refrDepth = S.refrDepth; const int received = S.received;
    const bool recv = received >= 0;
    if ((recv || KEEP_ALL) && (!COOP || (lane & (CG - 1u)) == 0u)) {
        RtsEndRecord r;
        r.rayLength = rayLength; r.power = power; r.doppler = (KEEP_ALL || REFR) ? doppler : 0.0;      // deferred: rts_shade, expand_row
        r.prevx = prev.x; r.prevy = prev.y; r.prevz = prev.z;
        r.firstx = s_first[tid]; r.firsty = s_first[RTS_BLOCK + tid]; r.firstz = s_first[2 * RTS_BLOCK + tid];
        r.path_lo = s_path[tid]; r.path_hi = s_path[RTS_BLOCK + tid]; r.slot = slot; r.received = received; r.reflDepth = reflDepth;
        r.pad = chain | (refrDepth << 2) | ((chain == 0 ? refr_code0 : 0u) << 8) | ((pending & 6u) << 15);   // chain, refrDepth, prefill code, spawned children
        if (KEEP_ALL) a.all_records[(size_t)chain * a.n_rays + slot] = r;
        if (recv) {
            // the compiler folds this into one atomic per wave (v_mbcnt + s_bcnt1)
            unsigned long long idx = atomicAdd(&a.counters[0], 1ULL);
            a.recv_records[idx] = r;
        }
    }
}

// Can a PRIMARY ray of (unit, f32) direction d -- or, delta > 0, any primary ray within the angle delta of it -- be CAPTURED by the receiver whose
// constants are K (k_trace's s_rxp row; RTS_RXP_*)?  Two conservative questions, both in f32:
//  (1) can it come within the widened radius of the capture sphere at all (round 2; the widening covers the f32 arithmetic here and the
//      reference's own cancelling quadratic, k_trace);
//  (2) round 5 -- if it does cross the sphere: can either crossing point lie inside the receiver's angular capture window
//      (ray_tracer.cu:326-375)?  A monostatic radar's capture sphere sits right in front of the antenna and EVERY ray of the beam crosses it
//      (so do all of BASELINE configs[3]'s: its transmitter is 112 m from the centre of an 80 m sphere); the window, which looks back at the
//      receiver, excludes them -- but only the exact f64 miss program knew, at ~500 instructions per receiver and ray.  Here: the
//      crossing points p = t d - q for t = b -+ s (b = q.d, s^2 = b^2 - (|q|^2 - r^2)) and two inequalities without an arctangent --
//      the azimuth test as a cosine against the window's centre direction, the elevation test as p_z against r sin(min / max phi).
//      Error budget (e2, ep below; derivation in DESIGN.md section 4): the reference's roots differ from geometry by eC / (2 s) along the
//      ray, eC <= 1e-14 (|o|^2 + |c|^2) the rounding of its constant term; this arithmetic's s^2 is off by <= 2e-6 |q|^2 + 1e-6 r^2 (f32 q, the
//      pre-filter's direction good to 3e-7 rad); a point error ep turns the azimuth by <= 1.1 ep / rho (rho = the point's distance from the
//      sphere's polar axis) and moves p_z by ep.  Near-tangent rays (s^2 < 16 e2), points near the polar axis (ep > rho / 4), windows that
//      reach over a pole or span >= 3 rad in azimuth (RTS_RXP_OK = 0): "maybe".  A root is ignored when it is certainly below the
//      reference's t > SCENE_EPS (the transmitter ON the sphere: the monostatic case).
//      delta > 0: every ray of the bundle lies within delta of d; its b differs by <= |q| delta, its roots by dt = |q| delta (1 + (2 |b| +
//      |q| delta) / (1.7 s)), its crossing points by dt + |t| delta (valid while the bundle's smallest s^2 >= s^2 / 2, else "maybe").
__device__ __forceinline__ bool rts_rx_maybe(const float* K, const float dx, const float dy, const float dz, const float dd, const float delta)
{
    const float qx = K[0], qy = K[1], qz = K[2], qn = K[RTS_RXP_QN];
    const float b = qx*dx + qy*dy + qz*dz;
    const float R = K[RTS_RXP_RW] + qn * delta, r2w = delta > 0.0f ? R * R : K[RTS_RXP_R2W];
    const bool inside = K[RTS_RXP_QQW] <= K[RTS_RXP_R2W] * 1.01f;                                   // the transmitter is inside the (widened) sphere
    const bool ahead = b > -(1.0e-3f + delta) * qn && (b*b - (K[RTS_RXP_QQW] - r2w) * dd) >= 0.0f;
    if (!(inside || ahead)) return false;                        // (1) it cannot reach the sphere
    if (K[RTS_RXP_OK] == 0.0f) return true;
    // (2) the crossing points against the window
    const float e2 = K[RTS_RXP_E2], s2 = b*b - (K[RTS_RXP_QQ] - K[RTS_RXP_R2]);
    if (!(s2 >= 16.0f * e2)) return true;                        // near-tangent (or not a number): the roots are not known well enough
    const float s = __builtin_sqrtf(s2);
    float dt = 0.0f;
    if (delta > 0.0f) {
        const float qd = qn * delta;
        if (!(s2 - qd * (2.2f * fabsf(b) + qd) >= 0.5f * s2)) return true;      // the bundle comes too close to tangency
        dt = qd * (1.0f + (2.0f * fabsf(b) + qd) / (1.7f * s));
    }
    const float et = 0.75f * e2 / s + 1.0e-6f * qn;
    // The SMALL root through the product of the roots, t_small t_large = |q|^2 - r^2: that constant does not depend on the direction, so for EVERY ray of a bundle
    // |t_small| <= (| |q|^2 - r^2 | + e2) / (|t_large| - et - dt) -- and with the transmitter ON the capture sphere (the monostatic case: |q| = r) that is ~1e-4 m whatever
    // the direction, where b - s carries the whole bundle's spread dt (0.5 m for a wave tile's rays from 50 m: the tile-level screen then said "maybe" for every tile
    // of a monostatic scene -- BASELINE configs[4]: an empty launch 0.23 instead of 0.03 ms, profiles/r05z_monostatic_bundle.log).
    const float tl_min = fabsf(b >= 0.0f ? b + s : b - s) - et - dt;
    const bool small_gone = tl_min > 0.0f && (fabsf(K[RTS_RXP_QQ] - K[RTS_RXP_R2]) + e2) / tl_min + et < 0.004999f;      // the small root is certainly below the reference's t > SCENE_EPS
    bool maybe = false;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float t = k == 0 ? b - s : b + s;
        if (t + et + dt < 0.004999f) continue;                   // certainly not a valid root (ray_tracer.cu:314: t >= 0, rayLength + t > SCENE_EPS)
        if (small_gone && (k == 0) == (b >= 0.0f)) continue;     // (the small root: b - s for b >= 0, b + s else)
        const float px = __builtin_fmaf(t, dx, -qx), py = __builtin_fmaf(t, dy, -qy), pz = __builtin_fmaf(t, dz, -qz);
        const float ep = et + dt + (fabsf(t) + qn) * (2.0e-6f + delta);
        const float rho = __builtin_sqrtf(px*px + py*py);
        if (!(ep <= 0.25f * rho)) { maybe = true; continue; }    // near the polar axis: the azimuth is not known
        const bool az_out = (px * K[RTS_RXP_CT] + py * K[RTS_RXP_ST]) < rho * (K[RTS_RXP_CH] - 1.1f * ep / rho - 2.0e-6f);
        const bool el_out = pz < K[RTS_RXP_SLO] - 2.0f * ep - 2.0e-6f * qn || pz > K[RTS_RXP_SHI] + 2.0f * ep + 2.0e-6f * qn;
        maybe = maybe || !(az_out || el_out);
    }
    return maybe;
}

// The conservative f32 pre-filter of a primary ray: can it meet any triangle's projection (the pulse's direction bitmap,
// RtsMaskFrame) / come within the widened radius of a receiver sphere?  rxp: the receivers' constants (k_trace's s_rxp).
__device__ __forceinline__ void rts_prefilter(const RtsLaunchConsts& lc, const uint32_t slot, const bool mask_on, const uint32_t* __restrict__ pmask, const uint32_t n_rx,
                                              const float (*rxp)[RTS_RXP_N], bool& may_target, bool& may_rx)
{
    // ray_generation in f32: lattice point, ONE matrix (Rot1 Rot: the reference's two normalisations in between only scale), normalise.
    // Good to ~2e-7 rad; the mask's margin is a whole cell (>= 1e-5), the receivers' radii are widened by 1 %
    uint32_t lx, ly, lz; rts_lattice_coords(lc, slot, lx, ly, lz);
    const float vx = __builtin_fmaf(lc.f_st[0], (float)lx, lc.f_bs[0]), vy = __builtin_fmaf(lc.f_st[1], (float)ly, lc.f_bs[1]), vz = __builtin_fmaf(lc.f_st[2], (float)lz, lc.f_bs[2]);
    float dx = lc.f_m[0]*vx + lc.f_m[1]*vy + lc.f_m[2]*vz, dy = lc.f_m[3]*vx + lc.f_m[4]*vy + lc.f_m[5]*vz, dz = lc.f_m[6]*vx + lc.f_m[7]*vy + lc.f_m[8]*vz;
    const float inv = __frsqrt_rn(dx*dx + dy*dy + dz*dz); dx *= inv; dy *= inv; dz *= inv;
    if (mask_on) {                                       // is any triangle's projection near this direction? (RtsMaskFrame)
        const RtsMaskFrame& mf = lc.mask;
        const float w = dx * mf.bx + dy * mf.by + dz * mf.bz;
        const float iw = __builtin_amdgcn_rcpf(w);          // (v_rcp_f32, 1 ulp: 1e-7 of a coordinate that is compared with cells of >= 1e-5 after a one-cell margin; two IEEE divisions were ~20 instructions)
        const float fu = ((dx * mf.ux + dy * mf.uy + dz * mf.uz) * iw - mf.u0) * mf.inv_du, fv = ((dx * mf.vx + dy * mf.vy + dz * mf.vz) * iw - mf.v0) * mf.inv_dv;
        if (w > 0.0f && fu >= 0.0f && fv >= 0.0f && fu < (float)mf.n && fv < (float)mf.n) {
            const uint32_t cell = (uint32_t)fv * mf.n + (uint32_t)fu;
            may_target = ((pmask[cell >> 5] >> (cell & 31u)) & 1u) != 0u;
        }
    }
    may_rx = false;
    const float dd = dx*dx + dy*dy + dz*dz;
    for (uint32_t Rx_i = 0; Rx_i < n_rx; Rx_i++) may_rx = may_rx || rts_rx_maybe(rxp[Rx_i], dx, dy, dz, dd, 0.0f);      // can this receiver capture the ray?
}

// The pre-filter's question asked of a whole ROW SEGMENT of the launch lattice -- launch indices lx_a .. lx_b of row (ly, lz) -- by ONE lane:
// can ANY of its primary rays meet a triangle's projection or come within the widened radius of a receiver sphere?  (Dead-tile batches
// of k_trace: most wave tiles of a pulse are dead, and a wave spent ~1.5 us on each -- a dependent chain of the order's load, 64 per-lane
// filters, the mask load, a ballot -- to learn it.)  Conservative with respect to the per-ray filter's own arithmetic:
//  * the un-normalised direction is LINEAR in lx, d(lx) = M (bs + st (lx, ly, lz)), so the rays of the segment lie in one plane through the
//    transmitter, on the great-circle arc between the two end rays;
//  * the mask coordinates fu, fv are ratios of linear functions of lx with the same denominator w = d . b: where w > 0 at both ends it is
//    positive in between and (fu, fv) runs along the STRAIGHT segment between the end rays' points (a projective map of the lattice row),
//    monotonically.  The cells that segment passes through are tested; the mask's one-cell margin covers the f32 rounding of the ends
//    exactly as it covers a single ray's.  An end outside the mask's frame, w <= 0, or a segment of more than 96 cells: "maybe";
//  * a ray that passes within R of a receiver's centre q does so at an angle asin(R / |q|) from q's direction; every ray of the segment is
//    within delta = angle(end a, end b) of end a, so end a passes within R + |q| delta: the per-ray test at end a with that radius (and the
//    "pointing away" bound relaxed by delta).  delta <= 1.6 x the chord of the normalised ends (+ 4e-6 for their f32 rounding).
__device__ __forceinline__ bool rts_segment_maybe(const RtsLaunchConsts& lc, const uint32_t lx_a, const uint32_t lx_b, const uint32_t ly, const uint32_t lz, const bool has_prims, const bool mask_on,
                                                  const uint32_t* __restrict__ pmask, const uint32_t n_rx, const float (*rxp)[RTS_RXP_N])
{
    const float vy = __builtin_fmaf(lc.f_st[1], (float)ly, lc.f_bs[1]), vz = __builtin_fmaf(lc.f_st[2], (float)lz, lc.f_bs[2]);
    const float vxa = __builtin_fmaf(lc.f_st[0], (float)lx_a, lc.f_bs[0]), vxb = __builtin_fmaf(lc.f_st[0], (float)lx_b, lc.f_bs[0]);
    float ax = lc.f_m[0]*vxa + lc.f_m[1]*vy + lc.f_m[2]*vz, ay = lc.f_m[3]*vxa + lc.f_m[4]*vy + lc.f_m[5]*vz, az = lc.f_m[6]*vxa + lc.f_m[7]*vy + lc.f_m[8]*vz;
    float bx = lc.f_m[0]*vxb + lc.f_m[1]*vy + lc.f_m[2]*vz, by = lc.f_m[3]*vxb + lc.f_m[4]*vy + lc.f_m[5]*vz, bz = lc.f_m[6]*vxb + lc.f_m[7]*vy + lc.f_m[8]*vz;
    { const float ia = __frsqrt_rn(ax*ax + ay*ay + az*az), ib = __frsqrt_rn(bx*bx + by*by + bz*bz); ax *= ia; ay *= ia; az *= ia; bx *= ib; by *= ib; bz *= ib; }
    if (has_prims) {
        if (!mask_on) return true;
        const RtsMaskFrame& mf = lc.mask;
        const float wa = ax * mf.bx + ay * mf.by + az * mf.bz, wb = bx * mf.bx + by * mf.by + bz * mf.bz;
        if (!(wa > 0.0f && wb > 0.0f)) return true;
        const float iwa = __builtin_amdgcn_rcpf(wa), iwb = __builtin_amdgcn_rcpf(wb);
        const float fua = ((ax * mf.ux + ay * mf.uy + az * mf.uz) * iwa - mf.u0) * mf.inv_du, fva = ((ax * mf.vx + ay * mf.vy + az * mf.vz) * iwa - mf.v0) * mf.inv_dv;
        const float fub = ((bx * mf.ux + by * mf.uy + bz * mf.uz) * iwb - mf.u0) * mf.inv_du, fvb = ((bx * mf.vx + by * mf.vy + bz * mf.vz) * iwb - mf.v0) * mf.inv_dv;
        const float ulo = fminf(fua, fub), uhi = fmaxf(fua, fub), vlo = fminf(fva, fvb), vhi = fmaxf(fva, fvb);
        if (!(ulo >= 0.0f && vlo >= 0.0f && uhi < (float)mf.n && vhi < (float)mf.n)) return true;      // (an end outside the frame -- or not a number -- : the per-ray filter says "maybe" there)
        // The rays' mask points lie ON the straight segment between the two ends' (a projective map takes the lattice row to a line):
        // along the range axis of the lattice that is up to ~16 cells at the beam's edge, mostly radial -- its bounding box would be
        // hundreds of cells.  The segment is walked column by column along its major axis; per column the cells its part of the segment
        // passes through, widened by 1/32 cell for the rounding of this interpolation (the rays' own rounding is the mask's one-cell margin's).
        const bool major_u = (uhi - ulo) >= (vhi - vlo);
        float a0 = major_u ? fua : fva, a1 = major_u ? fub : fvb, b0 = major_u ? fva : fua, b1 = major_u ? fvb : fub;
        if (a0 > a1) { const float t0 = a0; a0 = a1; a1 = t0; const float t1 = b0; b0 = b1; b1 = t1; }
        const uint32_t c0 = (uint32_t)a0, c1 = (uint32_t)a1;
        if (c1 - c0 > 96u) return true;                               // (a segment across a tenth of the beam: not a tile of nearly parallel rays)
        const float slope = (a1 - a0) > 1.0e-6f ? (b1 - b0) / (a1 - a0) : 0.0f;
        const float nmax = (float)(mf.n - 1u);
        uint32_t found = 0u;
        for (uint32_t c = c0; c <= c1; c++) {
            const float la = fmaxf(a0, (float)c), ha = fminf(a1, (float)(c + 1u));
            const float bl = b0 + slope * (la - a0), bh = b0 + slope * (ha - a0);
            const uint32_t r0 = (uint32_t)fminf(fmaxf(fminf(bl, bh) - 0.03125f, 0.0f), nmax), r1 = (uint32_t)fminf(fmaxf(fmaxf(bl, bh) + 0.03125f, 0.0f), nmax);
            for (uint32_t r = r0; r <= r1; r++) { const uint32_t cell = major_u ? r * mf.n + c : c * mf.n + r; found |= pmask[cell >> 5] >> (cell & 31u); }
        }
        if (found & 1u) return true;
    }
    const float ex = ax - bx, ey = ay - by, ez = az - bz;
    const float delta = 1.6f * __builtin_sqrtf(ex*ex + ey*ey + ez*ez) + 4.0e-6f;
    const float dd = ax*ax + ay*ay + az*az;
    for (uint32_t Rx_i = 0; Rx_i < n_rx; Rx_i++) if (rts_rx_maybe(rxp[Rx_i], ax, ay, az, dd, delta)) return true;      // (end a stands for the bundle: every ray is within delta of it)
    return false;
}
// ... of a WAVE TILE: 64 consecutive launch indices (aligned launches only: consecutive GLOBAL indices) = one row segment of the lattice, or
// the end of one row and the start of the next -- or, W < 64, several whole rows: the segments are taken one after the other
// (ONE inlined copy of the segment test in a loop: inlined at two call sites its loop-invariant addresses were hoisted into vector
// registers of the tile loop -- and from there into scratch)
__device__ __forceinline__ bool rts_tile_maybe(const RtsLaunchConsts& lc, const uint32_t tile, const uint32_t n_rays, const bool has_prims, const bool mask_on, const uint32_t* __restrict__ pmask,
                                               const uint32_t n_rx, const float (*rxp)[RTS_RXP_N])
{
    const uint32_t s0 = tile * 64u, s1 = min(s0 + 63u, n_rays - 1u);
    uint32_t lx, ly, lz;
    rts_lattice_coords(lc, s0, lx, ly, lz);
    uint32_t left = s1 - s0 + 1u;
    bool maybe = false;
    while (left != 0u && !maybe) {
        const uint32_t n = min(left, lc.W - lx);
        maybe = rts_segment_maybe(lc, lx, lx + n - 1u, ly, lz, has_prims, mask_on, pmask, n_rx, rxp);
        left -= n; lx = 0u;
        if (++ly == lc.W) { ly = 0u; lz++; }
    }
    return maybe;
}

// ray_generation + payload of a launch index (ray_tracer.cu:144-224), with the conservative f32 pre-filter of primary rays:
// may_target / may_rx come back false when the ray can meet no triangle / no receiver sphere.
__device__ __forceinline__ void rts_primary_setup(const RtsTraceArgs& a, const RtsLaunchConsts& lc, const RtsUnitLds& L_, const uint32_t tid, const uint32_t slot, const bool pre_on,
                                                  const bool mask_on, const dvec3& origin, RtsRay& S, bool& may_target, bool& may_rx, const bool pre_done = false)
{
    double* const s_first = L_.first; unsigned long long* const s_path = L_.path; const float (*const s_rxp)[RTS_RXP_N] = L_.rxp;
    dvec3& dir = S.dir; dvec3& prev = S.prev; double& rayLength = S.rayLength; double& power = S.power; double& doppler = S.doppler;
    // ------------------------------------------------------------ ray_generation + payload, ray_tracer.cu:144-224
    if (pre_on && !pre_done) rts_prefilter(lc, slot, mask_on, a.pmask, a.n_rx, s_rxp, may_target, may_rx);
    dir = (may_target || may_rx) ? rts_primary_dir(lc, slot) : mk3(0.0, 0.0, 0.0);
    prev = origin;
    s_first[tid] = 0.0; s_first[RTS_BLOCK + tid] = 0.0; s_first[2 * RTS_BLOCK + tid] = 0.0;
    s_path[tid] = 0ULL; s_path[RTS_BLOCK + tid] = 0ULL;
    rayLength = 0; power = 0; doppler = 0;
}

template <bool COUNT, bool KEEP_ALL, bool REFR, bool COOP, bool VERS = false>
__device__ __forceinline__ void rts_trace_unit(const RtsTraceArgs& a, const RtsLaunchConsts& lc, const RtsUnitLds& L_, const uint32_t tid, const uint32_t gtid, const uint32_t lane,
                                               const uint32_t slot, const bool pre_on, const bool mask_on, const uint32_t D, const uint32_t max_refr, const dvec3& origin,
                                               uint32_t& n_nodes, uint32_t& n_tris, bool& hard_overflow, unsigned long long (&lane_stats)[5],
                                               const uint32_t pre = 0u)      // pre: bit 0 = k_trace ran the pre-filter already, bits 1 / 2 = its may_target / may_rx
{
      int32_t* const s_stack = L_.stack; int32_t* const s_exch = L_.exch; double* const s_first = L_.first; unsigned long long* const s_path = L_.path; uint32_t* const s_n = L_.n;
      constexpr uint32_t CG = (COOP && VERS) ? (uint32_t)RTS_COOP_GROUP : 64u;      // lanes that share a ray in a cooperative unit (rts_walk_coop)

      uint32_t pending = 0;                   // bit k: chain k has been spawned
      uint32_t refr_code0 = 0;                // (target + 1) of chain 0's refraction, for the path prefill of rows >= 3
      for (uint32_t chain = 0; chain < (REFR ? 3u : 1u); chain++) {
        RtsRay S;
        dvec3& dir = S.dir; dvec3& prev = S.prev;
        double& rayLength = S.rayLength; double& power = S.power; double& doppler = S.doppler; double& refx = S.refx; double& refy = S.refy;
        uint32_t& reflDepth = S.reflDepth; uint32_t& refrDepth = S.refrDepth; bool& end = S.end; bool& chain_start = S.chain_start;
        refx = 1; refy = 1; reflDepth = 0; refrDepth = 0; S.received = -1; end = false;
        bool may_target = a.n_prims > 0, may_rx = a.n_rx > 0;   // (primary ray: what the pre-filter could not exclude)
        if (pre & 1u) { may_target = (pre & 2u) != 0u; may_rx = (pre & 4u) != 0u; }      // (k_trace ran the pre-filter already, to see whether the tile is dead)
        if (chain == 0) {
            rts_primary_setup(a, lc, L_, tid, slot, pre_on, mask_on, origin, S, may_target, may_rx, (pre & 1u) != 0u);
        } else {
            if (!REFR || !(pending & (1u << chain))) continue;
            const RtsChildState cs = a.child[(size_t)(chain - 1) * a.slab_threads + gtid];
            dir = mk3((double)cs.dx, (double)cs.dy, (double)cs.dz);          // prd_refr.rayDirection = widened f32 refract() result (:252)
            prev = mk3(cs.prevx, cs.prevy, cs.prevz);
            s_first[tid] = cs.firstx; s_first[RTS_BLOCK + tid] = cs.firsty; s_first[2 * RTS_BLOCK + tid] = cs.firstz;
            rayLength = cs.rayLength; power = cs.power; doppler = cs.doppler; refx = cs.refx; refy = cs.refy;
            refrDepth = cs.refrDepth; end = cs.end != 0;
            // path prefill by the FIRST refraction (:221-239): row W^3 gets every column, row 2 W^3 columns 0..1
            const uint64_t code = cs.refr_code;
            uint64_t path_lo = 0, path_hi = 0;
            for (uint32_t col = 0; col < (chain == 1 ? D : 2u); col++) { if (col < 8) path_lo |= code << (8 * col); else path_hi |= code << (8 * (col - 8)); }
            s_path[tid] = path_lo; s_path[RTS_BLOCK + tid] = path_hi;
        }
        chain_start = true;                    // first segment of this chain: incident epsilon, f32 direction rule below

        for (;;) {
            // ------------------------------------------------------------ rtTrace: closest hit over the targets' hierarchies
            if (!COOP || (lane & (CG - 1u)) == 0u) atomicAdd(&s_n[tid], RTS_SEG_ONE);           // (ds_add_u32, no return; bits 0-21 the lane's segments of the launch, bits 22-31 those of the current tile)
            if (!COOP) { const unsigned long long ex_ = __ballot(true);             // one bounce round more in the tile's walk statistics (first lane still in the chain)
                         if (__builtin_amdgcn_mbcnt_hi((uint32_t)(ex_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex_, 0u)) == 0u) atomicAdd(&L_.walk[2u * __builtin_amdgcn_readfirstlane(tid >> 6) + 1u], 1u); }
            const float tmin = chain_start ? SCENE_EPS : SCENE_EPS_R;          // ray_tracer.cu:209, normal_shader.cu:242,297
            float best_t = RTS_DEFAULT_TMAX;
            int best_leaf = -1; uint32_t best_prim = 0xffffffffu;
            const bool primary = chain == 0 && chain_start;
            const bool may_hit = primary ? may_target : a.n_prims > 0;
            uint32_t steps = 0;                                                // walk steps of this lane in this segment (all targets)
            if (may_hit) {
                float t_prune = RTS_DEFAULT_TMAX;
                for (uint32_t targ = 0; targ < a.n_targets; targ++) {
                    const RtsTargetDev& TG = a.targets[targ];                            // uniform index: scalar loads
                    if (TG.root < 0) continue;
                    {   // bounding sphere of the placed target (f64 on the live payload: no registers survive into the walk
                        // below, an f32 box test here cost a wave of occupancy); radius is padded on the host
                        const dvec3 q = mk3(TG.cx - prev.x, TG.cy - prev.y, TG.cz - prev.z);
                        const double qq = q.x*q.x + q.y*q.y + q.z*q.z, b = q.x*dir.x + q.y*dir.y + q.z*dir.z;
                        if (qq > TG.r2) {                                                 // origin outside the sphere
                            const double dd = dir.x*dir.x + dir.y*dir.y + dir.z*dir.z;
                            if (!(b > 0.0) || !(b*b >= (qq - TG.r2) * dd * 0.999999)) continue;   // pointing away, or passing outside
                        }
                    }
                    // the ray in target space: local = rinv * (world - pos); an affine map keeps the ray parameter, so
                    // slab distances compare directly with the f32 t of the exact (world-space, f64) triangle test
                    const dvec3 q = mk3(prev.x - TG.px, prev.y - TG.py, prev.z - TG.pz);
                    const dvec3 ol = mk3(TG.rinv[0]*q.x + TG.rinv[1]*q.y + TG.rinv[2]*q.z, TG.rinv[3]*q.x + TG.rinv[4]*q.y + TG.rinv[5]*q.z, TG.rinv[6]*q.x + TG.rinv[7]*q.y + TG.rinv[8]*q.z);
                    const dvec3 dl = mk3(TG.rinv[0]*dir.x + TG.rinv[1]*dir.y + TG.rinv[2]*dir.z, TG.rinv[3]*dir.x + TG.rinv[4]*dir.y + TG.rinv[5]*dir.z, TG.rinv[6]*dir.x + TG.rinv[7]*dir.y + TG.rinv[8]*dir.z);
                    const RtsSlabRay lr = rts_slab_setup(ol, dl, TG.ew);
                    // (VERS) the ray's octant in the target's frame: bit k = the ray runs towards -axis k = the sign of its entry reciprocal,
                    // the rule by which the role fetch picks the entry plane (a dropped axis has iN = +0: either plane will do)
                    const uint32_t oct = VERS ? ((__float_as_uint(lr.iNx) >> 31) | ((__float_as_uint(lr.iNy) >> 31) << 1) | ((__float_as_uint(lr.iNz) >> 31) << 2)) : 0u;
                    // Traversal stack: entry e of lane `tid` lives at s_stack[e * RTS_BLOCK + tid] for e < stack_lds and in the
                    // global slab above that.  Entry 0 holds a sentinel, so "pop" never needs an emptiness test and the walk
                    // ends when the sentinel comes off.  The common step is branch free: the entry a lane falls back to
                    // (`below`) is read at the top of the step, beside the step's global loads; children are stored
                    // unconditionally at the top of the stack and the stack pointer advances by the hit predicate.
                    const int SENTINEL = RTS_STACK_SENTINEL;
                    const int lds_cap = (int)a.stack_lds;
                    if (COOP) {
                        rts_walk_coop<COUNT, VERS, CG>(a, s_stack, s_exch, tid, gtid, lane, lds_cap, &s_n[2 * RTS_BLOCK + tid], VERS ? (int)(((uint32_t)TG.root << 3) | oct) : TG.root, lr, prev, dir, tmin, best_t, best_leaf, best_prim, t_prune,
                                             n_nodes, n_tris, hard_overflow);
                        if (COUNT) steps = 1u;                                       // (the segment entered a hierarchy: RtsStats::walked_segments)
                    } else {
                        s_stack[tid] = SENTINEL;
                        int sp = 1;
                        int node = VERS ? (int)(((uint32_t)TG.root << 3) | oct) : TG.root;
                        uint32_t wave_steps = 0;                                     // (wave-uniform: a scalar register; the per-lane count is the counting build's)
                        while (node != SENTINEL) {
                            wave_steps = __builtin_amdgcn_readfirstlane(wave_steps) + 1u;      // (uniform by construction; said so, or the count lives in a vector register of the walk)
                            if (wave_steps > (1u << 24)) { hard_overflow = true; break; }   // malformed tree guard: every wave must drain
                            if (COUNT) steps++;
                            rts_walk_step<COUNT, VERS ? RTS_WALK_VERSIONS : RTS_WALK_ROLES>(a, s_stack, tid, gtid, lds_cap, &s_n[2 * RTS_BLOCK + tid], node, sp, lr, prev, dir, tmin, best_t, best_leaf, best_prim, t_prune,
                                                 n_nodes, n_tris, hard_overflow);
                        }
                        // the tile's walk statistics (LONG WALKS flag, k_trace): iterations the wave spent in this walk -- its slowest lane's;
                        // by the first lane that is in the walk at all
                        { const unsigned long long ex_ = __ballot(true);
                          if (__builtin_amdgcn_mbcnt_hi((uint32_t)(ex_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex_, 0u)) == 0u) { atomicAdd(&L_.walk[2u * __builtin_amdgcn_readfirstlane(tid >> 6)], wave_steps); } }
                    }
                }
            }
            if (COUNT && !COOP) {
                // lane statistics of the bounce round (counting build): the wave issues max(steps) walk iterations for its 64 lanes;
                // `alive` of them take part in the round at all, and they walk sum(steps) iterations between them
                //   [0] += 64 max   (lane-steps issued)   [1] += alive max   (... to lanes that are in the round)   [2] += sum   (useful)
                uint32_t* ls = L_.lane_scratch + 2u * (tid >> 6);
                const unsigned long long act = __ballot(true);
                if (lane == (uint32_t)(__ffsll((long long)act) - 1)) { ls[0] = 0u; ls[1] = 0u; }
                atomicMax(&ls[0], steps); atomicAdd(&ls[1], steps);
                const uint32_t smax = ls[0], ssum = ls[1];
                lane_stats[0] += 64ull * smax; lane_stats[1] += (unsigned long long)__popcll(act) * smax; lane_stats[2] += ssum;
                lane_stats[3] += (unsigned long long)__popcll(__ballot(steps > 0u));      // segments that walked at all (the others: cleared by the pre-filter or by every target's bounding sphere)
                lane_stats[4] += (unsigned long long)__popcll(__ballot(steps == 0u)) * smax;      // lane-steps issued to lanes that are in the round but NEVER STARTED a walk in it -- what packing the live launch indices of several tiles into dense waves could recover
            }
            if (COUNT && COOP && steps) lane_stats[3] += 1ull;
            {   // (a cooperative unit of several rays: a ray whose chain has ended waits for the unit's other rays -- the walk's ballots are the wave's)
                const bool go_on_ = rts_shade<KEEP_ALL, REFR, COOP, CG>(a, L_, tid, gtid, lane, slot, chain, D, max_refr, origin, primary, may_rx, best_t, best_leaf, best_prim, tmin, S, pending, refr_code0);
                if (!go_on_) break;
            }
        }

        rts_write_back<KEEP_ALL, REFR, COOP, CG>(a, L_, tid, lane, slot, chain, S, pending, refr_code0);
      }   // chain
}

// a duration on the constant-rate counter (100 MHz ticks) in the unit of the tile-cost records: 64 shader clocks at 2.4 GHz = 8/3 ticks (k_trace)
#define RTS_COST_UNITS(ticks) (((unsigned long long)(ticks) * 3ULL) >> 3)

// ASYNCHRONOUS BOUNCES (the north star's "wave-level ballot / compaction of active rays"): the same launch
// index per lane, but the lanes of a wave no longer move from segment to segment in lock step.  In rts_trace_unit a wave walks
// until its SLOWEST lane has finished the segment (counting build, rts_get_lane_stats: 21-28 % of the issued lane-steps belong
// to lanes that wait for that one, against 2 % to lanes whose ray has ended -- all that re-packing survivors between rounds
// could recover).  Here a lane is in one of two states -- WALKING (node != sentinel) or ADVANCING (its walk of the current
// target is over: try the next target's bounding sphere, or shade the segment and open the next one, or end) -- and the wave
// alternates between an advance phase, run for the lanes that need it, and a walk phase that ends when no lane walks any more
// OR when `idle_limit` lanes have come out of their walks: those are then advanced and re-join the walkers, whose walk state
// (node, stack pointer, closest hit, prune bound, target: registers; the stack: LDS) simply stays where it is.
// idle_limit = 64 is the lock-step schedule.  A low limit costs shading passes with few lanes in them (the shading code is
// ~30 walk steps' worth of instructions), so the limit is a property of the TILE'S AGE: young tiles -- the 97 % that end within
// a.async_age -- use a.async_idle0, tiles older than that a.async_idle1 (rts_api.hip: RTS_ASYNC_IDLE0/1, RTS_ASYNC_AGE).
// No refraction chains here (REFR launches use rts_trace_unit).
template <bool COUNT, bool KEEP_ALL>
__device__ __forceinline__ void rts_trace_unit_async(const RtsTraceArgs& a, const RtsLaunchConsts& lc, const RtsUnitLds& L_, const uint32_t tid, const uint32_t gtid, const uint32_t lane,
                                                     const uint32_t slot_in, const bool pre_on, const bool mask_on, const uint32_t D, const dvec3& origin, const long long tile_t0,
                                                     uint32_t& n_nodes, uint32_t& n_tris, bool& hard_overflow, unsigned long long (&lane_stats)[5])
{
    int32_t* const s_stack = L_.stack; uint32_t* const s_n = L_.n;
    const int SENTINEL = RTS_STACK_SENTINEL;
    const int lds_cap = (int)a.stack_lds;
    // Registers are what this schedule is short of: the walk state of the lanes that are NOT being advanced has to survive the
    // shading code, on a kernel held to 128 VGPRs.  So between the phases a lane keeps its small payload fields packed in one
    // word (`st`), the launch index is re-formed from the tile's first index (scalar) and the lane number where it is needed,
    // and the malformed-tree guard counts the WAVE's walk iterations (scalar).
    //   st: bits 0-7 reflDepth, 8 end, 9 chain_start, 10 live, 11 may_rx, 16-31 received + 1
    const uint32_t slot0 = __builtin_amdgcn_readfirstlane(slot_in - lane);
#define RTS_SLOT() (slot0 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
    uint32_t st;
    dvec3 dir, prev; double rayLength, power, doppler;
    uint32_t targ;                                              // next target to try; the one being walked is targ - 1
    {
        RtsRay S;
        bool may_target = a.n_prims > 0, may_rx = a.n_rx > 0;
        rts_primary_setup(a, lc, L_, tid, slot_in, pre_on, mask_on, origin, S, may_target, may_rx);
        dir = S.dir; prev = S.prev; rayLength = S.rayLength; power = S.power; doppler = S.doppler;
        st = (1u << 9) | (1u << 10) | (may_rx ? 1u << 11 : 0u);
        targ = may_target ? 0u : a.n_targets;                   // (a primary ray the pre-filter cleared tries no target)
    }
    // the open segment
    float best_t = RTS_DEFAULT_TMAX, t_prune = RTS_DEFAULT_TMAX;
    int best_leaf = -1; uint32_t best_prim = 0xffffffffu;
    int node = SENTINEL, sp = 1;
    uint32_t wave_steps = 0;
    atomicAdd(&s_n[tid], RTS_SEG_ONE);
    for (;;) {
        // ---------------------------------------------------------------- advance: until the lane walks or its ray has ended
        while ((st & (1u << 10)) && node == SENTINEL) {
            if (targ < a.n_targets) {
                const RtsTargetDev& TG = a.targets[targ++];
                if (TG.root < 0) continue;
                const dvec3 q = mk3(TG.cx - prev.x, TG.cy - prev.y, TG.cz - prev.z);          // bounding sphere of the placed target, as in rts_trace_unit
                const double qq = q.x*q.x + q.y*q.y + q.z*q.z, b = q.x*dir.x + q.y*dir.y + q.z*dir.z;
                if (qq > TG.r2) {
                    const double dd = dir.x*dir.x + dir.y*dir.y + dir.z*dir.z;
                    if (!(b > 0.0) || !(b*b >= (qq - TG.r2) * dd * 0.999999)) continue;
                }
                node = TG.root; sp = 1; s_stack[tid] = SENTINEL;
            } else {
                RtsRay S;
                S.dir = dir; S.prev = prev; S.rayLength = rayLength; S.power = power; S.doppler = doppler; S.refx = 1; S.refy = 1;
                S.reflDepth = st & 0xffu; S.refrDepth = 0; S.received = (int)(st >> 16) - 1; S.end = (st & (1u << 8)) != 0u; S.chain_start = (st & (1u << 9)) != 0u;
                const bool primary = S.chain_start;
                const float tmin = primary ? SCENE_EPS : SCENE_EPS_R;
                uint32_t pending = 0, refr_code0 = 0;
                const uint32_t slot = RTS_SLOT();
                const bool go_on = rts_shade<KEEP_ALL, false, false>(a, L_, tid, gtid, lane, slot, 0u, D, 0u, origin, primary, (st & (1u << 11)) != 0u, best_t, best_leaf, best_prim, tmin, S, pending, refr_code0);
                if (go_on) {
                    atomicAdd(&s_n[tid], RTS_SEG_ONE);
                    best_t = RTS_DEFAULT_TMAX; t_prune = RTS_DEFAULT_TMAX; best_leaf = -1; best_prim = 0xffffffffu;
                    targ = a.n_prims > 0 ? 0u : a.n_targets;
                } else {
                    rts_write_back<KEEP_ALL, false, false>(a, L_, tid, lane, slot, 0u, S, pending, refr_code0);
                }
                dir = S.dir; prev = S.prev; rayLength = S.rayLength; power = S.power; doppler = S.doppler;
                st = (S.reflDepth & 0xffu) | (S.end ? 1u << 8 : 0u) | (S.chain_start ? 1u << 9 : 0u) | (go_on ? 1u << 10 : 0u) | (st & (1u << 11)) | ((uint32_t)(S.received + 1) << 16);
            }
        }
        const uint32_t n_live = (uint32_t)__popcll(__ballot((st & (1u << 10)) != 0u));
        if (n_live == 0u) break;
        // ---------------------------------------------------------------- the ray in the space of the lane's target (every live lane walks now;
        // lanes that come back to a walk in progress recompute what they had: nothing of it is carried through the shading code)
        RtsSlabRay lr;
        {
            const RtsTargetDev& TG = a.targets[targ - 1u];
            const dvec3 q = mk3(prev.x - TG.px, prev.y - TG.py, prev.z - TG.pz);
            const dvec3 ol = mk3(TG.rinv[0]*q.x + TG.rinv[1]*q.y + TG.rinv[2]*q.z, TG.rinv[3]*q.x + TG.rinv[4]*q.y + TG.rinv[5]*q.z, TG.rinv[6]*q.x + TG.rinv[7]*q.y + TG.rinv[8]*q.z);
            const dvec3 dl = mk3(TG.rinv[0]*dir.x + TG.rinv[1]*dir.y + TG.rinv[2]*dir.z, TG.rinv[3]*dir.x + TG.rinv[4]*dir.y + TG.rinv[5]*dir.z,
                                 TG.rinv[6]*dir.x + TG.rinv[7]*dir.y + TG.rinv[8]*dir.z);
            lr = rts_slab_setup(ol, dl, TG.ew);
        }
        const float tmin = (st & (1u << 9)) ? SCENE_EPS : SCENE_EPS_R;
        const uint32_t age = (uint32_t)RTS_COST_UNITS(wall_clock64() - tile_t0);                 // (s_memrealtime: wave-uniform)
        const uint32_t idle_limit = age >= a.async_age ? a.async_idle1 : a.async_idle0;
        // ---------------------------------------------------------------- walk
        for (;;) {
            if (node != SENTINEL) {
                rts_walk_step<COUNT>(a, s_stack, tid, gtid, lds_cap, &s_n[2 * RTS_BLOCK + tid], node, sp, lr, prev, dir, tmin, best_t, best_leaf, best_prim, t_prune,
                                     n_nodes, n_tris, hard_overflow);
            }
            if (++wave_steps > (1u << 26)) { hard_overflow = true; node = SENTINEL; }            // malformed tree guard (per wave and tile): every wave must drain
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(node != SENTINEL));
            if (COUNT) { lane_stats[0] += 64u; lane_stats[1] += n_live; lane_stats[2] += n_walk; }     // (after the step: lanes that took it = n_walk + those that just finished; close enough for a ratio)
            if (n_walk == 0u || n_live - n_walk >= idle_limit) break;
        }
    }
#undef RTS_SLOT
}

// counters[1..6] = sum over the blocks of a launch (256 threads of ONE block; the launch has at most a few thousand blocks)
// ... and all of them into the handle's pinned host block (host_cnt: device address of RtsPinned::cnt) -- the host reads them
// after its wait for the stream, without a copy of their own.  s: 256 u64 of LDS.  Loads at agent scope (see k_trace's epilogue).
__device__ __forceinline__ void rts_sum_counters_body(const uint32_t t, unsigned long long* s, const unsigned long long* block_counters, unsigned int n_blocks,
                                                      unsigned long long* counters, const uint32_t* head_count, unsigned long long* host_cnt)
{
#define RTS_LD64(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
    if (t == 7) { const unsigned long long h = head_count ? (unsigned long long)__hip_atomic_load(head_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ULL; counters[7] = h; host_cnt[7] = h; }     // the order's head count travels home with the counters (sizes the next cooperative grid)
    if (t == 0) host_cnt[0] = RTS_LD64(&counters[0]);                          // received rays (appended by the trace kernels)
    if ((t >= 8 && t <= 12) || t == 14 || t == 15) host_cnt[t] = RTS_LD64(&counters[t]);      // lane statistics and walked segments of the counting build; [12] cost records dropped, [14] tiles (| XCC mask << 56) on which the shader clock ran backwards, [15] never-started lane-steps
    const unsigned int k = t & 7u, lane = t >> 3;                              // 32 partial sums per counter
    unsigned long long v = 0;
    // (counting builds poison the rows before the launch -- rts_trace_launch -- : a row no block wrote is COUNTED, counters[13], instead
    // of summed; round 3 saw such launches -- a garbage sum in counters[6] -- and could not reproduce them: this names them if they return)
    unsigned long long poisoned = 0;
    if (k >= 1 && k <= 6) for (unsigned int b = lane; b < n_blocks; b += 32) { const unsigned long long x = RTS_LD64(&block_counters[(size_t)b * 8 + k]); if (x == ~0ULL) poisoned++; else v += x; }
    s[t] = v;
    if (k == 1 && poisoned) { atomicAdd(&counters[13], poisoned); __threadfence(); }      // (the rows' first counter; never in a healthy launch)
    __syncthreads();
    if (t >= 1 && t <= 6) {
        unsigned long long sum = 0;
        for (unsigned int l = 0; l < 32; l++) sum += s[l * 8 + t];
        counters[t] = sum; host_cnt[t] = sum;
    }
    if (t == 13) host_cnt[13] = RTS_LD64(&counters[13]);                       // rows of block_counters still poisoned (counting builds)
#undef RTS_LD64
}

// KEEP_ALL is a template parameter, not a run-time flag: hipcc (ROCm 7.2) lowered the uniform
// `if (a.keep_all)` to a per-lane v_cmp mask computed under the divergent exec of the bounce loop
// and re-used it at the write-back under a different exec, so lanes that were inactive at the
// definition stored through the null all_records pointer.
// REFR builds the refraction branch of closest_hit (normal_shader.cu:191-282): a launch index then owns up
// to three independent ray chains -- 0: the reflection chain, 1: the ray refracted INTO the first-hit
// target (spawned by chain 0 at its first hit), 2: the ray refracted back OUT (spawned by chain 1 at its
// first hit) -- whose results live in rows rayIndex + k*W^3 of the output buffers (:214, :272-279).
// The reference recurses depth first; the chains are independent once spawned (the payload is copied,
// :191), so they are traced one after the other and the spawned state is parked in global memory.
// COOP: the kernel of the cooperative units (launched beside the ordinary one, on its own stream, when the handle has a cost
// history): it traces the 64 n_head launch indices of the tiles at the head of the cost order, one per wave; the ordinary
// kernel then starts at position n_head of the order.  A kernel of its own because the shared walk needs ~40 registers more
// than the 128 the ordinary kernel is held to (four waves per SIMD).
template <bool COUNT, bool KEEP_ALL, bool REFR, bool COOP, bool ASYNC = false, bool AFFINE = false, bool VERS = false>
__global__ void __launch_bounds__(RTS_BLOCK, (REFR && COOP) ? 2 : ((REFR || COOP) ? 3 : 4)) k_trace(const RtsTraceArgs a)
{
    __shared__ __attribute__((aligned(16))) int32_t s_stack[RTS_STACK_LDS * RTS_BLOCK];
    const uint32_t tid = threadIdx.x;
    // (the COOP kernel owns the slabs behind the ordinary kernel's: per-thread spill / child rows, draw counters, block counters)
    const uint32_t gtid = (COOP ? a.total_threads : 0u) + blockIdx.x * RTS_BLOCK + tid;
    // Launch constants and the first RTS_RX_LDS receivers are copied into LDS once per (persistent) block.  Read through their
    // device pointers they are wave-wide BROADCAST loads on the vector memory path -- which returns 64 x 16 bytes per dwordx4
    // whether the lanes' addresses differ or not -- and they are re-issued for every tile and every missing segment (the
    // fetch asm clobbers memory, so nothing read through a pointer stays in registers): on a launch that hits nothing they
    // alone kept the return path 87 % busy (5.3 M loads, counters of tools/trace_bench.py c3nomesh).
    __shared__ __attribute__((aligned(16))) RtsLaunchConsts s_lc;
    __shared__ __attribute__((aligned(16))) RtsRxDev s_rx[RTS_RX_LDS];
    // Payload that the traversal loop does not touch lives in LDS, entry-major like the stack (lane `tid` owns element
    // k * RTS_BLOCK + tid: conflict-free 8-byte accesses): the first hit point, the two path words and the per-lane
    // counters -- 13 dwords per lane that the register allocator otherwise carried through the walk (128-VGPR budget at
    // four waves per SIMD) by spilling to scratch.  They are touched per SHADED hit and at write-back only.
    __shared__ __attribute__((aligned(16))) double s_first[3 * RTS_BLOCK];
    __shared__ __attribute__((aligned(16))) unsigned long long s_path[2 * RTS_BLOCK];
    __shared__ uint32_t s_n[3 * RTS_BLOCK];                      // segments, shaded hits, stack entries spilled (per lane)
    s_n[threadIdx.x] = 0; s_n[RTS_BLOCK + threadIdx.x] = 0; s_n[2 * RTS_BLOCK + threadIdx.x] = 0;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.lc); uint32_t* dst = reinterpret_cast<uint32_t*>(&s_lc);
        for (uint32_t i = threadIdx.x; i < sizeof(RtsLaunchConsts) / 4; i += RTS_BLOCK) dst[i] = src[i];
        const uint32_t n_rx_lds = a.n_rx < RTS_RX_LDS ? a.n_rx : RTS_RX_LDS;
        const uint32_t* rsrc = reinterpret_cast<const uint32_t*>(a.rx); uint32_t* rdst = reinterpret_cast<uint32_t*>(s_rx);
        for (uint32_t i = threadIdx.x; i < n_rx_lds * (sizeof(RtsRxDev) / 4); i += RTS_BLOCK) rdst[i] = rsrc[i];
    }
    __syncthreads();
    const RtsLaunchConsts& lc = s_lc;
    const dvec3 origin = mk3(lc.ox, lc.oy, lc.oz);
    // pre-filter constants of the receivers, for rays that start at the transmitter: q = centre - origin (f64, then f32),
    // |q|, |q|^2 and the widened radius^2.  Widening: the f32 direction is good to ~2e-6 rad and the f32 discriminant
    // b^2 - (|q|^2 - r^2) |d|^2 to ~1e-6 |q|^2; r'^2 = r^2 (1 + 1e-3) + 1e-5 |q|^2 + 1e-6 covers both ten times over.
    // The filter has to be conservative with respect to the REFERENCE'S arithmetic, not to geometry: its quadratic forms
    // C = |prev|^2 + |c|^2 - 2 c.prev - r^2 from world-scale terms (ray_tracer.cu:285), which at Earth-centred coordinates
    // cancel catastrophically -- terms of 4e13 m^2, rounded a dozen times: the sphere it tests is up to ~0.05 m^2 larger or
    // smaller in r^2 than the one it was given (found by tools/fuzz_equal.py: a 0.86 m sphere 3.7 m from the transmitter
    // captured 32 rays the geometric filter had excluded).  + 1e-14 (|o|^2 + |c|^2): 45 ulps of the largest term.
    __shared__ float s_rxp[RTS_RX_LDS][RTS_RXP_N];
    if (tid < RTS_RX_LDS && tid < a.n_rx) {
        const RtsRxDev r = s_rx[tid];
        const double qx = r.cx - lc.ox, qy = r.cy - lc.oy, qz = r.cz - lc.oz, qq = qx*qx + qy*qy + qz*qz;
        const double w2 = lc.ox*lc.ox + lc.oy*lc.oy + lc.oz*lc.oz + r.cx*r.cx + r.cy*r.cy + r.cz*r.cz;
        const double r2wd = r.radius * r.radius * 1.001 + 1.0e-5 * qq + 1.0e-6 + 1.0e-14 * w2;
        s_rxp[tid][0] = (float)qx; s_rxp[tid][1] = (float)qy; s_rxp[tid][2] = (float)qz; s_rxp[tid][RTS_RXP_QQW] = (float)qq * 1.000001f;
        s_rxp[tid][RTS_RXP_R2W] = (float)r2wd;
        s_rxp[tid][RTS_RXP_QN] = (float)sqrt(qq);
        // the window screen's constants (rts_rx_maybe): window = (minTheta, maxTheta) x (minPhi, maxPhi), ray_tracer.cu:343-375; applicable when it does not
        // reach over a pole (the reference then tests a second, mirrored window) and spans < 3 rad in azimuth (angle_in_range rejects spans >= pi)
        const double th_c = 0.5 * (r.minTheta + r.maxTheta), th_h = 0.5 * (r.maxTheta - r.minTheta);
        const bool ok = r.minPhi >= -RTS_PI / 2 && r.maxPhi <= RTS_PI / 2 && r.minPhi < r.maxPhi && th_h > 0.0 && th_h < 1.5 && r.radius > 0.0 && isfinite(th_c) && isfinite(qq);
        s_rxp[tid][RTS_RXP_QQ] = (float)qq; s_rxp[tid][RTS_RXP_R2] = (float)(r.radius * r.radius);
        s_rxp[tid][RTS_RXP_E2] = (float)(2.0e-6 * qq + 1.0e-6 * r.radius * r.radius + 1.0e-14 * w2 + 1.0e-12);
        s_rxp[tid][RTS_RXP_CT] = (float)cos(th_c); s_rxp[tid][RTS_RXP_ST] = (float)sin(th_c); s_rxp[tid][RTS_RXP_CH] = (float)cos(th_h);
        s_rxp[tid][RTS_RXP_SLO] = (float)(r.radius * sin(r.minPhi)); s_rxp[tid][RTS_RXP_SHI] = (float)(r.radius * sin(r.maxPhi));
        s_rxp[tid][RTS_RXP_OK] = (ok && a.rx_window_screen) ? 1.0f : 0.0f;
        s_rxp[tid][RTS_RXP_RW] = (float)(sqrt(r2wd) * 1.000001);
    }
    __syncthreads();
    uint32_t n_nodes = 0, n_tris = 0;                            // counting build only (per lane and launch: far below 2^32)
    bool hard_overflow = false;
    const bool mask_on = a.pmask != nullptr && lc.mask.n != 0 && a.pmask[(size_t)lc.mask.n * lc.mask.n / 32u] == 0u;     // (uniform) not voided by k_primary_mask
    // the pre-filter can only pay if it can clear a ray of the targets: with geometry but no valid mask every primary needs
    // its exact direction anyway
    const bool pre_on = lc.W > 1 && a.n_rx <= RTS_RX_LDS && (mask_on || a.n_prims == 0) && a.pre_filter;
    const uint32_t max_refr = REFR ? 2u : 0u;
    const uint32_t D = a.max_refl + max_refr;

    unsigned long long tl_t0 = 0;
    if (!COOP && a.timeline && tid == 0) { tl_t0 = wall_clock64(); a.timeline[(size_t)blockIdx.x * 2] = tl_t0; }      // (product builds: RTS_TIMELINE_BLOCKS, the blocks' ticks only)
    // Work units are WAVE TILES of 64 consecutive launch indices, taken by the waves one at a time from RTS_TILE_CTRS
    // striped counters: wave w draws k = atomicAdd(ctr[w % C]) and traces position k*C + (w % C) of the tile order.  Tile
    // durations are extremely skewed (median ~2.5 us: every ray misses; 99.9th percentile ~0.3 ms; a handful near 0.9 ms whose
    // rays bounce six times through a hundred steps each), so an in-order sweep left a ~1 ms tail in which a few waves finished
    // their slow tiles on an otherwise idle GPU.  Pulses of an interval look alike, so every launch records what each tile cost
    // (tile_cost, shader clocks >> 6) and the next launches of the handle trace the tiles in descending order of the cost
    // last seen (tile_order, built by rts_tile_order_build): longest-processing-time-first list scheduling.  Striping the
    // counter -- ONE 128-byte line per stripe (RTS_TILE_CTR_STRIDE): atomics on one line serialise in L2 at ~10 ns each
    // whatever their address -- keeps the draws off the critical path: 69 k of them per launch over 64 lines.
    const uint32_t n_tiles = (a.n_rays + 63u) / 64u;
    const uint32_t lane = tid & 63u;
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the wave's number in the block)
    const uint32_t stripe = __builtin_amdgcn_readfirstlane((blockIdx.x * (RTS_BLOCK / 64u) + (tid >> 6)) % RTS_TILE_CTRS);   // wave-uniform: the queue arithmetic below stays scalar
    // Draw schedule of a stripe (positions k*C + stripe, k = 0, 1, ...): the first quarter -- the expensive end of the
    // order -- one tile per draw, the cheap rest four tiles per draw (a miss-only tile is ~2 us of work, and so is a draw's
    // latency).  In the cheap part the next draw is issued before the current tiles are traced, so its latency hides behind them.
    // COOPERATIVE UNITS: the first n_head tiles of the cost order (rts_tile_order_build: those whose last cost exceeds a
    // fraction of the launch's balanced time) are traced as 64 units of ONE launch index each, all 64 lanes of a wave walking
    // that ray's hierarchy together (rts_trace_unit<.., COOP = true>) -- by the COOP kernel, whose unit v is
    // (tile_order[v / 64], ray v % 64); unit v of the ordinary kernel is tile_order[n_head + v].
    const uint32_t n_head = (a.tile_head && a.tile_order) ? min(min(__builtin_amdgcn_readfirstlane(a.tile_head[0]), n_tiles), 16384u) : 0u;
    constexpr uint32_t CG = (COOP && VERS) ? (uint32_t)RTS_COOP_GROUP : 64u;      // lanes per ray of a cooperative unit; a unit holds 64 / CG rays, a head tile is CG units (rts_walk_coop)
    constexpr uint32_t RPU = 64u / CG;
    const uint32_t n_units = COOP ? CG * n_head : n_tiles - n_head;
    __shared__ int32_t s_exch[COOP ? RTS_BLOCK : 1];             // exchange rows of the cooperative walk (one 64-entry row per wave)
    __shared__ uint32_t s_lane_scratch[COUNT ? 2 * (RTS_BLOCK / 64) : 1];      // (counting build: per-wave max / sum of a round's walk steps)
    __shared__ uint32_t s_walk[ASYNC ? 1 : 2 * (RTS_BLOCK / 64)];           // per wave: walk iterations of the current tile (each walk's slowest lane), walks
    const RtsUnitLds ul = {s_stack, s_exch, s_first, s_path, s_n, s_rx, s_rxp, s_lane_scratch, s_walk};
    unsigned long long lane_stats[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
    // XCD-AFFINE SUB-ORDERS (a.xcd_seg, big launches; ordinary kernel only): the order behind the head is cut into one segment per XCD
    // -- a band of the lattice that held an eighth of the cost last seen, longest tiles first inside it -- so that the ~500 waves of
    // an XCD trace neighbouring tiles at the same time and its 4 MB of L2 serves ONE part of a scene that is a hundred times that
    // size (BASELINE configs[3]: every XCD used to stream the whole scene, 13.5 GB of fabric traffic per launch at an L2 hit rate
    // of 0.66).  Segment 0 = what is left of the head when no cooperative kernel runs, segments 1 .. 8 = the bands; each has
    // RTS_SEG_STRIPES draw counters.  A wave starts on segment 0, goes on with ITS XCD's band (HW_REG_XCC_ID) and, when that runs
    // dry, with the next bands in turn -- every wave visits every segment, so every position is drawn whoever sits where.
    // (a kernel of its own -- AFFINE -- whose queue state lives in LDS, one row per wave: as five more scalars of the ordinary kernel
    // they pushed it over its 128 registers, 14 VGPRs into scratch)
    static_assert(!AFFINE || (!COOP && !ASYNC), "affine sub-orders: ordinary lock-step kernel only");
    __shared__ uint32_t s_seg[AFFINE ? RTS_XCD + 2 : 1];         // s_seg[i] .. s_seg[i + 1]: order positions of segment i
    __shared__ uint32_t s_q[AFFINE ? (RTS_BLOCK / 64) * 8 : 1];  // per wave: [0] segment, [1] its first unit, [2] its length, [3] positions per stripe, [4] single draws, [5] segments visited
    // (LDS addresses formed from the scalar wave number AT EACH USE: hoisted out of the tile loop as vector registers they were
    // parked in scratch and reloaded eight times per draw)
#define RTS_OPAQUE_S(x) ({ uint32_t o_ = (x); asm volatile("" : "+s"(o_)); o_; })
#define RTS_Q(k) s_q[RTS_OPAQUE_S(wave_u) * 8u + (k)]
    if (AFFINE) {
        if (tid <= RTS_XCD) s_seg[tid + 1u] = max(min(a.xcd_seg[tid], n_tiles), n_head);
        if (tid == 0) s_seg[0] = n_head;
        __syncthreads();
        if (lane == 0) {
            const uint32_t len0 = s_seg[1] - s_seg[0], ps0 = (len0 + RTS_SEG_STRIPES - 1u) / RTS_SEG_STRIPES;
            RTS_Q(0) = 0u; RTS_Q(1) = s_seg[0] - n_head; RTS_Q(2) = len0; RTS_Q(3) = ps0; RTS_Q(4) = ps0 >= 128u ? ps0 / 4u : ps0; RTS_Q(5) = 0u;
        }
    }
    // THE COOPERATIVE KERNEL keeps a head tile's 64 units on ONE XCD: head tile h belongs to the list of XCD h mod 8 (the head is
    // sorted by cost, so dealing it out in turn balances the lists), each list drawn through 8 counters; a wave works through its
    // own XCD's list first and then through the others'.  The 64 rays of a tile are 64 nearly parallel rays along one radial line
    // of the lattice: they walk the same few hundred KB of records, thousands of steps each -- spread over all eight XCDs (the
    // striped queue dealt unit v to stripe v mod 64) every L2 fetched every head tile's records: 5.4 of the 6.4 GB a BASELINE
    // configs[3] launch moves over the fabric, at an L2 hit rate of 0.64 (profiles/r04_c4_xcd_affine_pmc.log).
    // a.coop_spread = P in {1, 2, 4, 8}: a head tile's 64 units are dealt to P of the eight lists (rays r mod P = c to list
    // (h + (8 / P) c) mod 8); P = 1: the whole tile on one XCD, P = 8: eight rays on each (every L2 sees every tile).
    uint32_t coop_x = 0, coop_sweep = 0, coop_len = 0;           // (COOP) current list, lists visited, head tiles that feed the list
    const uint32_t coop_P = COOP ? a.coop_spread : 1u, coop_G = RTS_XCD / coop_P;
#define RTS_COOP_LEN(x) (n_head > ((x) % coop_G) ? (n_head - ((x) % coop_G) + coop_G - 1u) / coop_G : 0u)
    if (COOP) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(coop_x));
        coop_x &= (RTS_XCD - 1u);
        coop_len = RTS_COOP_LEN(coop_x);
    }
    const uint32_t S = (AFFINE || COOP) ? (uint32_t)RTS_SEG_STRIPES : (uint32_t)RTS_TILE_CTRS;      // stripes of a segment
#define RTS_LSTRIPE (AFFINE ? (RTS_OPAQUE_S(stripe) & (RTS_SEG_STRIPES - 1u)) : (COOP ? (stripe & (RTS_SEG_STRIPES - 1u)) : RTS_OPAQUE_S(stripe)))      // (opaque: the draw counter's ADDRESS is formed at each draw -- hoisted, it is a vector register pair of the tile loop)
    const uint32_t per_stripe_all = (n_units + RTS_TILE_CTRS - 1u) / RTS_TILE_CTRS;
    // DEAD-TILE BATCHES (round 5).  Most wave tiles of a pulse are DEAD -- the pre-filter clears all 64 launch indices: 84 % of BASELINE
    // configs[2]'s tiles -- and the cost order keeps them at its end (cost record 1).  That part of the order is drawn 64 positions at
    // a time and every LANE asks the pre-filter's question of one whole tile (rts_tile_maybe: the tile's rays are a row segment of the
    // lattice, its two end rays bound what all of them can reach): the tiles that are still dead are accounted for by their lanes --
    // 64 segments counted, cost record 1 -- and the few that have come to life are traced by the wave one after the other, as any
    // tile is.  k_dead = the stripe's first position in the dead part (a.tile_live: written by the order build), 0 with
    // a.batch_dead == 2 (every position is screened that way first: tests).  Ordinary lock-step kernel of aligned launches only.
    // (the schedule's three numbers and a batch's mask of live positions live in LDS, one row per wave, and are read once per DRAW: as
    // scalars of the kernel they were live through every tile's bounce loops -- 30 more scalar spills, and the pending draw in scratch)
    constexpr bool BATCHABLE = !COOP && !ASYNC && !AFFINE && !KEEP_ALL;
    __shared__ uint32_t s_sched[(RTS_BLOCK / 64) * 8];      // per wave: [0] single draws, [1] draws of four, [2] k_dead, [3] [4] live mask of the batch in hand
#define RTS_SCHED(k) s_sched[RTS_OPAQUE_S(wave_u) * 8u + (k)]
    const uint32_t single_draws_nobatch = per_stripe_all >= 1024u ? per_stripe_all / 4u : per_stripe_all;      // (short queues: one tile per draw throughout)
    if (BATCHABLE) {
        uint32_t k_dead = per_stripe_all;
        if (a.batch_dead != 0u && pre_on) {
            if (a.batch_dead == 2u) k_dead = 0u;
            else if (a.tile_live && a.tile_order) {
                const uint32_t lv = __builtin_amdgcn_readfirstlane(a.tile_live[0]);          // 1 + tiles at the front of the order that cost more than a dead one (0: unknown)
                if (lv) { const uint32_t v_dead = lv - 1u > n_head ? lv - 1u - n_head : 0u; k_dead = min(per_stripe_all, v_dead > stripe ? (v_dead - stripe + RTS_TILE_CTRS - 1u) / RTS_TILE_CTRS : 0u); }
            }
        }
        // draws of a stripe: positions [0, s1) one per draw, [s1, k_dead) four per draw (n2 draws), [k_dead, ..) 64 per draw
        const uint32_t s1 = min(single_draws_nobatch, k_dead);
        if (lane == 0) { RTS_SCHED(0) = s1; RTS_SCHED(1) = (k_dead - s1 + 3u) / 4u; RTS_SCHED(2) = k_dead; }
    }
    const uint32_t single_draws_all = single_draws_nobatch;
    // The pending draw is held in a register of lane 0 (so that its latency hides behind the tiles traced meanwhile) -- except
    // in the counting builds, which are short of registers: there the allocator spilled it to scratch and launches lost whole
    // tiles' worth of counters from run to run, until it was moved to LDS.  What the ISA of that build shows (round 3,
    // commit 835ee13~1, k_trace<true,..>: `hipcc -S`, spill slot 8): the draw is stored by `scratch_store_dword off, v0, off
    // offset:8` under EXEC = lane 0, AFTER `s_waitcnt vmcnt(0)` on the atomic's return value, and reloaded at the loop top by
    // `scratch_load_dword v0, off, off offset:8` under the full mask, followed by `s_waitcnt vmcnt(0)` and
    // `v_readfirstlane_b32` (= lane 0, whose dword the store wrote): same per-lane address, program order, the wait before the
    // use -- the emitted code has NO ordering defect, so "the reload overtaking the store" (last round's suspicion) is not it.
    // The only other values in scratch there are the timeline tick (slot 0) and `tid` (slot 12: stored in the prologue,
    // reloaded in the epilogue for the `tid in 1..6` test that guards the block_counters store).  The symptom -- a garbage SUM in
    // counters[6], i.e. block_counters rows that no block had written -- fits a wrong `tid` there better than a wrong draw; it
    // was not reproduced this round (the builds differ: 32-bit counters came in with the same commit), so the cause stays
    // unproven.  Product builds reload nothing from scratch (tests/test_host_logic.py checks the ISA).
    __shared__ uint32_t s_draw[RTS_BLOCK / 64];
    uint32_t draw_next = 0;
#define RTS_DRAW() { const uint32_t dv_ = atomicAdd(&a.tile_ctr[(COOP ? RTS_OFF_CTR_COOP : 0) + ((AFFINE ? RTS_Q(0) * S : (COOP ? coop_x * S : 0u)) + RTS_LSTRIPE) * RTS_TILE_CTR_STRIDE], 1u); if (AFFINE) s_draw[RTS_OPAQUE_S(wave_u)] = dv_; else if (COUNT || KEEP_ALL) s_draw[wave_u] = dv_; else draw_next = dv_; }
    if (lane == 0) RTS_DRAW()
    for (;;) {
      const uint32_t draw = __builtin_amdgcn_readfirstlane(AFFINE ? s_draw[RTS_OPAQUE_S(wave_u)] : ((COUNT || KEEP_ALL) ? s_draw[wave_u] : draw_next));      // (KEEP_ALL builds -- tests -- keep it in LDS too: their refraction instantiation parked it in scratch inside the tile loop)      // (AFFINE: in LDS like the counting builds' -- the kernel has no register for it across the tile loop, see above)
      const uint32_t per_stripe = AFFINE ? __builtin_amdgcn_readfirstlane(RTS_Q(3)) : (COOP ? (coop_len * (CG / coop_P) + RTS_SEG_STRIPES - 1u) / RTS_SEG_STRIPES : per_stripe_all);
      const uint32_t single_draws = AFFINE ? __builtin_amdgcn_readfirstlane(RTS_Q(4)) : (COOP ? per_stripe : (BATCHABLE ? __builtin_amdgcn_readfirstlane(RTS_SCHED(0)) : single_draws_all));      // (COOP: one unit per draw -- units are long)
      const uint32_t four_draws = BATCHABLE ? __builtin_amdgcn_readfirstlane(RTS_SCHED(1)) : 0u, k_dead = BATCHABLE ? __builtin_amdgcn_readfirstlane(RTS_SCHED(2)) : 0u;
      const bool batch = BATCHABLE && draw >= single_draws + four_draws;      // (uniform) a batch of 64 positions of the order's dead part
      const uint32_t k0 = draw < single_draws ? draw : (batch ? k_dead + 64u * (draw - single_draws - four_draws) : single_draws + 4u * (draw - single_draws));
      const uint32_t kn = draw < single_draws ? 1u : (batch ? 64u : (BATCHABLE ? min(4u, k_dead - k0) : 4u));
      if (k0 >= per_stripe) {
          if (COOP) {                                                       // this XCD's list is empty: the next XCD's
              if (++coop_sweep >= RTS_XCD) break;
              coop_x = (coop_x + 1u) & (RTS_XCD - 1u);
              coop_len = RTS_COOP_LEN(coop_x);
              if (lane == 0) RTS_DRAW()
              continue;
          }
          if (!AFFINE) break;
          const uint32_t sweep = __builtin_amdgcn_readfirstlane(RTS_Q(5)) + 1u;
          if (sweep > RTS_XCD) break;
          if (lane == 0) {
              uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));      // (read where it is used: a scalar less to carry through the tile loop)
              const uint32_t si = 1u + ((xcc & (RTS_XCD - 1u)) + sweep - 1u) % RTS_XCD;       // own band first, then the others in turn
              const uint32_t len = s_seg[si + 1u] - s_seg[si], ps = (len + RTS_SEG_STRIPES - 1u) / RTS_SEG_STRIPES;
              RTS_Q(0) = si; RTS_Q(1) = s_seg[si] - n_head; RTS_Q(2) = len; RTS_Q(3) = ps; RTS_Q(4) = ps >= 128u ? ps / 4u : ps; RTS_Q(5) = sweep;
              RTS_DRAW()
          }
          continue;
      }
      const uint32_t seg_base = AFFINE ? __builtin_amdgcn_readfirstlane(RTS_Q(1)) : 0u, seg_len = AFFINE ? __builtin_amdgcn_readfirstlane(RTS_Q(2)) : (COOP ? coop_len * (CG / coop_P) : n_units);
      // (only in the cheap part of the order: a draw made before an EXPENSIVE tile would reserve the stripe's next most
      // expensive tile for as long as this one takes -- the launch then ends with that tile, traced alone)
      const bool ahead = draw >= single_draws;
      if (ahead && lane == 0) RTS_DRAW()
     // a batch: lane j screens the tile at position k0 + j; what is left of it are the positions whose tiles may have come to life
     if (BATCHABLE && batch) {
         const uint64_t vl = (uint64_t)(k0 + lane) * S + RTS_LSTRIPE;
         bool maybe = false;
         if (k0 + lane < per_stripe && vl < seg_len) {
             const uint32_t tp = (uint32_t)vl + n_head;
             const uint32_t tj = a.tile_order ? a.tile_order[tp] : ((tp & 1u) ? (n_tiles - 1u) / 2u + (tp + 1u) / 2u : (n_tiles - 1u) / 2u - tp / 2u);
             const uint32_t* pm_ = a.pmask; asm volatile("" : "+s"(pm_));      // (opaque: addresses derived from it stay inside the batch)
             maybe = rts_tile_maybe(lc, tj, a.n_rays, a.n_prims > 0, mask_on, pm_, a.n_rx, s_rxp);
             if (!maybe) {                                             // still dead: its launch indices' one segment each, and its cost record
                 atomicAdd(&s_n[tid], min(64u, a.n_rays - tj * 64u));
                 if (a.tile_cost) a.tile_cost[tj] = 1u;
             }
         }
         const unsigned long long live_m = __ballot(maybe);
         if (lane == 0) { RTS_SCHED(3) = (uint32_t)live_m; RTS_SCHED(4) = (uint32_t)(live_m >> 32); }
     }
     for (uint32_t kb = 0; kb < kn; kb++) {
      uint32_t kq = kb;
      if (BATCHABLE && kn == 64u) {                                    // (a batch: the next position whose tile may have come to life)
          const uint32_t m_lo = __builtin_amdgcn_readfirstlane(RTS_SCHED(3)), m_hi = __builtin_amdgcn_readfirstlane(RTS_SCHED(4));
          if ((m_lo | m_hi) == 0u) break;
          kq = m_lo ? (uint32_t)__builtin_ctz(m_lo) : 32u + (uint32_t)__builtin_ctz(m_hi);
          if (lane == 0) { if (m_lo) RTS_SCHED(3) = m_lo & (m_lo - 1u); else RTS_SCHED(4) = m_hi & (m_hi - 1u); }
      }
      const uint64_t vloc64 = (uint64_t)(k0 + kq) * S + RTS_LSTRIPE;      // position inside the segment
      if (vloc64 >= seg_len) break;
      // (COOP: unit u of list x: head tile h = x mod G + G (u / (64 / P)), ray c + P (u mod (64 / P)) with c = ((x - h) mod 8) / G;
      // vpos = 64 x head tile + ray, as before)
      uint32_t vpos = seg_base + (uint32_t)vloc64;
      if (COOP) {
          const uint32_t per = CG / coop_P, j = (uint32_t)vloc64 / per, i = (uint32_t)vloc64 - j * per, h = (coop_x % coop_G) + coop_G * j;      // (CG units per head tile: unit q holds its rays RPU q .. RPU q + RPU - 1)
          vpos = (h << 6) | ((((coop_x - h) & (RTS_XCD - 1u)) / coop_G) + coop_P * i);
      }
      const bool coop_unit = COOP;
      const uint32_t tpos = COOP ? (vpos >> 6) : vpos + n_head;
      // no history yet (first launch of the handle): centre-out over the launch range -- the beam is normally centred on
      // the targets, so the expensive tiles sit in the middle of the lattice and should be started first
      const uint32_t tile = a.tile_order ? a.tile_order[tpos] : ((tpos & 1u) ? (n_tiles - 1u) / 2u + (tpos + 1u) / 2u : (n_tiles - 1u) / 2u - tpos / 2u);
      const uint32_t slot = coop_unit ? tile * 64u + (vpos & 63u) * RPU + lane / CG : tile * 64u + lane;
      // A DEAD tile -- the pre-filter clears every one of its launch indices (most tiles of a pulse: the beam is wider than the
      // targets) -- ends here: its launch indices' one segment each is counted, nothing else of the per-tile machinery runs
      // (payload initialisation, the bounce loop's tests, clocks, the cost record's arithmetic: half of the ~190 vector and ~90 scalar
      // instructions such a tile cost, 10 % of a BASELINE configs[2] launch's instructions)
      uint32_t pre = 0u;
      if (!COOP && !ASYNC && !KEEP_ALL && pre_on) {
          bool pre_target = a.n_prims > 0, pre_rx = a.n_rx > 0;
          if (slot < a.n_rays) rts_prefilter(lc, slot, mask_on, a.pmask, a.n_rx, s_rxp, pre_target, pre_rx);
          if (!__any(slot < a.n_rays && (pre_target || pre_rx))) {
              if (slot < a.n_rays) atomicAdd(&s_n[tid], 1u);       // (bits 0-21: the lane's segments of the launch)
              if (lane == 0 && a.tile_cost) a.tile_cost[tile] = 1u;
              continue;
          }
          pre = 1u | (pre_target ? 2u : 0u) | (pre_rx ? 4u : 0u);
      }
      // Tile durations on the CONSTANT-RATE counter (wall_clock64 = s_memrealtime, 100 MHz), scaled to the old unit -- 64 shader clocks at
      // 2.4 GHz = 26.7 ns = 8/3 ticks -- so that every threshold kept its meaning (RTS_COST_UNITS).  Rounds 2-4 timed tiles with clock64()
      // (s_memtime, the shader clock counter), and on gfx950 that counter does not only tick: with three BASELINE configs[3] pulses in flight
      // about one launch in fifteen had ~1 000 tiles -- all those open at one instant -- whose end read EARLIER than their start
      // (profiles/r04_c4_cost_glitch.log).  The counting build still reads both and counts the tiles on which the shader clock ran
      // backwards while the constant-rate one did not, with the XCCs they ran on (counters[14]; RTS_DEBUG_COOP).
      const long long tile_t0 = wall_clock64();
      const long long tile_s0 = COUNT ? clock64() : 0;
      if (!COOP) atomicAnd(&s_n[tid], 0x003fffffu);              // (ds_and_b32: the tile's own segment count starts at zero; a register for it would be the 129th)
      if (!COOP && !ASYNC && lane == 0) { const uint32_t z_ = RTS_OPAQUE_S(0u); s_walk[2u * wave_u] = z_; s_walk[2u * wave_u + 1u] = z_; }      // (an opaque zero: hoisted out of the tile loop as a register pair the constant was parked in SCRATCH and reloaded per tile)
      const unsigned long long tl_tile = (COUNT && a.timeline && lane == 0) ? wall_clock64() : 0ULL;
      if (slot < a.n_rays) {
          if (ASYNC) rts_trace_unit_async<COUNT, KEEP_ALL>(a, lc, ul, tid, gtid, lane, slot, pre_on, mask_on, D, origin, tile_t0, n_nodes, n_tris, hard_overflow, lane_stats);
          else rts_trace_unit<COUNT, KEEP_ALL, REFR, COOP, VERS>(a, lc, ul, tid, gtid, lane, slot, pre_on, mask_on, D, max_refr, origin, n_nodes, n_tris, hard_overflow, lane_stats, pre);
      }   // slot < n_rays
      // (the segment count of the tile is only formed for tiles long enough to matter: an all-miss tile is ~100 instructions)
      const unsigned long long dt = RTS_COST_UNITS(wall_clock64() - tile_t0);                     // (s_memrealtime: wave-uniform)
      if (COUNT && !COOP && lane == 0 && clock64() - tile_s0 < 0) {                               // the shader clock ran backwards over this tile
          uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
          atomicAdd(&a.counters[14], 1ULL); atomicOr(&a.counters[14], 1ULL << (56u + (xcc & 7u)));      // (count in the low bits, the XCCs' mask in bits 56-63)
      }
      bool long_walks = false, longish_walks = false;
      if (!COOP && !ASYNC && a.tile_cost && dt >= a.coop_min_cost) {      // (the asynchronous-bounce experiment keeps no walk statistics: it flags nothing)
          // LONG WALKS: the tile's bounce rounds took a.coop_walk_steps walk iterations of the wave on average -- rays that graze along a surface
          // through thousands of boxes (BASELINE configs[3]: ~4 000 per segment; an ordinary tile's walks: 20-150).  Counted, not
          // timed: a duration per segment (rounds 2-3) had to be judged against the launch's mean, and a launch that consists of its
          // tail -- one GPU's interleaved eighth of a configs[3] pulse -- raised that mean until nothing was flagged (r04: 5-12 ms per
          // eighth instead of 1.5).
          const uint32_t wsteps = __builtin_amdgcn_readfirstlane(s_walk[2u * wave_u]), walks = __builtin_amdgcn_readfirstlane(s_walk[2u * wave_u + 1u]);      // (walks: bounce rounds)
          long_walks = walks != 0u && (unsigned long long)wsteps >= (unsigned long long)a.coop_walk_steps * walks;
          longish_walks = walks != 0u && (unsigned long long)wsteps >= (unsigned long long)a.coop_walk_steps_lo * walks;
      }
      if (lane == 0) {
          // Cost record of the tile: bits 0-29 its duration (shader clocks >> 6, + 1), bit 31 "LONG WALKS": its bounce rounds took
          // coop_walk_steps walk iterations on average (bit 30: coop_walk_steps_lo) -- rays that walk thousands of steps per segment,
          // the only kind whose walk is long enough to be worth sharing out between 64 lanes (the per-ray arithmetic outside
          // the walk is executed by a whole wave for ONE ray in a cooperative unit: a tile of short walks and many bounces
          // costs ten times its ordinary wave time that way; BASELINE configs[2]'s slowest tiles are of that kind).
          // (a tile traced as cooperative units costs the SUM of its units' wave time -- at least what it costs one wave -- and
          // keeps its flag, so a tile once at the head stays there: no flip-flopping between the two modes from launch to launch)
          // (a duration of 2^26 x 64 clocks -- half an hour -- is no duration: the shader clock read back a value from BEFORE the tile's
          // start.  Seen on gfx950 with three pulses in flight: ~1 000 tiles of ONE launch in fifteen, all at once, came out "negative";
          // clamped to the field's maximum they made the launch's cost sum 1 000 x too large, the head rule of the handle's next order
          // found no tile above half THAT balanced time, and a BASELINE configs[3] pulse went without its cooperative kernel: 13 ms
          // instead of 8.5, profiles/r04_c4_cost_glitch.log.  Such a tile leaves no record; the history keeps what it had.)
          if (a.tile_cost && dt >= (1ULL << 26)) atomicAdd(&a.counters[12], 1ULL);      // (cannot happen on the constant-rate counter; counted -- RtsStats::cost_records_dropped -- instead of assumed)
          if (a.tile_cost && dt < (1ULL << 26)) {
              if (COOP) { atomicAdd(&a.tile_cost[tile], ((unsigned int)(dt > 0x00fffffeULL ? 0x00fffffeULL : dt) + 1u) * RPU); if ((vpos & 63u) == 0u) atomicOr(&a.tile_cost[tile], 0x80000000u); }      // (x RPU: a unit of several rays stands for as many one-ray units in the head rule's sums)
              else a.tile_cost[tile] = ((unsigned int)(dt > 0x3ffffffeULL ? 0x3ffffffeULL : dt) + 1u) | (long_walks ? 0x80000000u : 0u) | (longish_walks ? 0x40000000u : 0u);
          }
          if (COUNT && a.timeline && !coop_unit) { a.timeline[(size_t)gridDim.x * 2 + tile] = wall_clock64() - tl_tile; a.timeline[(size_t)gridDim.x * 2 + n_tiles + tile] = tl_tile; }   // debug timeline (RTS_TIMELINE): duration, start tick
      }
     }   // tiles of the draw
      if (!ahead && lane == 0) RTS_DRAW()
    }
    if (!COOP && a.timeline && tid == 0) a.timeline[(size_t)blockIdx.x * 2 + 1] = wall_clock64();

    // ------------------------------------------------------------------ counters: wave reduce -> block reduce (LDS, the
    // traversal stack is dead by now) -> one plain store per block; k_sum_counters adds the blocks up.  (One atomic per
    // wave on the same two addresses -- 32 k same-line L2 atomics -- cost a fixed ~0.35 ms at the tail of every launch.)
    if (COUNT && !COOP && lane == 0 && lane_stats[0]) { atomicAdd(&a.counters[8], lane_stats[0]); atomicAdd(&a.counters[9], lane_stats[1]); atomicAdd(&a.counters[10], lane_stats[2]); }
    if (COUNT && (lane & (CG - 1u)) == 0u && lane_stats[3]) atomicAdd(&a.counters[11], lane_stats[3]);
    if (COUNT && !COOP && lane == 0 && lane_stats[4]) atomicAdd(&a.counters[15], lane_stats[4]);
    // (the thread index is re-formed here from the wave's number -- scalar, kept since the start -- and the lane number instead of
    // being carried through the kernel: the allocator, at its 128-register limit, otherwise parks threadIdx.x in scratch in the
    // prologue and reloads it here)
    const uint32_t tid_e = (wave_u << 6) | __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned long long n_seg = s_n[tid_e] & 0x003fffffu, n_shaded = s_n[RTS_BLOCK + tid_e], n_spill = s_n[2 * RTS_BLOCK + tid_e], n_nodes_w = n_nodes, n_tris_w = n_tris;
    for (int off = 32; off > 0; off >>= 1) {
        n_seg += __shfl_down(n_seg, off); n_shaded += __shfl_down(n_shaded, off);
        if (COUNT) { n_nodes_w += __shfl_down(n_nodes_w, off); n_tris_w += __shfl_down(n_tris_w, off); }
        n_spill += __shfl_down(n_spill, off);
    }
    __syncthreads();
    unsigned long long* s_cnt = reinterpret_cast<unsigned long long*>(s_stack);        // [waves][8]
    const int wave = (int)wave_u;
    if ((tid_e & 63) == 0) {
        s_cnt[wave * 8 + 1] = n_seg; s_cnt[wave * 8 + 2] = n_shaded; s_cnt[wave * 8 + 3] = COUNT ? n_nodes_w : 0ULL;
        s_cnt[wave * 8 + 4] = COUNT ? n_tris_w : 0ULL; s_cnt[wave * 8 + 5] = n_spill;
    }
    const bool any_overflow = __syncthreads_or(hard_overflow ? 1 : 0) != 0;
    if (tid_e >= 1 && tid_e <= 6) {
        unsigned long long v = 0;
        if (tid_e <= 5) { for (int w = 0; w < RTS_BLOCK / 64; w++) v += s_cnt[w * 8 + tid_e]; }
        else v = any_overflow ? 1ULL : 0ULL;
        a.block_counters[((size_t)(COOP ? a.total_threads / RTS_BLOCK : 0u) + blockIdx.x) * 8 + tid_e] = v;
    }
    // The LAST block of the launch -- of either kernel -- to get here adds the blocks' rows up and writes the launch's counters
    // into the handle's pinned host block: no kernel of its own between the end of the trace and the host's wake-up (it ran
    // 20-35 us among the neighbouring pulses' blocks).  Release / acquire around the ticket: every thread fences after its
    // stores, the last block fences before its loads, which bypass the non-coherent caches (the head count a few words away
    // was read by every block when it started).
    if (a.done_ctr) {
        __threadfence();
        __syncthreads();
        uint32_t* s_ticket = reinterpret_cast<uint32_t*>(s_stack) + 1024;                 // (beyond the [waves][8] sums above)
        if (tid_e == 0) *s_ticket = atomicAdd(a.done_ctr, 1u);
        __syncthreads();
        if (*s_ticket == a.n_blocks_all - 1u) {
            __threadfence();
            rts_sum_counters_body(tid_e, reinterpret_cast<unsigned long long*>(s_stack) + 1024, a.block_counters, a.n_blocks_all, a.counters, a.tile_head_all, a.host_cnt);
        }
    }
}

// k_sum_counters for a launch without launch indices (an interleaved part that is empty): zeros go home
__global__ void k_sum_counters(const unsigned long long* __restrict__ block_counters, unsigned int n_blocks, unsigned long long* __restrict__ counters, const uint32_t* __restrict__ head_count,
                               unsigned long long* __restrict__ host_cnt)
{
    __shared__ unsigned long long s[256];
    rts_sum_counters_body(threadIdx.x, s, block_counters, n_blocks, counters, head_count, host_cnt);
}

template <bool COOP>
static void rts_trace_dispatch(const RtsTraceArgs& a, bool count_traversal, unsigned grid, hipStream_t st)
{
    const int sel = (a.max_refr ? 4 : 0) | (a.keep_all ? 2 : 0) | (count_traversal ? 1 : 0);
    if (!COOP && a.xcd_seg && a.tile_order && !a.async_idle0 && sel < 2) {      // XCD-affine sub-orders: product and counting build of the plain reflection chain (every other
        if (sel == 0) k_trace<false, false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a);      // build traces the same order without the segments)
        else k_trace<true, false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a);
        return;
    }
    if (!COOP && a.async_idle0 && sel < 4) {                      // asynchronous bounces (rts_trace_unit_async): ordinary kernel, no refraction chains
        switch (sel) {
            case 0: k_trace<false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 1: k_trace<true, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 2: k_trace<false, true, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            default: k_trace<true, true, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        }
        return;
    }
    if (!COOP && a.nodes4v) {                                    // octant versions of the node records (ordinary lock-step kernel)
        switch (sel) {
            case 0: k_trace<false, false, false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 1: k_trace<true, false, false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 2: k_trace<false, true, false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 3: k_trace<true, true, false, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 4: k_trace<false, false, true, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 5: k_trace<true, false, true, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            case 6: k_trace<false, true, true, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
            default: k_trace<true, true, true, false, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        }
        return;
    }
    // (round 5, late) the cooperative kernel walks the octant versions too -- product / counting build of the plain chain: its node step loses the sorting network and the
    // low / high selects like the ordinary kernel's did, and all 64 lanes of a unit walk ONE ray, one octant: BASELINE configs[3] lone launch 4.72 -> 4.52 ms, pipelined 5.21 -> 5.10,
    // an eighth of the pulse 1.18 -> 1.14 ms (profiles/r05p_coop_versions.log).  RTS_COOP_VERSIONS=0: the plain records.
    if (COOP && a.nodes4v && a.coop_versions && sel < 2) {
        if (sel == 0) k_trace<false, false, false, true, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a);
        else k_trace<true, false, false, true, false, false, true><<<grid, RTS_BLOCK, 0, st>>>(a);
        return;
    }
    switch (sel) {
        case 0: k_trace<false, false, false, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        case 1: k_trace<true, false, false, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        case 2: k_trace<false, true, false, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        case 3: k_trace<true, true, false, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        case 4: k_trace<false, false, true, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        case 5: k_trace<true, false, true, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        case 6: k_trace<false, true, true, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
        default: k_trace<true, true, true, COOP><<<grid, RTS_BLOCK, 0, st>>>(a); break;
    }
}

// coop_grid > 0: the cooperative kernel is launched FIRST (it holds the most expensive work of the launch), on a stream of its
// own at the trace stream's priority, so that the two kernels' blocks fill the chip side by side from the start (on the
// handle's high-priority stream the cooperative blocks -- three per CU -- crowd the ordinary kernel out until they retire:
// C4 9.4 instead of 8.7 ms).  The stream is created by the first launch that needs it: HIP maps streams onto a handful of
// hardware queues, and one more stream per handle made unrelated handles of a three-pulse pipeline share a queue --
// 0.69 -> 0.93 ms per pulse on C3, where there is no cooperative work at all.
int rts_trace_launch(RtsContext* c, const RtsTraceArgs& a_in, bool count_traversal, unsigned coop_grid)
{
    RtsTraceArgs a = a_in;
    a.done_ctr = c->sum_in_kernel ? a.tile_ctr + RTS_OFF_HEAD + 3 : nullptr;      // (the pad word behind the head words: zeroed with them)
    a.n_blocks_all = a.total_threads / RTS_BLOCK + coop_grid; a.host_cnt = c->pin_dev->cnt;
    if (a.n_rays == 0) {                                            // nothing to trace (an interleaved part without launch indices): the counters still go home, as zeros
        k_sum_counters<<<1, 256, 0, c->tstream_now>>>(a.block_counters, 0u, a.counters, nullptr, c->pin_dev->cnt);
        RTS_HIP(hipGetLastError());
        return RTS_OK;
    }
    const unsigned grid = a.total_threads / RTS_BLOCK;
    hipStream_t st = c->tstream_now;
    if (count_traversal) RTS_HIP(hipMemsetAsync(a.block_counters, 0xff, sizeof(unsigned long long) * 8 * ((size_t)grid + coop_grid), st));      // poison: every block must write its row
    if (coop_grid) {
        if (!c->cstream) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); RTS_HIP(hipStreamCreateWithPriority(&c->cstream, hipStreamNonBlocking, lo)); }
        RTS_HIP(hipEventRecord(c->ev_coop[0], st));
        RTS_HIP(hipStreamWaitEvent(c->cstream, c->ev_coop[0], 0));
        rts_trace_dispatch<true>(a, count_traversal, coop_grid, c->cstream);
        RTS_HIP(hipEventRecord(c->ev_coop[1], c->cstream));
    }
    rts_trace_dispatch<false>(a, count_traversal, grid, st);
    if (coop_grid) RTS_HIP(hipStreamWaitEvent(st, c->ev_coop[1], 0));
    if (!c->sum_in_kernel) k_sum_counters<<<1, 256, 0, st>>>(a.block_counters, grid + coop_grid, a.counters, a.tile_head_all, c->pin_dev->cnt);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// rts_create: have the runtime load the two kernels a first pulse may need, so that the first launch of the cooperative
// kernel -- in the middle of an interval -- does not pay for the upload of its code object
void rts_trace_preload()
{
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_trace<false, false, false, false>));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_trace<false, false, false, true>));
}
