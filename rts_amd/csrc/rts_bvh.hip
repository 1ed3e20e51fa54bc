// rts_bvh.hip -- per-pulse scene placement and LBVH build on the device.
//
// Replaces (does not port) the closed-source OptiX "Bvh" acceleration that the reference
// rebuilds every pulse (ray_tracer.cpp:1126-1130) and the per-primitive `bound` program
// (triangle_mesh.cu:204-233).  Pipeline, all on one stream:
//   place      : world vertices/normals = R * local + position          (ray_tracer.cpp:993-1014)
//   prim_boxes : f64 min/max -> f32 rounded outward (+ conservative pad), scene bounds
//   morton     : 63-bit Morton code of the box centre
//   sort       : radix sort (rocPRIM) of (code, primitive)
//   leaves     : gather the three f64 vertices of each primitive into leaf order
//   hierarchy  : Karras 2012 radix tree over the sorted codes
//   refit      : bottom-up child boxes with one acq_rel counter per node
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "rts_internal.h"

// --------------------------------------------------------------------------- placement
// verts_rot[v][i] = sum_k R[i][k] * vert[v][k], accumulated from 0 in k order
// (matrix_multiply, ray_tracer.cpp:120-137,166), then += position (:1010-1014).
__global__ void k_place(const double* __restrict__ local, double* __restrict__ world, const uint32_t* __restrict__ targ_of,
                        const RtsTargetMotion* __restrict__ motion, uint32_t n, int add_position)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtsTargetMotion m = motion[targ_of[i]];
    double x = local[3*i], y = local[3*i+1], z = local[3*i+2];
    if (m.has_rotation) {
        double rx = 0.0, ry = 0.0, rz = 0.0;
        rx += m.rotation[0] * x; rx += m.rotation[1] * y; rx += m.rotation[2] * z;
        ry += m.rotation[3] * x; ry += m.rotation[4] * y; ry += m.rotation[5] * z;
        rz += m.rotation[6] * x; rz += m.rotation[7] * y; rz += m.rotation[8] * z;
        x = rx; y = ry; z = rz;
    }
    if (add_position) { x += m.position[0]; y += m.position[1]; z += m.position[2]; }
    world[3*i] = x; world[3*i+1] = y; world[3*i+2] = z;
}

// order-preserving float <-> uint map for atomic min/max
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// --------------------------------------------------------------------------- primitive boxes
// Box = [rd(min - pad), ru(max + pad)] in f32.  The reference's `bound` rounds the f64
// extent outward to f32 (triangle_mesh.cu:223-229); the extra pad (2^-22 of the largest
// coordinate magnitude of the box) makes the f64 slab test of the traversal kernel
// conservative with respect to the f64 triangle test even for coordinates that are exactly
// representable (e.g. a plate in the plane z = 0, whose rounded box would have zero thickness).
__global__ void k_prim_boxes(const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, float* __restrict__ prim_box,
                             uint32_t* __restrict__ bounds, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float cx = 0, cy = 0, cz = 0; bool ok = false;
    if (i < n) {
        uint32_t a = tri_vidx[3*i], b = tri_vidx[3*i+1], c = tri_vidx[3*i+2];
        double ax = verts[3*a], ay = verts[3*a+1], az = verts[3*a+2];
        double bx = verts[3*b], by = verts[3*b+1], bz = verts[3*b+2];
        double cx_ = verts[3*c], cy_ = verts[3*c+1], cz_ = verts[3*c+2];
        double lox = fmin(fmin(ax, bx), cx_), loy = fmin(fmin(ay, by), cy_), loz = fmin(fmin(az, bz), cz_);
        double hix = fmax(fmax(ax, bx), cx_), hiy = fmax(fmax(ay, by), cy_), hiz = fmax(fmax(az, bz), cz_);
        bool finite = isfinite(ax) && isfinite(ay) && isfinite(az) && isfinite(bx) && isfinite(by) && isfinite(bz) &&
                      isfinite(cx_) && isfinite(cy_) && isfinite(cz_);
        float* o = prim_box + 6*(size_t)i;
        if (finite) {
            double s = fmax(fmax(fmax(fabs(lox), fabs(hix)), fmax(fabs(loy), fabs(hiy))), fmax(fabs(loz), fabs(hiz)));
            double pad = s * 2.384185791015625e-07 + 1e-30;
            o[0] = f32_down(lox - pad); o[1] = f32_down(loy - pad); o[2] = f32_down(loz - pad);
            o[3] = f32_up(hix + pad); o[4] = f32_up(hiy + pad); o[5] = f32_up(hiz + pad);
            cx = (float)((lox + hix) * 0.5); cy = (float)((loy + hiy) * 0.5); cz = (float)((loz + hiz) * 0.5);
            ok = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(o[3]) && isfinite(o[4]) && isfinite(o[5]);
        }
        if (!ok) { o[0] = o[1] = o[2] = 3.0e38f; o[3] = o[4] = o[5] = -3.0e38f; }   // empty box: never hit
    }
    // block reduction of centre bounds (wave shuffles, then LDS), one atomic set per block
    float mnx = ok ? cx : 3.0e38f, mny = ok ? cy : 3.0e38f, mnz = ok ? cz : 3.0e38f;
    float mxx = ok ? cx : -3.0e38f, mxy = ok ? cy : -3.0e38f, mxz = ok ? cz : -3.0e38f;
    for (int off = 32; off > 0; off >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, off)); mny = fminf(mny, __shfl_down(mny, off)); mnz = fminf(mnz, __shfl_down(mnz, off));
        mxx = fmaxf(mxx, __shfl_down(mxx, off)); mxy = fmaxf(mxy, __shfl_down(mxy, off)); mxz = fmaxf(mxz, __shfl_down(mxz, off));
    }
    __shared__ float s_red[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[wave][0] = mnx; s_red[wave][1] = mny; s_red[wave][2] = mnz; s_red[wave][3] = mxx; s_red[wave][4] = mxy; s_red[wave][5] = mxz; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = s_red[0][k];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) v = (k < 3) ? fminf(v, s_red[w][k]) : fmaxf(v, s_red[w][k]);
        if (k < 3) atomicMin(&bounds[k], f2ord(v)); else atomicMax(&bounds[k], f2ord(v));
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

__global__ void k_morton(const float* __restrict__ prim_box, const uint32_t* __restrict__ bounds, uint64_t* __restrict__ keys,
                         uint32_t* __restrict__ vals, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* b = prim_box + 6*(size_t)i;
    vals[i] = i;
    if (b[0] > b[3]) { keys[i] = 0x7fffffffffffffffULL; return; }           // empty box sorts last
    float lx = ord2f(bounds[0]), ly = ord2f(bounds[1]), lz = ord2f(bounds[2]);
    float hx = ord2f(bounds[3]), hy = ord2f(bounds[4]), hz = ord2f(bounds[5]);
    float ex = fmaxf(hx - lx, 1e-30f), ey = fmaxf(hy - ly, 1e-30f), ez = fmaxf(hz - lz, 1e-30f);
    float e = fmaxf(ex, fmaxf(ey, ez));                                     // cubic grid keeps cells isotropic
    float cx = (b[0] + b[3]) * 0.5f, cy = (b[1] + b[4]) * 0.5f, cz = (b[2] + b[5]) * 0.5f;
    double sx = fmin(fmax((double)(cx - lx) / e, 0.0), 1.0), sy = fmin(fmax((double)(cy - ly) / e, 0.0), 1.0), sz = fmin(fmax((double)(cz - lz) / e, 0.0), 1.0);
    uint64_t qx = (uint64_t)(sx * 2097151.0), qy = (uint64_t)(sy * 2097151.0), qz = (uint64_t)(sz * 2097151.0);
    keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
}

__global__ void k_leaves(const uint32_t* __restrict__ sorted_prim, const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts,
                         const uint32_t* __restrict__ prim_targ, RtsLeafTri* __restrict__ leaves, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t g = sorted_prim[i];
    uint32_t a = tri_vidx[3*g], b = tri_vidx[3*g+1], c = tri_vidx[3*g+2];
    RtsLeafTri L;
    L.p0x = verts[3*a]; L.p0y = verts[3*a+1]; L.p0z = verts[3*a+2];
    L.p1x = verts[3*b]; L.p1y = verts[3*b+1]; L.p1z = verts[3*b+2];
    L.p2x = verts[3*c]; L.p2y = verts[3*c+1]; L.p2z = verts[3*c+2];
    L.prim = g; L.targ = prim_targ[g];
    leaves[i] = L;
}

// --------------------------------------------------------------------------- Karras radix tree
__device__ __forceinline__ int lcp(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ void k_hierarchy(const uint64_t* __restrict__ keys, RtsNode* __restrict__ nodes, int32_t* __restrict__ parent,
                            int32_t* __restrict__ leaf_parent, int2* __restrict__ range, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lcp(keys, n, i, i - d);
    int lmax = 2;
    while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lcp(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
        if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    int gamma = i + s * d + (d < 0 ? d : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int left, right;
    if (lo == gamma) { left = ~gamma; leaf_parent[gamma] = i; } else { left = gamma; parent[gamma] = i; }
    if (hi == gamma + 1) { right = ~(gamma + 1); leaf_parent[gamma + 1] = i; } else { right = gamma + 1; parent[gamma + 1] = i; }
    nodes[i].c0 = left; nodes[i].c1 = right; nodes[i].pad0 = 0; nodes[i].pad1 = 0;
    range[i] = make_int2(lo, hi);
    if (i == 0) parent[0] = -1;
}

// Bottom-up refit.  Each node record stores the boxes of its two children; a walker that
// carries the finished box of a subtree writes it into its parent's child slot and bumps the
// parent's arrival counter: the first arriver stops, the second reads the sibling box, forms the
// union and carries on upward.
//   * local phase: one workgroup owns RF_CHUNK consecutive leaves; every node whose leaf range
//     lies inside the chunk (it then has index in [c0, c0+RF_CHUNK)) is resolved through LDS
//     counters and LDS copies of the child boxes -- no inter-workgroup traffic at all;
//   * global phase: the O(log n) subtree roots per chunk whose parents span chunks go through
//     global memory: sc1 stores of the box, ONE acq_rel agent-scope counter bump per level (per-XCD
//     L2s and per-CU L1s are not coherent), sc1 loads of the sibling box.
#define RF_CHUNK 1024
#define RF_THREADS 256
#define RF_PEND 256

__device__ __forceinline__ void box_union(float a[6], const float b[6]) {
    a[0] = fminf(a[0], b[0]); a[1] = fminf(a[1], b[1]); a[2] = fminf(a[2], b[2]);
    a[3] = fmaxf(a[3], b[3]); a[4] = fmaxf(a[4], b[4]); a[5] = fmaxf(a[5], b[5]);
}

__device__ __forceinline__ void refit_global_walk(RtsNode* nodes, const int32_t* __restrict__ parent, uint32_t* flags, int p, int child, float box[6])
{
    while (p >= 0) {
        float* nd = reinterpret_cast<float*>(nodes + p);
        const int slot = (reinterpret_cast<const int32_t*>(nd)[12] == child) ? 0 : 1;
        for (int k = 0; k < 6; k++) __hip_atomic_store(nd + 6*slot + k, box[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned old = __hip_atomic_fetch_add(&flags[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0) return;
        float sib[6];
        for (int k = 0; k < 6; k++) sib[k] = __hip_atomic_load(nd + 6*(slot ^ 1) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        box_union(box, sib);
        child = p; p = parent[p];
    }
}

__global__ void __launch_bounds__(RF_THREADS) k_refit(const float* __restrict__ prim_box, const uint32_t* __restrict__ sorted_prim, RtsNode* nodes,
        const int32_t* __restrict__ parent, const int32_t* __restrict__ leaf_parent, const int2* __restrict__ range, uint32_t* flags, int n)
{
    __shared__ float s_cbox[RF_CHUNK][2][6];
    __shared__ int s_flag[RF_CHUNK];
    __shared__ float s_pbox[RF_PEND][6];
    __shared__ int s_pnode[RF_PEND], s_pchild[RF_PEND];
    __shared__ int s_npend;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * RF_CHUNK;
    const int c1 = min(c0 + RF_CHUNK, n) - 1;
    for (int k = tid; k < RF_CHUNK; k += RF_THREADS) s_flag[k] = 0;
    if (tid == 0) s_npend = 0;
    __syncthreads();
    for (int k = 0; k < RF_CHUNK / RF_THREADS; k++) {
        const int leaf = c0 + k * RF_THREADS + tid;
        if (leaf > c1) continue;
        float box[6];
        { const float* pb = prim_box + 6*(size_t)sorted_prim[leaf]; for (int q = 0; q < 6; q++) box[q] = pb[q]; }
        int child = ~leaf, p = leaf_parent[leaf];
        while (p >= 0) {
            const int2 r = range[p];
            if (r.x < c0 || r.y > c1) {                       // parent spans chunks: hand over to the global phase
                const int idx = atomicAdd(&s_npend, 1);
                if (idx < RF_PEND) { s_pnode[idx] = p; s_pchild[idx] = child; for (int q = 0; q < 6; q++) s_pbox[idx][q] = box[q]; }
                else refit_global_walk(nodes, parent, flags, p, child, box);      // list full (cannot happen for depth < 128)
                break;
            }
            float* nd = reinterpret_cast<float*>(nodes + p);
            const int slot = (reinterpret_cast<const int32_t*>(nd)[12] == child) ? 0 : 1;
            const int li = p - c0;
            for (int q = 0; q < 6; q++) { nd[6*slot + q] = box[q]; s_cbox[li][slot][q] = box[q]; }
            __threadfence_block();
            const int old = atomicAdd(&s_flag[li], 1);
            if (old == 0) break;
            __threadfence_block();
            float sib[6];
            for (int q = 0; q < 6; q++) sib[q] = s_cbox[li][slot ^ 1][q];
            box_union(box, sib);
            child = p; p = parent[p];
        }
    }
    __syncthreads();
    const int np = min(s_npend, RF_PEND);
    for (int i = tid; i < np; i += RF_THREADS) {
        float box[6];
        for (int q = 0; q < 6; q++) box[q] = s_pbox[i][q];
        refit_global_walk(nodes, parent, flags, s_pnode[i], s_pchild[i], box);
    }
}

// BVH2 -> BVH4 collapse: record i holds, for BVH2 node i, the children of its internal children (boxes taken from
// the children's own records) and its leaf children as they are.  Only the records reachable from the root through
// these links are ever visited; the others cost nothing but their 128 bytes.
__global__ void k_collapse4(const RtsNode* __restrict__ nodes, RtsNode4* __restrict__ nodes4, int n_nodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const RtsNode nd = nodes[i];
    RtsNode4 o;
    for (int k = 0; k < 4; k++) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = -3.0e38f; o.child[k] = 0x7fffffff; o.pad[k] = 0; }
    int m = 0;
    auto put = [&](int c, float lx, float ly, float lz, float hx, float hy, float hz) {
        o.lox[m] = lx; o.loy[m] = ly; o.loz[m] = lz; o.hix[m] = hx; o.hiy[m] = hy; o.hiz[m] = hz; o.child[m] = c; m++;
    };
    auto expand = [&](int c, float lx, float ly, float lz, float hx, float hy, float hz) {
        if (c < 0) { put(c, lx, ly, lz, hx, hy, hz); return; }
        const RtsNode ch = nodes[c];
        put(ch.c0, ch.lo0x, ch.lo0y, ch.lo0z, ch.hi0x, ch.hi0y, ch.hi0z);
        put(ch.c1, ch.lo1x, ch.lo1y, ch.lo1z, ch.hi1x, ch.hi1y, ch.hi1z);
    };
    expand(nd.c0, nd.lo0x, nd.lo0y, nd.lo0z, nd.hi0x, nd.hi0y, nd.hi0z);
    if (!(nd.lo1x > nd.hi1x && nd.c1 == nd.c0)) expand(nd.c1, nd.lo1x, nd.lo1y, nd.lo1z, nd.hi1x, nd.hi1y, nd.hi1z);   // (single-leaf scene: second slot is a dummy)
    nodes4[i] = o;
}

// a scene with a single primitive: one node whose second child is an empty box
__global__ void k_single_leaf(const float* __restrict__ prim_box, RtsNode* nodes)
{
    RtsNode nd;
    nd.lo0x = prim_box[0]; nd.lo0y = prim_box[1]; nd.lo0z = prim_box[2]; nd.hi0x = prim_box[3]; nd.hi0y = prim_box[4]; nd.hi0z = prim_box[5];
    nd.lo1x = nd.lo1y = nd.lo1z = 3.0e38f; nd.hi1x = nd.hi1y = nd.hi1z = -3.0e38f;
    nd.c0 = ~0; nd.c1 = ~0; nd.pad0 = nd.pad1 = 0;
    nodes[0] = nd;
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

int rts_bvh_build(RtsContext* c)
{
    const uint32_t n = c->n_prims;
    hipStream_t st = c->stream;
    // placement
    if (c->n_verts) k_place<<<blocks_for(c->n_verts, 256), 256, 0, st>>>(c->d_verts_local.p, c->d_verts_world.p, c->d_vert_targ.p, c->d_motion.p, c->n_verts, 1);
    if (c->n_normals) k_place<<<blocks_for(c->n_normals, 256), 256, 0, st>>>(c->d_normals_local.p, c->d_normals_world.p, c->d_norm_targ.p, c->d_motion.p, c->n_normals, 0);
    c->n_nodes = 0;
    if (n == 0) { RTS_HIP(hipGetLastError()); return RTS_OK; }
    RTS_HIP(c->d_prim_box.reserve(6*(size_t)n)); RTS_HIP(c->d_node_box.reserve(6*(size_t)n));
    RTS_HIP(c->d_keys.reserve(n)); RTS_HIP(c->d_keys_sorted.reserve(n)); RTS_HIP(c->d_vals.reserve(n)); RTS_HIP(c->d_vals_sorted.reserve(n));
    RTS_HIP(c->d_bounds.reserve(8)); RTS_HIP(c->d_parent.reserve(n)); RTS_HIP(c->d_leaf_parent.reserve(n)); RTS_HIP(c->d_flags.reserve(n));
    RTS_HIP(c->d_nodes.reserve(n)); RTS_HIP(c->d_nodes4.reserve(n)); RTS_HIP(c->d_leaves.reserve(n));
    static const uint32_t init_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    RTS_HIP(hipMemcpyAsync(c->d_bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, st));
    k_prim_boxes<<<blocks_for(n, 256), 256, 0, st>>>(c->d_tri_vidx.p, c->d_verts_world.p, c->d_prim_box.p, c->d_bounds.p, n);
    k_morton<<<blocks_for(n, 256), 256, 0, st>>>(c->d_prim_box.p, c->d_bounds.p, c->d_keys.p, c->d_vals.p, n);
    size_t tmp = 0;
    RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d_keys.p, c->d_keys_sorted.p, c->d_vals.p, c->d_vals_sorted.p, n, 0, 64, st));
    RTS_HIP(c->d_sort_tmp.reserve(tmp));
    RTS_HIP(rocprim::radix_sort_pairs(c->d_sort_tmp.p, tmp, c->d_keys.p, c->d_keys_sorted.p, c->d_vals.p, c->d_vals_sorted.p, n, 0, 64, st));
    k_leaves<<<blocks_for(n, 256), 256, 0, st>>>(c->d_vals_sorted.p, c->d_tri_vidx.p, c->d_verts_world.p, c->d_prim_targ.p, c->d_leaves.p, n);
    if (n == 1) {
        k_single_leaf<<<1, 1, 0, st>>>(c->d_prim_box.p, c->d_nodes.p);
        c->n_nodes = 1;
    } else {
        RTS_HIP(hipMemsetAsync(c->d_flags.p, 0, sizeof(uint32_t) * n, st));
        k_hierarchy<<<blocks_for(n - 1, 256), 256, 0, st>>>(c->d_keys_sorted.p, c->d_nodes.p, c->d_parent.p, c->d_leaf_parent.p, (int2*)c->d_node_box.p, (int)n);
        k_refit<<<blocks_for(n, RF_CHUNK), RF_THREADS, 0, st>>>(c->d_prim_box.p, c->d_vals_sorted.p, c->d_nodes.p, c->d_parent.p, c->d_leaf_parent.p, (const int2*)c->d_node_box.p, c->d_flags.p, (int)n);
        c->n_nodes = n - 1;
    }
    k_collapse4<<<blocks_for(c->n_nodes, 256), 256, 0, st>>>(c->d_nodes.p, c->d_nodes4.p, (int)c->n_nodes);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}
