// rts_lbvh.hip -- the static target-space hierarchy built ON THE DEVICE (RTS_FLAG_DEVICE_BUILD / RTS_BUILDER=device).
//
// The reference has OptiX rebuild its closed-source "Bvh" acceleration on the device every pulse (ray_tracer.cpp:1126-1130);
// here the hierarchy of a rigid target never changes (rts_sah.cpp), so it is built once per rts_set_scene -- by the host SAH
// builder (default: best traversal cost, seconds for a million triangles on one core per mesh) or by this file (milliseconds):
//   prim_boxes : f64 extent of each triangle in TARGET space -> f32 rounded outward + conservative pad (put_box of
//                rts_sah.cpp; the reference's `bound` program, triangle_mesh.cu:204-233, works on world-space boxes)
//   morton     : 63-bit Morton code of the box centre in the mesh's (cubic) bounds
//   sort       : rocPRIM radix sort of (code, triangle); triangles with a non-finite vertex sort last and get no leaf
//   hierarchy  : Karras 2012 radix tree over the sorted codes, one thread per internal node
//   refit      : bottom-up child boxes; a 1024-leaf chunk is resolved through LDS counters, the few subtree roots whose
//                parents span chunks through agent-scope atomics (per-XCD L2s are not coherent)
//   collapse   : record i = BVH2 node i with its internal children opened -> the 4-wide, 128-byte node format the trace
//                kernel walks (RtsNode4); records not reachable from the root are never visited
// Same node format and leaf-order convention as the host builder, no split references; results cannot differ (the f64
// triangle test alone decides hits), only the number of nodes and triangles a ray visits.
#include <cmath>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "rts_internal.h"

namespace {

struct __attribute__((aligned(64))) Node2 {          // binary node: the boxes of its two children
    float lo0x, lo0y, lo0z, hi0x, hi0y, hi0z, lo1x, lo1y, lo1z, hi1x, hi1y, hi1z;
    int32_t c0, c1, pad0, pad1;                      // >= 0 node, < 0 ~leaf (sorted position)
};
static_assert(sizeof(Node2) == 64, "node2 size");

// order-preserving float <-> uint map for atomic min/max
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// Box = [rd(min - pad), ru(max + pad)] in f32, pad = 2^-22 of the largest coordinate magnitude (rts_sah.cpp: put_box).
// Invalid (non-finite) triangles get the INVERTED box, neutral under union; they sort last and stay outside the tree.
__global__ void k_prim_boxes(const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, float* __restrict__ prim_box,
                             uint32_t* __restrict__ bounds, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float cx = 0, cy = 0, cz = 0; bool ok = false;
    if (i < n) {
        uint32_t a = tri_vidx[3*i], b = tri_vidx[3*i+1], c = tri_vidx[3*i+2];
        double ax = verts[3*(size_t)a], ay = verts[3*(size_t)a+1], az = verts[3*(size_t)a+2];
        double bx = verts[3*(size_t)b], by = verts[3*(size_t)b+1], bz = verts[3*(size_t)b+2];
        double cx_ = verts[3*(size_t)c], cy_ = verts[3*(size_t)c+1], cz_ = verts[3*(size_t)c+2];
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
        if (!ok) { o[0] = o[1] = o[2] = 3.0e38f; o[3] = o[4] = o[5] = -3.0e38f; }
    }
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
    if (b[0] > b[3]) { keys[i] = 0x7fffffffffffffffULL; return; }           // invalid triangle: sorts last
    float lx = ord2f(bounds[0]), ly = ord2f(bounds[1]), lz = ord2f(bounds[2]);
    float hx = ord2f(bounds[3]), hy = ord2f(bounds[4]), hz = ord2f(bounds[5]);
    float ex = fmaxf(hx - lx, 1e-30f), ey = fmaxf(hy - ly, 1e-30f), ez = fmaxf(hz - lz, 1e-30f);
    float e = fmaxf(ex, fmaxf(ey, ez));                                     // cubic grid keeps cells isotropic
    float cx = (b[0] + b[3]) * 0.5f, cy = (b[1] + b[4]) * 0.5f, cz = (b[2] + b[5]) * 0.5f;
    double sx = fmin(fmax((double)(cx - lx) / e, 0.0), 1.0), sy = fmin(fmax((double)(cy - ly) / e, 0.0), 1.0), sz = fmin(fmax((double)(cz - lz) / e, 0.0), 1.0);
    uint64_t qx = (uint64_t)(sx * 2097151.0), qy = (uint64_t)(sy * 2097151.0), qz = (uint64_t)(sz * 2097151.0);
    keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
}

// leaf order of the mesh: leaf slot (leaf_base + sorted position) -> GLOBAL primitive id
__global__ void k_leaf_order(const uint32_t* __restrict__ sorted_prim, uint32_t tri_base, uint32_t* __restrict__ leaf_prim, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaf_prim[i] = tri_base + sorted_prim[i];
}

// --------------------------------------------------------------------------- Karras radix tree
__device__ __forceinline__ int lcp(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

__global__ void k_hierarchy(const uint64_t* __restrict__ keys, Node2* __restrict__ nodes, int32_t* __restrict__ parent,
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

#define RF_CHUNK 1024
#define RF_THREADS 256
#define RF_PEND 256

__device__ __forceinline__ void box_union(float a[6], const float b[6]) {
    a[0] = fminf(a[0], b[0]); a[1] = fminf(a[1], b[1]); a[2] = fminf(a[2], b[2]);
    a[3] = fmaxf(a[3], b[3]); a[4] = fmaxf(a[4], b[4]); a[5] = fmaxf(a[5], b[5]);
}

__device__ __forceinline__ void refit_global_walk(Node2* nodes, const int32_t* __restrict__ parent, uint32_t* flags, int p, int child, float box[6])
{
    int guard = 0;
    while (p >= 0 && guard++ < 4096) {                 // (a radix tree over n leaves is at most n deep; every walker terminates)
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

__global__ void __launch_bounds__(RF_THREADS) k_refit(const float* __restrict__ prim_box, const uint32_t* __restrict__ sorted_prim, Node2* nodes,
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

// BVH2 -> BVH4: record i holds, for BVH2 node i, the children of its internal children (boxes taken from the children's own
// records) and its leaf children as they are.  Child links become global: node_base + BVH2 index, ~(leaf_base + position).
// Unused slots: the degenerate box lo = hi = 3e38 (rts_sah.cpp: new_node).
__global__ void k_collapse4(const Node2* __restrict__ nodes, RtsNode4* __restrict__ nodes4, int n_nodes, int32_t node_base, int32_t leaf_base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const Node2 nd = nodes[i];
    RtsNode4 o;
    for (int k = 0; k < 4; k++) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = 3.0e38f; o.child[k] = 0x7fffffff; o.pad[k] = 0; }
    int m = 0;
    auto put = [&](int c, float lx, float ly, float lz, float hx, float hy, float hz) {
        o.lox[m] = lx; o.loy[m] = ly; o.loz[m] = lz; o.hix[m] = hx; o.hiy[m] = hy; o.hiz[m] = hz;
        o.child[m] = c >= 0 ? c + node_base : ~(~c + leaf_base); m++;
    };
    auto expand = [&](int c, float lx, float ly, float lz, float hx, float hy, float hz) {
        if (c < 0) { put(c, lx, ly, lz, hx, hy, hz); return; }
        const Node2 ch = nodes[c];
        put(ch.c0, ch.lo0x, ch.lo0y, ch.lo0z, ch.hi0x, ch.hi0y, ch.hi0z);
        put(ch.c1, ch.lo1x, ch.lo1y, ch.lo1z, ch.hi1x, ch.hi1y, ch.hi1z);
    };
    expand(nd.c0, nd.lo0x, nd.lo0y, nd.lo0z, nd.hi0x, nd.hi0y, nd.hi0z);
    expand(nd.c1, nd.lo1x, nd.lo1y, nd.lo1z, nd.hi1x, nd.hi1y, nd.hi1z);
    nodes4[i] = o;
}

// a mesh with a single valid triangle: one node with one leaf child
__global__ void k_single_leaf4(const float* __restrict__ prim_box, const uint32_t* __restrict__ sorted_prim, RtsNode4* nodes4, int32_t leaf_base)
{
    RtsNode4 o;
    for (int k = 0; k < 4; k++) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = 3.0e38f; o.child[k] = 0x7fffffff; o.pad[k] = 0; }
    const float* b = prim_box + 6*(size_t)sorted_prim[0];
    o.lox[0] = b[0]; o.loy[0] = b[1]; o.loz[0] = b[2]; o.hix[0] = b[3]; o.hiy[0] = b[4]; o.hiz[0] = b[5]; o.child[0] = ~leaf_base;
    nodes4[0] = o;
}

inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

// vidx: [n_prims][3] GLOBAL vertex indices (host copy of ns->d_tri_vidx); mh: per-mesh slices.  Fills ns->d_nodes4,
// ns->d_leaf_prim, ns->blas, ns->n_nodes, ns->n_leaves.  Uses the handle's stream; temporaries are freed before returning.
int rts_lbvh_build_device(RtsContext* c, RtsScene* ns, const std::vector<uint32_t>& vidx, const std::vector<RtsMeshHost>& mh)
{
    hipStream_t st = c->stream;
    const uint32_t n_targets = (uint32_t)mh.size();
    ns->blas.assign(n_targets, RtsBlasInfo{});
    // host pass: which triangles are finite, the f64 bounds of every mesh (for the per-pulse placement constants)
    std::vector<double> hv(3 * (size_t)ns->n_verts);
    if (ns->n_verts) RTS_HIP(hipMemcpy(hv.data(), ns->d_verts_local.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost));
    std::vector<uint32_t> n_valid(n_targets, 0);
    uint64_t node_total = 0, leaf_total = 0; uint32_t n_max = 0;
    for (uint32_t t = 0; t < n_targets; t++) {
        RtsBlasInfo& b = ns->blas[t]; b.root = -1; b.depth = 64;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < mh[t].n_tris; i++) {
            bool finite = true;
            for (int k = 0; k < 3; k++) { const double* p = &hv[3 * (size_t)vidx[3 * ((size_t)mh[t].tri_base + i) + k]]; finite = finite && std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }
            if (!finite) continue;
            n_valid[t]++;
            for (int k = 0; k < 3; k++) { const double* p = &hv[3 * (size_t)vidx[3 * ((size_t)mh[t].tri_base + i) + k]]; for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
        }
        b.n_leaves = n_valid[t]; b.n_nodes = n_valid[t] == 0 ? 0 : std::max<uint32_t>(n_valid[t] - 1, 1);
        for (int a = 0; a < 3; a++) { b.lo[a] = n_valid[t] ? lo[a] : 0; b.hi[a] = n_valid[t] ? hi[a] : 0; b.max_abs = std::max(b.max_abs, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]))); }
        node_total += b.n_nodes; leaf_total += b.n_leaves; n_max = std::max(n_max, mh[t].n_tris);
    }
    RTS_HIP(ns->d_nodes4.reserve((size_t)node_total + 1)); RTS_HIP(ns->d_leaf_prim.reserve((size_t)leaf_total + 1));
    ns->n_nodes = (uint32_t)node_total; ns->n_leaves = (uint32_t)leaf_total;
    if (n_max == 0) return RTS_OK;
    DevBuf<float> d_prim_box; DevBuf<uint64_t> d_keys, d_keys_sorted; DevBuf<uint32_t> d_vals, d_vals_sorted, d_bounds, d_flags; DevBuf<int32_t> d_parent, d_leaf_parent;
    DevBuf<Node2> d_nodes2; DevBuf<int2> d_range; DevBuf<char> d_tmp;
    struct Free { DevBuf<float>& a; DevBuf<uint64_t>& b; DevBuf<uint64_t>& b2; DevBuf<uint32_t>& c1; DevBuf<uint32_t>& c2; DevBuf<uint32_t>& c3; DevBuf<uint32_t>& c4; DevBuf<int32_t>& d1; DevBuf<int32_t>& d2; DevBuf<Node2>& e; DevBuf<int2>& f; DevBuf<char>& g;
                  ~Free() { a.release(); b.release(); b2.release(); c1.release(); c2.release(); c3.release(); c4.release(); d1.release(); d2.release(); e.release(); f.release(); g.release(); } }
        free_all{d_prim_box, d_keys, d_keys_sorted, d_vals, d_vals_sorted, d_bounds, d_flags, d_parent, d_leaf_parent, d_nodes2, d_range, d_tmp};
    RTS_HIP(d_prim_box.reserve(6 * (size_t)n_max)); RTS_HIP(d_keys.reserve(n_max)); RTS_HIP(d_keys_sorted.reserve(n_max)); RTS_HIP(d_vals.reserve(n_max)); RTS_HIP(d_vals_sorted.reserve(n_max));
    RTS_HIP(d_bounds.reserve(8)); RTS_HIP(d_flags.reserve(n_max)); RTS_HIP(d_parent.reserve(n_max)); RTS_HIP(d_leaf_parent.reserve(n_max)); RTS_HIP(d_nodes2.reserve(n_max)); RTS_HIP(d_range.reserve(n_max));
    size_t tmp = 0;
    RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, d_keys.p, d_keys_sorted.p, d_vals.p, d_vals_sorted.p, n_max, 0, 64, st));
    RTS_HIP(d_tmp.reserve(tmp));
    int32_t node_base = 0, leaf_base = 0;
    for (uint32_t t = 0; t < n_targets; t++) {
        const uint32_t n = mh[t].n_tris, nv = n_valid[t];
        RtsBlasInfo& b = ns->blas[t];
        if (nv == 0) continue;
        static const uint32_t init_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
        RTS_HIP(hipMemcpyAsync(d_bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, st));
        const uint32_t* tv = ns->d_tri_vidx.p + 3 * (size_t)mh[t].tri_base;
        k_prim_boxes<<<blocks_for(n, 256), 256, 0, st>>>(tv, ns->d_verts_local.p, d_prim_box.p, d_bounds.p, n);
        k_morton<<<blocks_for(n, 256), 256, 0, st>>>(d_prim_box.p, d_bounds.p, d_keys.p, d_vals.p, n);
        size_t tmp_n = tmp;
        RTS_HIP(rocprim::radix_sort_pairs(d_tmp.p, tmp_n, d_keys.p, d_keys_sorted.p, d_vals.p, d_vals_sorted.p, n, 0, 64, st));
        k_leaf_order<<<blocks_for(nv, 256), 256, 0, st>>>(d_vals_sorted.p, mh[t].tri_base, ns->d_leaf_prim.p + leaf_base, nv);
        if (nv == 1) {
            k_single_leaf4<<<1, 1, 0, st>>>(d_prim_box.p, d_vals_sorted.p, ns->d_nodes4.p + node_base, leaf_base);
        } else {
            RTS_HIP(hipMemsetAsync(d_flags.p, 0, sizeof(uint32_t) * nv, st));
            k_hierarchy<<<blocks_for(nv - 1, 256), 256, 0, st>>>(d_keys_sorted.p, d_nodes2.p, d_parent.p, d_leaf_parent.p, d_range.p, (int)nv);
            k_refit<<<blocks_for(nv, RF_CHUNK), RF_THREADS, 0, st>>>(d_prim_box.p, d_vals_sorted.p, d_nodes2.p, d_parent.p, d_leaf_parent.p, d_range.p, d_flags.p, (int)nv);
            k_collapse4<<<blocks_for(nv - 1, 256), 256, 0, st>>>(d_nodes2.p, ns->d_nodes4.p + node_base, (int)(nv - 1), node_base, leaf_base);
        }
        RTS_HIP(hipGetLastError());
        b.root = node_base;
        node_base += (int32_t)b.n_nodes; leaf_base += (int32_t)b.n_leaves;
        RTS_HIP(hipStreamSynchronize(st));                       // the temporaries are reused by the next mesh
    }
    return RTS_OK;
}
