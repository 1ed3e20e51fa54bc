// rts_lbvh.hip -- the static target-space hierarchy built ON THE DEVICE (RTS_FLAG_DEVICE_BUILD / RTS_BUILDER=device).
//
// The reference has OptiX rebuild its closed-source "Bvh" acceleration on the device every pulse (ray_tracer.cpp:1126-1130);
// here the hierarchy of a rigid target never changes (rts_sah.cpp), so it is built once per rts_set_scene -- by the host SAH
// builder (default: best traversal cost, seconds for a million triangles on one core per mesh) or by this file (milliseconds):
//   references : a triangle whose box is large against the mesh's mean is cut into up to 8 REFERENCES -- slabs across the
//                longest axis of its box, each with the box of the part of the triangle inside the slab (early split
//                clipping, the device counterpart of the split references of rts_sah.cpp): k = ceil(sqrt(area / threshold)),
//                counted, scanned and emitted on the device.  Every reference of a triangle leads to the same exact test.
//   ref_boxes  : f64 extent of each reference in TARGET space -> f32 rounded outward + conservative pad (put_box of
//                rts_sah.cpp; the reference's `bound` program, triangle_mesh.cu:204-233, works on world-space boxes)
//   morton     : 63-bit Morton code of the box centre in the mesh's (cubic) bounds
//   sort       : rocPRIM radix sort of (code, reference); triangles with a non-finite vertex sort last and get no leaf
//   hierarchy  : Karras 2012 radix tree over the sorted codes, one thread per internal node
//   refit      : bottom-up child boxes; a 1024-leaf chunk is resolved through LDS counters, the few subtree roots whose
//                parents span chunks through agent-scope atomics (per-XCD L2s are not coherent)
//   collapse   : record i = BVH2 node i with its internal children opened -> the 4-wide, 128-byte node format the trace
//                kernel walks (RtsNode4); records not reachable from the root are never visited
// Same node format and leaf-order convention as the host builder; results cannot differ (the f64
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

struct TriV { double v[3][3]; bool finite; };
__device__ __forceinline__ TriV load_tri(const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, uint32_t i)
{
    TriV t; t.finite = true;
    for (int k = 0; k < 3; k++) {
        const uint32_t a = tri_vidx[3*(size_t)i + k];
        // (finite AND far enough inside the f32 range for the padded, outward-rounded box of every reference to be finite: a
        // triangle counted as valid by k_ref_count must never come out of k_ref_boxes with the sentinel key)
        for (int c = 0; c < 3; c++) { t.v[k][c] = verts[3*(size_t)a + c]; t.finite = t.finite && fabs(t.v[k][c]) < 3.0e38; }
    }
    return t;
}
__device__ __forceinline__ void tri_extent(const TriV& t, double lo[3], double hi[3])
{
    for (int c = 0; c < 3; c++) { lo[c] = fmin(fmin(t.v[0][c], t.v[1][c]), t.v[2][c]); hi[c] = fmax(fmax(t.v[0][c], t.v[1][c]), t.v[2][c]); }
}
// box of the part of triangle t between the planes x[ax] = s0 and x[ax] = s1 (its vertices inside the slab and the points
// where its edges cross the two planes; f64, the pad of the reference boxes dwarfs the rounding), never outside [tlo, thi]
__device__ __forceinline__ void tri_slab_box(const TriV& t, const double tlo[3], const double thi[3], int ax, double s0, double s1, double lo[3], double hi[3])
{
    for (int c = 0; c < 3; c++) { lo[c] = tlo[c]; hi[c] = thi[c]; }
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int a = 0; a < 3; a++) {
        const double* p = t.v[a]; const double* q = t.v[(a + 1) % 3];
        if (p[ax] >= s0 && p[ax] <= s1) for (int c = 0; c < 3; c++) { clo[c] = fmin(clo[c], p[c]); chi[c] = fmax(chi[c], p[c]); }
        for (int w = 0; w < 2; w++) {
            const double sp = w ? s1 : s0;
            if ((p[ax] < sp && q[ax] > sp) || (p[ax] > sp && q[ax] < sp)) {
                const double u = (sp - p[ax]) / (q[ax] - p[ax]);
                for (int c = 0; c < 3; c++) { const double x = c == ax ? sp : p[c] + u * (q[c] - p[c]); clo[c] = fmin(clo[c], x); chi[c] = fmax(chi[c], x); }
            }
        }
    }
    if (clo[0] <= chi[0] && clo[1] <= chi[1] && clo[2] <= chi[2]) {
        for (int c = 0; c < 3; c++) { lo[c] = fmax(clo[c], tlo[c]); hi[c] = fmin(chi[c], thi[c]); }
    }
    lo[ax] = fmax(lo[ax], s0); hi[ax] = fmin(hi[ax], s1);          // (an empty clip -- it cannot happen inside the extent -- keeps the slab of the whole box)
    if (!(lo[ax] <= hi[ax])) { lo[ax] = s0; hi[ax] = s1; }
}
__device__ __forceinline__ double box_area3(const double lo[3], const double hi[3]) { const double ex = hi[0]-lo[0], ey = hi[1]-lo[1], ez = hi[2]-lo[2]; return 2.0 * (ex*ey + ey*ez + ex*ez); }

// number of references of a triangle: slabs across the longest axis of its box, more for boxes that are large against `a_thr`
__device__ __forceinline__ uint32_t ref_count(const double lo[3], const double hi[3], double a_thr)
{
    const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    const double area = 2.0 * (ex*ey + ey*ez + ex*ez);
    if (!(a_thr > 0.0) || !(area > a_thr) || !(fmax(fmax(ex, ey), ez) > 0.0)) return 1u;
    const double k = ceil(sqrt(area / a_thr));
    return (uint32_t)fmin(fmax(k, 1.0), 8.0);
}

// pass 0: sum of the box areas of the valid triangles (for the split threshold) and their number
__global__ void k_tri_area(const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, uint32_t n, double* __restrict__ acc)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0, c = 0.0;
    if (i < n) {
        const TriV t = load_tri(tri_vidx, verts, i);
        if (t.finite) { double lo[3], hi[3]; tri_extent(t, lo, hi); const double ex = hi[0]-lo[0], ey = hi[1]-lo[1], ez = hi[2]-lo[2]; a = 2.0 * (ex*ey + ey*ez + ex*ez); c = 1.0; if (!isfinite(a)) { a = 0.0; c = 0.0; } }
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); c += __shfl_down(c, off); }
    if ((threadIdx.x & 63) == 0 && c > 0.0) { atomicAdd(&acc[0], a); atomicAdd(&acc[1], c); }
}
// pass 1: references per triangle (an invalid triangle keeps ONE, invalid, reference: it sorts last and gets no leaf)
__global__ void k_ref_count(const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, uint32_t n, double a_thr, int gain_rule, const uint32_t* __restrict__ boost, uint32_t* __restrict__ cnt, uint32_t* __restrict__ n_valid_refs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = 0, kv = 0;
    if (i < n) {
        const TriV t = load_tri(tri_vidx, verts, i);
        k = 1;
        if (t.finite) {
            double lo[3], hi[3]; tri_extent(t, lo, hi); k = ref_count(lo, hi, a_thr);
            if (k > 1 && gain_rule) {
                // ... but only where cutting pays (the rule of the host builder's first round, rts_sah.cpp): halving the box across
                // its longest axis must save a quarter of its surface -- long, thin, diagonal triangles; the near-equilateral
                // triangles of a uniformly tessellated surface keep ONE reference however large they are against the mean
                const double ex = hi[0]-lo[0], ey = hi[1]-lo[1], ez = hi[2]-lo[2];
                const int ax = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
                const double mid = 0.5 * lo[ax] + 0.5 * hi[ax];
                double l0[3], h0[3], l1[3], h1[3];
                tri_slab_box(t, lo, hi, ax, lo[ax], mid, l0, h0); tri_slab_box(t, lo, hi, ax, mid, hi[ax], l1, h1);
                const double a = box_area3(lo, hi), gain = a - (box_area3(l0, h0) + box_area3(l1, h1));
                if (!(gain > 0.25 * a)) k = 1;
            }
            if (boost && boost[i]) k = min(2u * k, 8u);                  // a crowded spot of the provisional tree (k_crowd)
            kv = k;
        }
        cnt[i] = k;
    }
    for (int off = 32; off > 0; off >>= 1) kv += __shfl_down(kv, off);
    if ((threadIdx.x & 63) == 0 && kv) atomicAdd(n_valid_refs, kv);
}

// Box = [rd(min - pad), ru(max + pad)] in f32, pad = 2^-22 of the largest coordinate magnitude (rts_sah.cpp: put_box).
// Invalid (non-finite) triangles get the INVERTED box, neutral under union; they sort last and stay outside the tree.
// Reference j of k: the part of the triangle between the planes lo + e j/k and lo + e (j+1)/k across the box's longest axis --
// its vertices inside the slab and the points where its edges cross the two planes (f64; the pad dwarfs their rounding).
__global__ void k_ref_boxes(const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off,
                            float* __restrict__ ref_box, uint32_t* __restrict__ ref_tri, uint32_t* __restrict__ bounds, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float mnx = 3.0e38f, mny = 3.0e38f, mnz = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f, mxz = -3.0e38f;
    if (i < n) {
        const TriV t = load_tri(tri_vidx, verts, i);
        const uint32_t k = cnt[i], base = off[i];
        double tlo[3], thi[3];
        if (t.finite) tri_extent(t, tlo, thi);
        int ax = 0;
        if (t.finite) { const double ex = thi[0]-tlo[0], ey = thi[1]-tlo[1], ez = thi[2]-tlo[2]; ax = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2); }
        for (uint32_t j = 0; j < k; j++) {
            float* o = ref_box + 6*(size_t)(base + j);
            ref_tri[base + j] = i;
            bool ok = false;
            if (t.finite) {
                double lo[3] = {tlo[0], tlo[1], tlo[2]}, hi[3] = {thi[0], thi[1], thi[2]};
                if (k > 1) {
                    const double e = thi[ax] - tlo[ax];
                    const double s0 = j == 0 ? tlo[ax] : tlo[ax] + e * ((double)j / (double)k), s1 = j + 1 == k ? thi[ax] : tlo[ax] + e * ((double)(j + 1) / (double)k);
                    tri_slab_box(t, tlo, thi, ax, s0, s1, lo, hi);
                }
                const double s = fmax(fmax(fmax(fabs(lo[0]), fabs(hi[0])), fmax(fabs(lo[1]), fabs(hi[1]))), fmax(fabs(lo[2]), fabs(hi[2])));
                const double pad = s * 2.384185791015625e-07 + 1e-30;
                o[0] = f32_down(lo[0] - pad); o[1] = f32_down(lo[1] - pad); o[2] = f32_down(lo[2] - pad);
                o[3] = f32_up(hi[0] + pad); o[4] = f32_up(hi[1] + pad); o[5] = f32_up(hi[2] + pad);
                ok = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(o[3]) && isfinite(o[4]) && isfinite(o[5]);
                if (ok) {
                    const float cx = (float)((lo[0] + hi[0]) * 0.5), cy = (float)((lo[1] + hi[1]) * 0.5), cz = (float)((lo[2] + hi[2]) * 0.5);
                    mnx = fminf(mnx, cx); mny = fminf(mny, cy); mnz = fminf(mnz, cz); mxx = fmaxf(mxx, cx); mxy = fmaxf(mxy, cy); mxz = fmaxf(mxz, cz);
                }
            }
            if (!ok) { o[0] = o[1] = o[2] = 3.0e38f; o[3] = o[4] = o[5] = -3.0e38f; }
        }
    }
    for (int off2 = 32; off2 > 0; off2 >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, off2)); mny = fminf(mny, __shfl_down(mny, off2)); mnz = fminf(mnz, __shfl_down(mnz, off2));
        mxx = fmaxf(mxx, __shfl_down(mxx, off2)); mxy = fmaxf(mxy, __shfl_down(mxy, off2)); mxz = fmaxf(mxz, __shfl_down(mxz, off2));
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
    if (b[0] > b[3]) { keys[i] = ~0ULL; return; }                           // invalid triangle: sorts last (valid codes have bit 63 clear; 2^63 - 1 IS a valid code)
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
__global__ void k_leaf_order(const uint32_t* __restrict__ sorted_ref, const uint32_t* __restrict__ ref_tri, uint32_t tri_base, uint32_t* __restrict__ leaf_prim, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) leaf_prim[i] = tri_base + ref_tri[sorted_ref[i]];
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

// BVH2 -> BVH4: record i holds up to four descendants of BVH2 node i -- its two children, then, while there is room, the
// INTERNAL one with the largest box opened into ITS two children (greedy by surface area: the child most likely to be hit
// is the one worth a wider record; a leaf child cannot be opened, so a fixed "open both children" pattern left records
// near the leaves half empty).  Child links become global: node_base + BVH2 index, ~(leaf_base + position); records of BVH2
// nodes that no record refers to are never visited.  Unused slots: the degenerate box lo = hi = 3e38 (rts_sah.cpp: new_node).
__global__ void k_collapse4(const Node2* __restrict__ nodes, RtsNode4* __restrict__ nodes4, int n_nodes, int32_t node_base, int32_t leaf_base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    int c[4]; float b[4][6]; int m = 0;
    auto take = [&](const Node2& nd) {
        c[m] = nd.c0; b[m][0] = nd.lo0x; b[m][1] = nd.lo0y; b[m][2] = nd.lo0z; b[m][3] = nd.hi0x; b[m][4] = nd.hi0y; b[m][5] = nd.hi0z; m++;
        c[m] = nd.c1; b[m][0] = nd.lo1x; b[m][1] = nd.lo1y; b[m][2] = nd.lo1z; b[m][3] = nd.hi1x; b[m][4] = nd.hi1y; b[m][5] = nd.hi1z; m++;
    };
    take(nodes[i]);
    while (m < 4) {
        int best = -1; float best_a = -1.0f;
        for (int k = 0; k < m; k++) {
            if (c[k] < 0) continue;                                                     // a leaf
            const float ex = b[k][3] - b[k][0], ey = b[k][4] - b[k][1], ez = b[k][5] - b[k][2];
            const float a = ex * ey + ey * ez + ex * ez;
            if (a > best_a) { best_a = a; best = k; }
        }
        if (best < 0) break;
        const Node2 ch = nodes[c[best]];
        c[best] = c[m - 1]; for (int q = 0; q < 6; q++) b[best][q] = b[m - 1][q];       // remove it (swap with the last) ...
        m--;
        take(ch);                                                                       // ... and append its two children
    }
    RtsNode4 o;
    for (int k = 0; k < 4; k++) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = 3.0e38f; o.child[k] = 0x7fffffff; o.pad[k] = 0; }
    for (int k = 0; k < m; k++) {
        o.lox[k] = b[k][0]; o.loy[k] = b[k][1]; o.loz[k] = b[k][2]; o.hix[k] = b[k][3]; o.hiy[k] = b[k][4]; o.hiz[k] = b[k][5];
        o.child[k] = c[k] >= 0 ? c[k] + node_base : ~(~c[k] + leaf_base);
    }
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

// --------------------------------------------------------------------------- top-down binned SAH over the references (device)
// The quality builder (default of the device path; RTS_DEVICE_TREE=lbvh selects the Morton / Karras tree above): the same
// algorithm as the host builder of rts_sah.cpp -- per node the centroid bounds, 16 bins on each of the three axes holding the
// union of the reference boxes and their count, the split of least  A_left n_left + A_right n_right, one reference per leaf --
// run level by level over ALL open nodes at once:
//   k_sah_bounds : centroid bounds (and the box) of every open node            per position, wave-aggregated atomics
//   k_sah_bins   : the 3 x 16 bins of every open node                          per position; a block whose positions all belong to
//                                                                              one node bins into LDS and flushes once
//   k_sah_split  : best split, child boxes, child links, the next level's open nodes       per open node
//   k_sah_scatter: partition of the node's positions (children are contiguous ranges)       per position
// A node's references occupy a contiguous range of the position array; a leaf is ~position, so the final position order IS
// the leaf order.  Output: the Node2 array the collapse to 4-wide records reads (root = node 0), child boxes exact (unions
// of bins hold every reference box).  A node whose references all have the same centroid is cut in the middle and both
// children get the node's box (conservative; such references are duplicates).  ~20-30 levels for 10^5-10^6 references.
#define SAH_BINS 16
struct SahWork { int32_t begin, end, node, pad; };
struct SahSplit { int32_t axis, bin, n_left, wl, wr, pad0, pad1, pad2; float lo, scale; };      // axis < 0: cut in the middle (wl / wr: child work index or -1)
#define SAH_BIN_WORDS (3 * SAH_BINS * 7)

__device__ __forceinline__ void ref_centroid(const float* __restrict__ b, float c[3]) { c[0] = (b[0] + b[3]) * 0.5f; c[1] = (b[1] + b[4]) * 0.5f; c[2] = (b[2] + b[5]) * 0.5f; }

__global__ void k_sah_init_bins(uint32_t* __restrict__ bins, uint32_t* __restrict__ cb, uint32_t n_work)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n_work * SAH_BIN_WORDS) { const uint32_t k = (uint32_t)(i % 7); bins[i] = k < 3 ? 0xffffffffu : 0u; }      // [lo x3 (ordered, min) | hi x3 (max) | count]
    if (i < (size_t)n_work * 12) { const uint32_t k = (uint32_t)(i % 12); cb[i] = (k % 6) < 3 ? 0xffffffffu : 0u; }               // [centroid lo3 hi3 | box lo3 hi3]
}

__global__ void __launch_bounds__(256) k_sah_bounds(const float* __restrict__ ref_box, const uint32_t* __restrict__ idx, const int32_t* __restrict__ work_of, uint32_t* __restrict__ cb, uint32_t n)
{
    __shared__ uint32_t s_cb[12];
    __shared__ int s_first, s_last;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = p < n ? work_of[p] : -1;
    if (threadIdx.x == 0) { s_first = work_of[blockIdx.x * blockDim.x]; const uint32_t lastp = min((blockIdx.x + 1) * blockDim.x, n) - 1; s_last = work_of[lastp]; }
    if (threadIdx.x < 12) s_cb[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    const bool one_node = s_first >= 0 && s_first == s_last;                     // the whole block is one node's (top levels): LDS first, twelve atomics per block
    if (w >= 0) {
        const float* b = ref_box + 6 * (size_t)idx[p];
        float c[3]; ref_centroid(b, c);
        uint32_t* o = one_node ? s_cb : cb + 12 * (size_t)w;
        for (int k = 0; k < 3; k++) { atomicMin(o + k, f2ord(c[k])); atomicMax(o + 3 + k, f2ord(c[k])); atomicMin(o + 6 + k, f2ord(b[k])); atomicMax(o + 9 + k, f2ord(b[3 + k])); }
    }
    if (one_node) {
        __syncthreads();
        if (threadIdx.x < 12) { uint32_t* g = cb + 12 * (size_t)s_first + threadIdx.x; if ((threadIdx.x % 6) < 3) atomicMin(g, s_cb[threadIdx.x]); else atomicMax(g, s_cb[threadIdx.x]); }
    }
}

__global__ void __launch_bounds__(256) k_sah_bins(const float* __restrict__ ref_box, const uint32_t* __restrict__ idx, const int32_t* __restrict__ work_of, const uint32_t* __restrict__ cb,
                                                  uint32_t* __restrict__ bins, uint32_t n)
{
    __shared__ uint32_t s_bins[SAH_BIN_WORDS];
    __shared__ int s_first, s_last;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = p < n ? work_of[p] : -1;
    if (threadIdx.x == 0) { s_first = work_of[blockIdx.x * blockDim.x]; const uint32_t lastp = min((blockIdx.x + 1) * blockDim.x, n) - 1; s_last = work_of[lastp]; }
    __syncthreads();
    const bool one_node = s_first >= 0 && s_first == s_last;                     // (positions of a node are contiguous: first == last => the whole block)
    if (one_node) { for (uint32_t k = threadIdx.x; k < SAH_BIN_WORDS; k += blockDim.x) s_bins[k] = (k % 7) < 3 ? 0xffffffffu : 0u; }
    __syncthreads();
    if (w >= 0) {
        const float* b = ref_box + 6 * (size_t)idx[p];
        float c[3]; ref_centroid(b, c);
        const uint32_t* cw = cb + 12 * (size_t)w;
        uint32_t* dst = one_node ? s_bins : bins + (size_t)w * SAH_BIN_WORDS;
        for (int ax = 0; ax < 3; ax++) {
            const float lo = ord2f(cw[ax]), hi = ord2f(cw[3 + ax]), ext = hi - lo;
            if (!(ext > 0.0f)) continue;
            int bi = (int)((c[ax] - lo) * ((float)SAH_BINS * 0.999999f / ext)); bi = bi < 0 ? 0 : (bi >= SAH_BINS ? SAH_BINS - 1 : bi);
            uint32_t* o = dst + (ax * SAH_BINS + bi) * 7;
            for (int k = 0; k < 3; k++) { atomicMin(o + k, f2ord(b[k])); atomicMax(o + 3 + k, f2ord(b[3 + k])); }
            atomicAdd(o + 6, 1u);
        }
    }
    if (one_node) {
        __syncthreads();
        uint32_t* g = bins + (size_t)s_first * SAH_BIN_WORDS;
        for (uint32_t k = threadIdx.x; k < SAH_BIN_WORDS; k += blockDim.x) {
            const uint32_t v = s_bins[k], kk = k % 7;
            if (kk < 3) { if (v != 0xffffffffu) atomicMin(g + k, v); } else if (kk < 6) { if (v != 0u) atomicMax(g + k, v); } else if (v) atomicAdd(g + k, v);
        }
    }
}

__device__ __forceinline__ float sah_area(const float b[6]) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return (dx < 0.0f || dy < 0.0f || dz < 0.0f) ? 0.0f : 2.0f * (dx*dy + dy*dz + dz*dx); }

__global__ void k_sah_split(const SahWork* __restrict__ work, uint32_t n_work, const uint32_t* __restrict__ cb, const uint32_t* __restrict__ bins, SahSplit* __restrict__ split,
                            Node2* __restrict__ nodes, SahWork* __restrict__ next, uint32_t* __restrict__ counters /* [0] next node id, [1] next work count */)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_work) return;
    const SahWork wk = work[w];
    const int n = wk.end - wk.begin;
    const uint32_t* cw = cb + 12 * (size_t)w;
    float best = __builtin_inff(); int best_axis = -1, best_bin = -1, best_nl = 0; float bl[6], br[6];
    for (int ax = 0; ax < 3; ax++) {
        const float lo = ord2f(cw[ax]), hi = ord2f(cw[3 + ax]);
        if (!(hi - lo > 0.0f)) continue;
        const uint32_t* B = bins + (size_t)w * SAH_BIN_WORDS + ax * SAH_BINS * 7;
        float ra[SAH_BINS]; uint32_t rc[SAH_BINS];
        float acc[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}; uint32_t c = 0;
        for (int b = SAH_BINS - 1; b > 0; b--) {
            const uint32_t* o = B + b * 7;
            if (o[6]) { for (int k = 0; k < 3; k++) { acc[k] = fminf(acc[k], ord2f(o[k])); acc[3 + k] = fmaxf(acc[3 + k], ord2f(o[3 + k])); } c += o[6]; }
            ra[b] = sah_area(acc); rc[b] = c;
        }
        for (int k = 0; k < 6; k++) acc[k] = k < 3 ? 3.0e38f : -3.0e38f;
        c = 0;
        for (int b = 0; b < SAH_BINS - 1; b++) {
            const uint32_t* o = B + b * 7;
            if (o[6]) { for (int k = 0; k < 3; k++) { acc[k] = fminf(acc[k], ord2f(o[k])); acc[3 + k] = fmaxf(acc[3 + k], ord2f(o[3 + k])); } c += o[6]; }
            if (c == 0 || rc[b + 1] == 0) continue;
            const float cost = sah_area(acc) * (float)c + ra[b + 1] * (float)rc[b + 1];
            if (cost < best) { best = cost; best_axis = ax; best_bin = b; best_nl = (int)c; }
        }
    }
    SahSplit sp; sp.axis = best_axis; sp.bin = best_bin; sp.pad0 = sp.pad1 = sp.pad2 = 0; sp.lo = 0.0f; sp.scale = 0.0f;
    if (best_axis >= 0) {
        const float lo = ord2f(cw[best_axis]), hi = ord2f(cw[3 + best_axis]);
        sp.lo = lo; sp.scale = (float)SAH_BINS * 0.999999f / (hi - lo);
        const uint32_t* B = bins + (size_t)w * SAH_BIN_WORDS + best_axis * SAH_BINS * 7;
        for (int k = 0; k < 6; k++) { bl[k] = br[k] = k < 3 ? 3.0e38f : -3.0e38f; }
        for (int b = 0; b < SAH_BINS; b++) {
            const uint32_t* o = B + b * 7;
            if (!o[6]) continue;
            float* d = b <= best_bin ? bl : br;
            for (int k = 0; k < 3; k++) { d[k] = fminf(d[k], ord2f(o[k])); d[3 + k] = fmaxf(d[3 + k], ord2f(o[3 + k])); }
        }
    } else {                                                                  // every centroid coincides: cut in the middle, both children take the node's box
        best_nl = n / 2;
        for (int k = 0; k < 3; k++) { bl[k] = br[k] = ord2f(cw[6 + k]); bl[3 + k] = br[3 + k] = ord2f(cw[9 + k]); }
    }
    sp.n_left = best_nl;
    const int nl = best_nl, nr = n - best_nl;
    int cl, cr; sp.wl = -1; sp.wr = -1;
    // node numbers in DEPTH-FIRST PREORDER, without a counter: a subtree over m references has m - 1 nodes, so the left child
    // is node + 1 and the right child node + n_left -- the records of a subtree are contiguous in memory, like its leaves
    // (the host builder lays its tree out the same way; numbered level by level the same tree traced 6 % slower on C3)
    if (nl > 1) { cl = wk.node + 1; atomicAdd(&counters[0], 1u); sp.wl = (int)atomicAdd(&counters[1], 1u); SahWork q; q.begin = wk.begin; q.end = wk.begin + nl; q.node = cl; q.pad = 0; next[sp.wl] = q; }
    else cl = ~wk.begin;
    if (nr > 1) { cr = wk.node + nl; atomicAdd(&counters[0], 1u); sp.wr = (int)atomicAdd(&counters[1], 1u); SahWork q; q.begin = wk.begin + nl; q.end = wk.end; q.node = cr; q.pad = 0; next[sp.wr] = q; }
    else cr = ~(wk.end - 1);
    split[w] = sp;
    Node2 nd; nd.lo0x = bl[0]; nd.lo0y = bl[1]; nd.lo0z = bl[2]; nd.hi0x = bl[3]; nd.hi0y = bl[4]; nd.hi0z = bl[5];
    nd.lo1x = br[0]; nd.lo1y = br[1]; nd.lo1z = br[2]; nd.hi1x = br[3]; nd.hi1y = br[4]; nd.hi1z = br[5]; nd.c0 = cl; nd.c1 = cr; nd.pad0 = 0; nd.pad1 = 0;
    nodes[wk.node] = nd;
}

__global__ void __launch_bounds__(256) k_sah_scatter(const float* __restrict__ ref_box, const uint32_t* __restrict__ idx, const int32_t* __restrict__ work_of, const SahWork* __restrict__ work,
                              const SahSplit* __restrict__ split, uint32_t* __restrict__ fill /* [n_work][2] */, uint32_t* __restrict__ idx_out, int32_t* __restrict__ work_out, uint32_t n)
{
    // Destinations are handed out by two counters per node (left, right).  At the top levels a node owns whole blocks of
    // positions: such a block counts its lefts and rights in LDS and draws ONE range per side (per position the 3 x 10^5
    // references of a C3 mesh queued behind two addresses: 0.64 ms per level, 37 ms of a 52 ms build).
    __shared__ uint32_t s_cnt[2], s_base[2];
    __shared__ int s_first, s_last;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = p < n ? work_of[p] : -1;
    if (threadIdx.x == 0) { s_first = work_of[blockIdx.x * blockDim.x]; const uint32_t lastp = min((blockIdx.x + 1) * blockDim.x, n) - 1; s_last = work_of[lastp]; s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    const bool one_node = s_first >= 0 && s_first == s_last;
    uint32_t r = 0; bool left = false, sorted = false; uint32_t local = 0;
    SahWork wk; SahSplit sp;
    if (w >= 0) {
        wk = work[w]; sp = split[w]; r = idx[p];
        if (sp.axis >= 0) {
            const float* b = ref_box + 6 * (size_t)r;
            const float c = (b[sp.axis] + b[3 + sp.axis]) * 0.5f;
            int bi = (int)((c - sp.lo) * sp.scale); bi = bi < 0 ? 0 : (bi >= SAH_BINS ? SAH_BINS - 1 : bi);
            left = bi <= sp.bin; sorted = true;
            if (one_node) local = atomicAdd(&s_cnt[left ? 0 : 1], 1u);
        } else left = (int)p < wk.begin + sp.n_left;
    }
    if (one_node) {
        __syncthreads();
        if (threadIdx.x < 2 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&fill[2 * (size_t)s_first + threadIdx.x], s_cnt[threadIdx.x]);
        __syncthreads();
    }
    if (p >= n) return;
    if (w < 0) { idx_out[p] = idx[p]; work_out[p] = -1; return; }                 // already a leaf: stays where it is
    uint32_t dest = p;
    if (sorted) {
        const uint32_t off = one_node ? s_base[left ? 0 : 1] + local : atomicAdd(&fill[2 * (size_t)w + (left ? 0 : 1)], 1u);
        dest = left ? (uint32_t)wk.begin + off : (uint32_t)(wk.begin + sp.n_left) + off;
    }
    idx_out[dest] = r; work_out[dest] = left ? sp.wl : sp.wr;
}

// Crowded spots of a provisional tree (the second and third rounds of the host builder's reference splitting, rts_sah.cpp):
// box area saved says nothing about HOW MANY references cover the same spot -- a ray through the hub of a fan of N slivers
// tests all N of them, however small they are.  Per reference: how many leaf boxes of the tree contain its centre; a dozen
// overlapping neighbours is normal on a curved surface, beyond that the reference's triangle is cut into twice as many slabs
// (if halving its box saves a tenth of its surface at all) and the tree is built again.
__global__ void k_crowd(const Node2* __restrict__ nodes, const float* __restrict__ ref_box, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ ref_tri,
                        const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, uint32_t nv, uint32_t* __restrict__ boost, uint32_t* __restrict__ n_boosted)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nv) return;
    const uint32_t r = idx[p];
    const float* b = ref_box + 6 * (size_t)r;
    const float cx = (b[0] + b[3]) * 0.5f, cy = (b[1] + b[4]) * 0.5f, cz = (b[2] + b[5]) * 0.5f;
    int stack[64]; int sp = 0; stack[sp++] = 0;
    int count = 0, steps = 0;
    while (sp > 0 && count < 64 && steps++ < 4096) {
        const Node2 nd = nodes[stack[--sp]];
        const bool in0 = cx >= nd.lo0x && cx <= nd.hi0x && cy >= nd.lo0y && cy <= nd.hi0y && cz >= nd.lo0z && cz <= nd.hi0z;
        const bool in1 = cx >= nd.lo1x && cx <= nd.hi1x && cy >= nd.lo1y && cy <= nd.hi1y && cz >= nd.lo1z && cz <= nd.hi1z;
        if (in0) { if (nd.c0 < 0) count++; else if (sp < 64) stack[sp++] = nd.c0; }
        if (in1) { if (nd.c1 < 0) count++; else if (sp < 64) stack[sp++] = nd.c1; }
    }
    if (count < 12) return;
    const uint32_t tri = ref_tri[r];
    const TriV t = load_tri(tri_vidx, verts, tri);
    if (!t.finite) return;
    double lo[3], hi[3]; tri_extent(t, lo, hi);
    const double ex = hi[0]-lo[0], ey = hi[1]-lo[1], ez = hi[2]-lo[2];
    const int ax = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    const double mid = 0.5 * lo[ax] + 0.5 * hi[ax];
    double l0[3], h0[3], l1[3], h1[3];
    tri_slab_box(t, lo, hi, ax, lo[ax], mid, l0, h0); tri_slab_box(t, lo, hi, ax, mid, hi[ax], l1, h1);
    const double a = box_area3(lo, hi);
    if (!(a - (box_area3(l0, h0) + box_area3(l1, h1)) > 0.10 * a)) return;
    if (atomicExch(&boost[tri], 1u) == 0u) atomicAdd(n_boosted, 1u);
}

// Compaction of the 4-wide records: k_collapse4 writes one record per BINARY node, and about half of them are never referred to
// (their node was opened into its parent's record).  Left in place they double the footprint of the hierarchy and halve the
// density of every cache line and page the walk touches (C3: 39 MB against the host builder's 19 MB; the same tree traced
// 5-8 % slower).  Reachable records are marked from the root, level by level, numbered by a scan -- which keeps the
// depth-first order -- and copied out with their child links renumbered.
__global__ void k_reach_step(const RtsNode4* __restrict__ rec, uint32_t* __restrict__ reach, uint32_t n, uint32_t* __restrict__ changed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || reach[i] != 1u) return;
    reach[i] = 2u;                                                            // expanded
    for (int k = 0; k < 4; k++) { const int32_t c = rec[i].child[k]; if (c >= 0 && c != 0x7fffffff && (uint32_t)c < n && reach[c] == 0u) { reach[c] = 1u; *changed = 1u; } }
}
__global__ void k_reach_flag(const uint32_t* __restrict__ reach, uint32_t* __restrict__ flag, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) flag[i] = reach[i] ? 1u : 0u; }
__global__ void k_compact4(const RtsNode4* __restrict__ rec, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ newid, uint32_t n, RtsNode4* __restrict__ out, int32_t node_base)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    RtsNode4 o = rec[i];
    for (int k = 0; k < 4; k++) { const int32_t c = o.child[k]; if (c >= 0 && c != 0x7fffffff) o.child[k] = (int32_t)newid[c] + node_base; }
    out[newid[i]] = o;
}

__global__ void k_iota(uint32_t* __restrict__ a, int32_t* __restrict__ w, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { a[i] = i; w[i] = 0; } }
// references with the sentinel box (non-finite triangles) moved behind the valid ones: stable compaction by a flag scan is
// not needed -- the caller sorts by (valid ? 0 : 1) with the radix sort it already has -- this kernel writes that key
__global__ void k_valid_key(const float* __restrict__ ref_box, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float* b = ref_box + 6 * (size_t)i; keys[i] = (b[0] > b[3]) ? 1ULL : 0ULL; vals[i] = i; }
}

}  // namespace

// vidx: [n_prims][3] GLOBAL vertex indices (host copy of ns->d_tri_vidx); mh: per-mesh slices.  Fills ns->d_nodes4,
// ns->d_leaf_prim, ns->blas, ns->n_nodes, ns->n_leaves.  Uses the handle's stream; temporaries are freed before returning.
// split_budget: aimed-at extra references per triangle (0: one reference per triangle); the threshold of ref_count is the
// mesh's mean box area / (1 + budget)^2, i.e. an average triangle gets about 1 + budget references.
int rts_lbvh_build_device(RtsContext* c, RtsScene* ns, const std::vector<uint32_t>& vidx, const std::vector<RtsMeshHost>& mh, double split_budget)
{
    hipStream_t st = c->stream;
    const uint32_t n_targets = (uint32_t)mh.size();
    ns->blas.assign(n_targets, RtsBlasInfo{});
    // host pass: the f64 bounds of every mesh over its finite triangles (for the per-pulse placement constants)
    std::vector<double> hv(3 * (size_t)ns->n_verts);
    if (ns->n_verts) RTS_HIP(hipMemcpy(hv.data(), ns->d_verts_local.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost));
    uint32_t n_max = 0;
    for (uint32_t t = 0; t < n_targets; t++) {
        RtsBlasInfo& b = ns->blas[t]; b.root = -1; b.depth = 64;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t nv = 0;
        for (uint32_t i = 0; i < mh[t].n_tris; i++) {
            bool finite = true;
            for (int k = 0; k < 3; k++) { const double* p = &hv[3 * (size_t)vidx[3 * ((size_t)mh[t].tri_base + i) + k]]; finite = finite && std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }
            if (!finite) continue;
            nv++;
            for (int k = 0; k < 3; k++) { const double* p = &hv[3 * (size_t)vidx[3 * ((size_t)mh[t].tri_base + i) + k]]; for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
        }
        for (int a = 0; a < 3; a++) { b.lo[a] = nv ? lo[a] : 0; b.hi[a] = nv ? hi[a] : 0; b.max_abs = std::max(b.max_abs, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]))); }
        n_max = std::max(n_max, mh[t].n_tris);
    }
    ns->n_nodes = 0; ns->n_leaves = 0;
    if (n_max == 0) { RTS_HIP(ns->d_nodes4.reserve(1)); RTS_HIP(ns->d_leaf_prim.reserve(1)); return RTS_OK; }

    // ---- temporaries of both passes (grown on demand)
    DevBuf<float> d_prim_box; DevBuf<uint64_t> d_keys, d_keys_sorted; DevBuf<uint32_t> d_vals, d_vals_sorted, d_bounds, d_flags, d_ref_tri; DevBuf<int32_t> d_parent, d_leaf_parent;
    DevBuf<Node2> d_nodes2; DevBuf<int2> d_range; DevBuf<char> d_tmp;
    struct Free { DevBuf<float>& a; DevBuf<uint64_t>& b; DevBuf<uint64_t>& b2; DevBuf<uint32_t>& c1; DevBuf<uint32_t>& c2; DevBuf<uint32_t>& c3; DevBuf<uint32_t>& c4; DevBuf<uint32_t>& c5; DevBuf<int32_t>& d1; DevBuf<int32_t>& d2; DevBuf<Node2>& e; DevBuf<int2>& f; DevBuf<char>& g;
                  ~Free() { a.release(); b.release(); b2.release(); c1.release(); c2.release(); c3.release(); c4.release(); c5.release(); d1.release(); d2.release(); e.release(); f.release(); g.release(); } }
        free_all{d_prim_box, d_keys, d_keys_sorted, d_vals, d_vals_sorted, d_bounds, d_flags, d_ref_tri, d_parent, d_leaf_parent, d_nodes2, d_range, d_tmp};
    size_t tmp = 0;
    auto ensure_tmp = [&](uint32_t r) -> int {
        RTS_HIP(d_prim_box.reserve(6 * (size_t)r)); RTS_HIP(d_keys.reserve(r)); RTS_HIP(d_keys_sorted.reserve(r)); RTS_HIP(d_vals.reserve(r)); RTS_HIP(d_vals_sorted.reserve(r)); RTS_HIP(d_ref_tri.reserve(r));
        RTS_HIP(d_bounds.reserve(8)); RTS_HIP(d_flags.reserve(r)); RTS_HIP(d_parent.reserve(r)); RTS_HIP(d_leaf_parent.reserve(r)); RTS_HIP(d_nodes2.reserve(r)); RTS_HIP(d_range.reserve(r));
        size_t need = 0;
        RTS_HIP(rocprim::radix_sort_pairs(nullptr, need, d_keys.p, d_keys_sorted.p, d_vals.p, d_vals_sorted.p, r, 0, 64, st));
        RTS_HIP(d_tmp.reserve(need)); tmp = std::max(tmp, need);
        return RTS_OK;
    };
    bool sah_tree = true;                                          // top-down binned SAH (default) or the Morton / Karras tree
    { const char* e = getenv("RTS_DEVICE_TREE"); if (e) sah_tree = strcmp(e, "lbvh") != 0; }
    // (crowding rounds -- a provisional tree, twice the slabs for triangles in crowded spots, the host builder's second stage: built,
    // measured on C3 at 0 / 1 / 2 rounds: 0.812 / 0.810 / 0.830 ms, and they double the build time: off unless RTS_CROWD_ROUNDS asks)
    int crowd_rounds = 0; { const char* e = getenv("RTS_CROWD_ROUNDS"); if (e) crowd_rounds = std::max(0, std::min(3, atoi(e))); }
    DevBuf<uint32_t> d_sah_bins, d_sah_cb, d_sah_fill, d_boost; DevBuf<SahSplit> d_sah_split; DevBuf<SahWork> d_sah_work_a, d_sah_work_b;
    DevBuf<RtsNode4> d_nodes4_tmp; DevBuf<uint32_t> d_reach, d_newid, d_rflag;
    struct FreeC { DevBuf<RtsNode4>& a; DevBuf<uint32_t>& b; DevBuf<uint32_t>& c1; DevBuf<uint32_t>& d; ~FreeC() { a.release(); b.release(); c1.release(); d.release(); } } free_c{d_nodes4_tmp, d_reach, d_newid, d_rflag};
    struct FreeSah { DevBuf<uint32_t>& a; DevBuf<uint32_t>& b; DevBuf<uint32_t>& c1; DevBuf<uint32_t>& c2; DevBuf<SahSplit>& d; DevBuf<SahWork>& e; DevBuf<SahWork>& f; ~FreeSah() { a.release(); b.release(); c1.release(); c2.release(); d.release(); e.release(); f.release(); } }
        free_sah{d_sah_bins, d_sah_cb, d_sah_fill, d_boost, d_sah_split, d_sah_work_a, d_sah_work_b};
    RTS_HIP(d_boost.reserve((size_t)ns->n_prims + 2)); RTS_HIP(hipMemsetAsync(d_boost.p, 0, sizeof(uint32_t) * ((size_t)ns->n_prims + 2), st));       // per triangle: 1 = twice the slabs (k_crowd); last word: a counter

    // the level loop of the binned-SAH builder over the nv valid references of the current mesh (boxes in d_prim_box):
    // leaves the binary tree in d_nodes2 (root 0, nv - 1 nodes) and the leaf order in d_vals_sorted
    auto sah_levels = [&](uint32_t nr, uint32_t nv) -> int {
        size_t tmp_n = tmp;
        k_valid_key<<<blocks_for(nr, 256), 256, 0, st>>>(d_prim_box.p, d_keys.p, d_vals.p, nr);
        RTS_HIP(rocprim::radix_sort_pairs(d_tmp.p, tmp_n, d_keys.p, d_keys_sorted.p, d_vals.p, d_vals_sorted.p, nr, 0, 1, st));     // valid references first (stable)
        uint32_t* idx_a = d_vals_sorted.p; uint32_t* idx_b = d_vals.p;
        int32_t* wo_a = d_parent.p; int32_t* wo_b = d_leaf_parent.p;                     // work-node index per position (the LBVH's link arrays are free in this mode)
        const uint32_t max_work = nv / 2 + 2;
        RTS_HIP(d_sah_bins.reserve((size_t)max_work * SAH_BIN_WORDS)); RTS_HIP(d_sah_cb.reserve((size_t)max_work * 12)); RTS_HIP(d_sah_fill.reserve((size_t)max_work * 2 + 2));
        RTS_HIP(d_sah_split.reserve(max_work)); RTS_HIP(d_sah_work_a.reserve(max_work)); RTS_HIP(d_sah_work_b.reserve(max_work));
        k_iota<<<blocks_for(nv, 256), 256, 0, st>>>(d_flags.p, wo_a, nv);            // (d_flags only as a dummy target for the iota; work_of := 0)
        SahWork w0; w0.begin = 0; w0.end = (int32_t)nv; w0.node = 0; w0.pad = 0;
        RTS_HIP(hipMemcpyAsync(d_sah_work_a.p, &w0, sizeof(w0), hipMemcpyHostToDevice, st));
        uint32_t* cnt = d_sah_fill.p + (size_t)max_work * 2;                           // [0] next node id, [1] open nodes of the next level
        uint32_t h_cnt[2] = {1u, 0u};
        RTS_HIP(hipMemcpyAsync(cnt, h_cnt, sizeof(h_cnt), hipMemcpyHostToDevice, st));
        SahWork* wk_a = d_sah_work_a.p; SahWork* wk_b = d_sah_work_b.p;
        uint32_t n_work = 1;
        for (int level = 0; n_work > 0 && level < 256; level++) {
            const size_t init_n = std::max((size_t)n_work * SAH_BIN_WORDS, (size_t)n_work * 12);
            k_sah_init_bins<<<blocks_for(init_n, 256), 256, 0, st>>>(d_sah_bins.p, d_sah_cb.p, n_work);
            RTS_HIP(hipMemsetAsync(d_sah_fill.p, 0, sizeof(uint32_t) * 2 * (size_t)n_work, st));
            RTS_HIP(hipMemsetAsync(cnt + 1, 0, sizeof(uint32_t), st));
            k_sah_bounds<<<blocks_for(nv, 256), 256, 0, st>>>(d_prim_box.p, idx_a, wo_a, d_sah_cb.p, nv);
            k_sah_bins<<<blocks_for(nv, 256), 256, 0, st>>>(d_prim_box.p, idx_a, wo_a, d_sah_cb.p, d_sah_bins.p, nv);
            k_sah_split<<<blocks_for(n_work, 64), 64, 0, st>>>(wk_a, n_work, d_sah_cb.p, d_sah_bins.p, d_sah_split.p, d_nodes2.p, wk_b, cnt);
            k_sah_scatter<<<blocks_for(nv, 256), 256, 0, st>>>(d_prim_box.p, idx_a, wo_a, wk_a, d_sah_split.p, d_sah_fill.p, idx_b, wo_b, nv);
            RTS_HIP(hipMemcpyAsync(h_cnt, cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st)); RTS_HIP(hipStreamSynchronize(st));
            n_work = h_cnt[1];
            if (n_work > max_work) { rts_set_error("rts_set_scene: device SAH builder: %u open nodes at level %d exceed the bound %u", n_work, level, max_work); return RTS_ERR_HIP; }
            std::swap(idx_a, idx_b); std::swap(wo_a, wo_b); std::swap(wk_a, wk_b);
        }
        if (n_work != 0 || h_cnt[0] != nv - 1) { rts_set_error("rts_set_scene: device SAH builder did not close (%u open nodes, %u of %u nodes)", n_work, h_cnt[0], nv - 1); return RTS_ERR_HIP; }
        if (idx_a != d_vals_sorted.p) RTS_HIP(hipMemcpyAsync(d_vals_sorted.p, idx_a, sizeof(uint32_t) * nv, hipMemcpyDeviceToDevice, st));
        return RTS_OK;
    };

    // ---- pass A: the split threshold of every mesh and its number of references (valid ones first after the sort)
    DevBuf<double> d_acc; DevBuf<uint32_t> d_cnt, d_off, d_nvr; DevBuf<char> d_scan_tmp;
    struct FreeA { DevBuf<double>& a; DevBuf<uint32_t>& b; DevBuf<uint32_t>& c1; DevBuf<uint32_t>& d; DevBuf<char>& e; ~FreeA() { a.release(); b.release(); c1.release(); d.release(); e.release(); } } free_a{d_acc, d_cnt, d_off, d_nvr, d_scan_tmp};
    RTS_HIP(d_acc.reserve(2)); RTS_HIP(d_cnt.reserve(n_max)); RTS_HIP(d_off.reserve(n_max)); RTS_HIP(d_nvr.reserve(1));
    size_t scan_tmp = 0;
    RTS_HIP(rocprim::exclusive_scan(nullptr, scan_tmp, d_cnt.p, d_off.p, 0u, n_max, rocprim::plus<uint32_t>(), st));
    RTS_HIP(d_scan_tmp.reserve(scan_tmp + 16));
    std::vector<double> a_thr(n_targets, 0.0); std::vector<uint32_t> n_refs(n_targets, 0), n_valid(n_targets, 0);
    int gain_rule = 1; { const char* e = getenv("RTS_REF_GAIN_RULE"); if (e) gain_rule = atoi(e) != 0; }
    auto count_refs = [&](uint32_t t) -> int {                       // cnt / off of mesh t (deterministic: pass B repeats it)
        const uint32_t n = mh[t].n_tris;
        const uint32_t* tv = ns->d_tri_vidx.p + 3 * (size_t)mh[t].tri_base;
        RTS_HIP(hipMemsetAsync(d_nvr.p, 0, sizeof(uint32_t), st));
        k_ref_count<<<blocks_for(n, 256), 256, 0, st>>>(tv, ns->d_verts_local.p, n, a_thr[t], gain_rule, d_boost.p + mh[t].tri_base, d_cnt.p, d_nvr.p);
        size_t tmp_n = scan_tmp;
        RTS_HIP(rocprim::exclusive_scan(d_scan_tmp.p, tmp_n, d_cnt.p, d_off.p, 0u, n, rocprim::plus<uint32_t>(), st));
        return RTS_OK;
    };
    auto read_counts = [&](uint32_t t) -> int {
        const uint32_t n = mh[t].n_tris;
        uint32_t last[2] = {0, 0}, nvr = 0;
        RTS_HIP(hipMemcpyAsync(&last[0], d_off.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RTS_HIP(hipMemcpyAsync(&last[1], d_cnt.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        RTS_HIP(hipMemcpyAsync(&nvr, d_nvr.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st)); RTS_HIP(hipStreamSynchronize(st));
        n_refs[t] = last[0] + last[1]; n_valid[t] = nvr;
        return RTS_OK;
    };
    static const uint32_t init_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    uint64_t node_total = 0, leaf_total = 0; uint32_t r_max = 0;
    for (uint32_t t = 0; t < n_targets; t++) {
        const uint32_t n = mh[t].n_tris;
        if (n == 0) continue;
        const uint32_t* tv = ns->d_tri_vidx.p + 3 * (size_t)mh[t].tri_base;
        RTS_HIP(hipMemsetAsync(d_acc.p, 0, 2 * sizeof(double), st));
        k_tri_area<<<blocks_for(n, 256), 256, 0, st>>>(tv, ns->d_verts_local.p, n, d_acc.p);
        double acc[2] = {0, 0};
        RTS_HIP(hipMemcpyAsync(acc, d_acc.p, sizeof(acc), hipMemcpyDeviceToHost, st)); RTS_HIP(hipStreamSynchronize(st));
        a_thr[t] = (split_budget > 0 && acc[1] > 0 && acc[0] > 0 && std::isfinite(acc[0])) ? (acc[0] / acc[1]) / ((1.0 + split_budget) * (1.0 + split_budget)) : 0.0;
        { int rc = count_refs(t); if (rc != RTS_OK) return rc; }
        { int rc = read_counts(t); if (rc != RTS_OK) return rc; }
        // crowded spots: a provisional tree over these references, the triangles behind crowded references get twice the
        // slabs, count again
        for (int round = 0; sah_tree && split_budget > 0 && round < crowd_rounds && n_valid[t] > 1; round++) {
            { int rc = ensure_tmp(n_refs[t]); if (rc != RTS_OK) return rc; }
            RTS_HIP(hipMemcpyAsync(d_bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, st));
            k_ref_boxes<<<blocks_for(n, 256), 256, 0, st>>>(tv, ns->d_verts_local.p, d_cnt.p, d_off.p, d_prim_box.p, d_ref_tri.p, d_bounds.p, n);
            { int rc = sah_levels(n_refs[t], n_valid[t]); if (rc != RTS_OK) return rc; }
            uint32_t* n_boosted = d_boost.p + ns->n_prims; uint32_t before = 0, after = 0;
            RTS_HIP(hipMemcpyAsync(&before, n_boosted, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            k_crowd<<<blocks_for(n_valid[t], 256), 256, 0, st>>>(d_nodes2.p, d_prim_box.p, d_vals_sorted.p, d_ref_tri.p, tv, ns->d_verts_local.p, n_valid[t], d_boost.p + mh[t].tri_base, n_boosted);
            RTS_HIP(hipMemcpyAsync(&after, n_boosted, sizeof(uint32_t), hipMemcpyDeviceToHost, st)); RTS_HIP(hipStreamSynchronize(st));
            if (getenv("RTS_DEBUG_BUILD")) fprintf(stderr, "[rts build] mesh %u round %d: %u triangles, %u references (%u valid), %u triangles boosted\n", t, round, n, n_refs[t], n_valid[t], after - before);
            if (after == before) break;
            { int rc = count_refs(t); if (rc != RTS_OK) return rc; }
            { int rc = read_counts(t); if (rc != RTS_OK) return rc; }
        }
        RtsBlasInfo& b = ns->blas[t];
        b.n_leaves = n_valid[t]; b.n_nodes = n_valid[t] == 0 ? 0 : std::max<uint32_t>(n_valid[t] - 1, 1);
        node_total += b.n_nodes; leaf_total += b.n_leaves; r_max = std::max(r_max, n_refs[t]);
    }
    if (node_total > 0x7ffffff0ULL || leaf_total > 0x7ffffff0ULL) { rts_set_error("rts_set_scene: too many references"); return RTS_ERR_UNSUPPORTED; }
    RTS_HIP(ns->d_nodes4.reserve((size_t)node_total + 1)); RTS_HIP(ns->d_leaf_prim.reserve((size_t)leaf_total + 1));
    ns->n_nodes = (uint32_t)node_total; ns->n_leaves = (uint32_t)leaf_total;
    if (r_max == 0) return RTS_OK;

    // ---- pass B: boxes, tree, 4-wide records -- per mesh, over its references
    { int rc = ensure_tmp(r_max); if (rc != RTS_OK) return rc; }
    int32_t node_base = 0, leaf_base = 0;
    for (uint32_t t = 0; t < n_targets; t++) {
        const uint32_t n = mh[t].n_tris, nr = n_refs[t], nv = n_valid[t];
        RtsBlasInfo& b = ns->blas[t];
        if (nv == 0) continue;
        RTS_HIP(hipMemcpyAsync(d_bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, st));
        const uint32_t* tv = ns->d_tri_vidx.p + 3 * (size_t)mh[t].tri_base;
        { int rc = count_refs(t); if (rc != RTS_OK) return rc; }
        k_ref_boxes<<<blocks_for(n, 256), 256, 0, st>>>(tv, ns->d_verts_local.p, d_cnt.p, d_off.p, d_prim_box.p, d_ref_tri.p, d_bounds.p, n);
        size_t tmp_n = tmp;
        if (sah_tree && nv > 1) {
            // ---- top-down binned SAH over the references (the quality tree)
            { int rc = sah_levels(nr, nv); if (rc != RTS_OK) return rc; }
            k_leaf_order<<<blocks_for(nv, 256), 256, 0, st>>>(d_vals_sorted.p, d_ref_tri.p, mh[t].tri_base, ns->d_leaf_prim.p + leaf_base, nv);
            const uint32_t n2 = nv - 1;
            RTS_HIP(d_nodes4_tmp.reserve(n2)); RTS_HIP(d_reach.reserve((size_t)n2 + 1)); RTS_HIP(d_newid.reserve(n2)); RTS_HIP(d_rflag.reserve(n2));
            k_collapse4<<<blocks_for(n2, 256), 256, 0, st>>>(d_nodes2.p, d_nodes4_tmp.p, (int)n2, 0, leaf_base);
            RTS_HIP(hipMemsetAsync(d_reach.p, 0, sizeof(uint32_t) * ((size_t)n2 + 1), st));
            { const uint32_t one = 1u; RTS_HIP(hipMemcpyAsync(d_reach.p, &one, sizeof(one), hipMemcpyHostToDevice, st)); }
            uint32_t* d_changed = d_reach.p + n2;
            for (int it = 0; it < 256; it++) {
                RTS_HIP(hipMemsetAsync(d_changed, 0, sizeof(uint32_t), st));
                k_reach_step<<<blocks_for(n2, 256), 256, 0, st>>>(d_nodes4_tmp.p, d_reach.p, n2, d_changed);
                uint32_t ch = 0; RTS_HIP(hipMemcpyAsync(&ch, d_changed, sizeof(ch), hipMemcpyDeviceToHost, st)); RTS_HIP(hipStreamSynchronize(st));
                if (!ch) break;
            }
            k_reach_flag<<<blocks_for(n2, 256), 256, 0, st>>>(d_reach.p, d_rflag.p, n2);
            { size_t need = 0; RTS_HIP(rocprim::exclusive_scan(nullptr, need, d_rflag.p, d_newid.p, 0u, n2, rocprim::plus<uint32_t>(), st));
              RTS_HIP(d_tmp.reserve(need + 16)); RTS_HIP(rocprim::exclusive_scan(d_tmp.p, need, d_rflag.p, d_newid.p, 0u, n2, rocprim::plus<uint32_t>(), st)); }
            uint32_t last_id = 0, last_flag = 0;
            RTS_HIP(hipMemcpyAsync(&last_id, d_newid.p + (n2 - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            RTS_HIP(hipMemcpyAsync(&last_flag, d_rflag.p + (n2 - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st)); RTS_HIP(hipStreamSynchronize(st));
            const uint32_t n4 = last_id + last_flag;                                   // reachable records
            k_compact4<<<blocks_for(n2, 256), 256, 0, st>>>(d_nodes4_tmp.p, d_rflag.p, d_newid.p, n2, ns->d_nodes4.p + node_base, node_base);
            RTS_HIP(hipGetLastError());
            b.root = node_base; b.n_nodes = n4;
            node_base += (int32_t)n4; leaf_base += (int32_t)b.n_leaves;
            RTS_HIP(hipStreamSynchronize(st));
            continue;
        }
        k_morton<<<blocks_for(nr, 256), 256, 0, st>>>(d_prim_box.p, d_bounds.p, d_keys.p, d_vals.p, nr);
        RTS_HIP(rocprim::radix_sort_pairs(d_tmp.p, tmp_n, d_keys.p, d_keys_sorted.p, d_vals.p, d_vals_sorted.p, nr, 0, 64, st));
        k_leaf_order<<<blocks_for(nv, 256), 256, 0, st>>>(d_vals_sorted.p, d_ref_tri.p, mh[t].tri_base, ns->d_leaf_prim.p + leaf_base, nv);
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
    ns->n_nodes = (uint32_t)node_base;                               // (reachable records only: the binned-SAH path compacts its trees)
    return RTS_OK;
}
