// rts_sah.cpp -- host builder of the static, target-space BVH4 of one mesh.
//
// The reference re-creates every target's world-space vertex buffer and marks its OptiX "Bvh" acceleration dirty
// each pulse (ray_tracer.cpp:966-1014, 1126-1130), so the closed-source builder runs per pulse.  Its targets are
// rigid, though: a pulse only changes the placement  world = R * local + position  (ray_tracer.cpp:993-1014).  The
// hierarchy is therefore built ONCE per mesh, in target space, when the scene is set: a binned-SAH binary tree
// (16 bins, all three axes, one triangle per leaf) collapsed into 4-wide nodes by repeatedly opening the child of
// largest surface area.  Per pulse only the leaf records (world-space f64 vertices for the exact test) are refreshed;
// the traversal kernel walks the tree with the ray mapped into target space (rts_trace.hip).
//
// Boxes are f32, rounded outward and padded by 2^-22 of their largest coordinate magnitude, so that the f32 slab test
// stays conservative with respect to the f64 triangle test (same budget as the reference's `bound` program plus pad,
// triangle_mesh.cu:204-233).  Triangles with a non-finite vertex get no leaf: they can never be hit.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include "rts_internal.h"

namespace {

struct Box { double lo[3], hi[3]; };
inline void box_empty(Box& b) { for (int k = 0; k < 3; k++) { b.lo[k] = std::numeric_limits<double>::infinity(); b.hi[k] = -std::numeric_limits<double>::infinity(); } }
inline void box_grow(Box& b, const Box& o) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
inline void box_grow_pt(Box& b, const double* p) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], p[k]); b.hi[k] = std::max(b.hi[k], p[k]); } }
inline double box_area(const Box& b) { const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2]; return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : 2.0 * (dx*dy + dy*dz + dz*dx); }

struct Node2 { Box box; int left, right; uint32_t prim; };       // leaf: left == -1, prim = local triangle index

struct Builder {
    const std::vector<Box>& pb; const std::vector<double>& cen;   // per (finite) primitive: box, centroid[3]
    std::vector<uint32_t>& idx;                                    // permutation being partitioned
    std::vector<Node2> nodes;
    enum { BINS = 16 };

    int build(uint32_t begin, uint32_t end, int depth = 0)
    {
        const int id = (int)nodes.size(); nodes.push_back(Node2());
        Box nb; box_empty(nb); Box cb; box_empty(cb);
        for (uint32_t i = begin; i < end; i++) { box_grow(nb, pb[idx[i]]); box_grow_pt(cb, &cen[3*(size_t)idx[i]]); }
        nodes[id].box = nb; nodes[id].left = nodes[id].right = -1; nodes[id].prim = idx[begin];
        const uint32_t n = end - begin;
        if (n == 1) return id;
        uint32_t mid = begin + n / 2; bool have = false;
        if (n == 2) { mid = begin + 1; have = true; }
        else if (depth < 80) {                                    // (deeper: plain halving, keeps the recursion and the traversal stack bounded)
            double best = std::numeric_limits<double>::infinity(); int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                const double lo = cb.lo[ax], ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0)) continue;
                Box bb[BINS]; uint32_t cnt[BINS];
                for (int b = 0; b < BINS; b++) { box_empty(bb[b]); cnt[b] = 0; }
                const double scale = BINS * (1.0 - 1e-12) / ext;
                for (uint32_t i = begin; i < end; i++) {
                    int b = (int)((cen[3*(size_t)idx[i] + ax] - lo) * scale); b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    cnt[b]++; box_grow(bb[b], pb[idx[i]]);
                }
                double right_area[BINS]; uint32_t right_cnt[BINS];
                Box acc; box_empty(acc); uint32_t c = 0;
                for (int b = BINS - 1; b > 0; b--) { box_grow(acc, bb[b]); c += cnt[b]; right_area[b] = box_area(acc); right_cnt[b] = c; }
                box_empty(acc); c = 0;
                for (int b = 0; b < BINS - 1; b++) {
                    box_grow(acc, bb[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    const double cost = box_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const double lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
                const double scale = BINS * (1.0 - 1e-12) / ext;
                auto it = std::partition(idx.begin() + begin, idx.begin() + end, [&](uint32_t p) {
                    int b = (int)((cen[3*(size_t)p + best_axis] - lo) * scale); b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    return b <= best_bin; });
                mid = (uint32_t)(it - idx.begin());
                have = mid > begin && mid < end;
            }
        }
        if (!have) {                                              // coincident centroids: split the index range in half
            mid = begin + n / 2;
            std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end);
        }
        const int l = build(begin, mid, depth + 1); const int r = build(mid, end, depth + 1);
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
};

inline float f32_dn(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
inline float f32_upw(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }

void put_box(RtsNode4& o, int k, const Box& b)
{
    double s = 0; for (int a = 0; a < 3; a++) s = std::max(s, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    const double pad = s * 2.384185791015625e-07 + 1e-30;
    o.lox[k] = f32_dn(b.lo[0] - pad); o.loy[k] = f32_dn(b.lo[1] - pad); o.loz[k] = f32_dn(b.lo[2] - pad);
    o.hix[k] = f32_upw(b.hi[0] + pad); o.hiy[k] = f32_upw(b.hi[1] + pad); o.hiz[k] = f32_upw(b.hi[2] + pad);
    const bool ok = std::isfinite(o.lox[k]) && std::isfinite(o.loy[k]) && std::isfinite(o.loz[k]) && std::isfinite(o.hix[k]) && std::isfinite(o.hiy[k]) && std::isfinite(o.hiz[k]);
    if (!ok) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = -3.0e38f; }   // coordinates beyond f32: never hit
}

}  // namespace

// verts: [n_verts][3] target-space vertices of the mesh; tris: [n_tris][3] indices into verts.
// Appends the mesh's nodes to `nodes` (child links are indices into that shared array) and its leaf order to
// `leaf_prim` (LOCAL triangle index per leaf slot; a leaf link is ~slot, slot counted over the shared array).
int rts_sah_build(const double* verts, const uint32_t* tris, uint32_t n_tris, std::vector<RtsNode4>& nodes, std::vector<uint32_t>& leaf_prim, RtsBlasInfo& out)
{
    out.root = -1; out.n_nodes = 0; out.n_leaves = 0; out.depth = 0;
    for (int k = 0; k < 3; k++) { out.lo[k] = 0; out.hi[k] = 0; }
    out.max_abs = 0;
    std::vector<Box> pb; std::vector<double> cen; std::vector<uint32_t> prim_of;
    pb.reserve(n_tris); cen.reserve(3*(size_t)n_tris); prim_of.reserve(n_tris);
    for (uint32_t i = 0; i < n_tris; i++) {
        Box b; box_empty(b); bool finite = true;
        for (int k = 0; k < 3; k++) {
            const double* p = verts + 3*(size_t)tris[3*(size_t)i + k];
            finite = finite && std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
            box_grow_pt(b, p);
        }
        if (!finite) continue;
        pb.push_back(b); prim_of.push_back(i);
        for (int k = 0; k < 3; k++) cen.push_back(0.5 * b.lo[k] + 0.5 * b.hi[k]);
    }
    const uint32_t n = (uint32_t)pb.size();
    if (n == 0) return RTS_OK;
    std::vector<uint32_t> idx(n); for (uint32_t i = 0; i < n; i++) idx[i] = i;
    Builder B{pb, cen, idx, {}};
    B.nodes.reserve(2*(size_t)n);
    B.build(0, n);
    const std::vector<Node2>& N = B.nodes;

    // ---- collapse to 4-wide nodes, depth first (node and leaf slots in visiting order)
    const int32_t node_base = (int32_t)nodes.size();
    struct Item { int n2; int32_t n4; int depth; };
    std::vector<Item> stack;
    auto new_node = [&]() { RtsNode4 o; for (int k = 0; k < 4; k++) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = -3.0e38f; o.child[k] = 0x7fffffff; o.pad[k] = 0; }
                            nodes.push_back(o); return (int32_t)nodes.size() - 1; };
    const int32_t root4 = new_node();
    stack.push_back(Item{0, root4, 1});
    int max_depth = 1;
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        max_depth = std::max(max_depth, it.depth);
        int kids[4]; int m = 0;
        if (N[it.n2].left < 0) kids[m++] = it.n2;                 // single-triangle mesh: the root itself is the leaf
        else { kids[m++] = N[it.n2].left; kids[m++] = N[it.n2].right; }
        while (m < 4) {                                           // open the internal child of largest surface area
            int pick = -1; double best = -1;
            for (int k = 0; k < m; k++) if (N[kids[k]].left >= 0) { const double a = box_area(N[kids[k]].box); if (a > best) { best = a; pick = k; } }
            if (pick < 0) break;
            const int c = kids[pick]; kids[pick] = N[c].left; kids[m++] = N[c].right;
        }
        int32_t refs[4];
        for (int k = 0; k < m; k++) {
            if (N[kids[k]].left < 0) { refs[k] = ~(int32_t)leaf_prim.size(); leaf_prim.push_back(prim_of[N[kids[k]].prim]); }
            else refs[k] = new_node();
        }
        for (int k = 0; k < m; k++) { put_box(nodes[it.n4], k, N[kids[k]].box); nodes[it.n4].child[k] = refs[k]; }
        for (int k = m - 1; k >= 0; k--) if (refs[k] >= 0) stack.push_back(Item{kids[k], refs[k], it.depth + 1});
    }
    out.root = root4; out.n_nodes = (uint32_t)((int32_t)nodes.size() - node_base); out.n_leaves = n; out.depth = (uint32_t)max_depth;
    for (int k = 0; k < 3; k++) { out.lo[k] = N[0].box.lo[k]; out.hi[k] = N[0].box.hi[k]; out.max_abs = std::max(out.max_abs, std::max(std::fabs(out.lo[k]), std::fabs(out.hi[k]))); }
    return RTS_OK;
}
