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
#include <queue>
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
    if (!ok) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = 3.0e38f; }   // coordinates beyond f32: never hit (see new_node)
}

// ---- reference splitting (early split clipping).  A triangle whose box is mostly empty -- long, thin and diagonal, like
// the fans of slivers around the pole of a lat-long tessellated ellipsoid, hundreds of which overlap each other -- is
// entered into the hierarchy as several references, each the box of the part of the triangle inside one half of the
// parent reference's box (split at the middle of its longest axis).  Every reference leads to the SAME exact test of the
// whole triangle, so results cannot change; a ray near such a fan meets tens of boxes instead of hundreds.
struct Poly { int n; double v[10][3]; };

inline void poly_box(const Poly& p, Box& b) { box_empty(b); for (int i = 0; i < p.n; i++) box_grow_pt(b, p.v[i]); }

// the part of convex polygon p with  sign * (x[axis] - pos) <= 0  (Sutherland-Hodgman against one plane)
inline void poly_clip(const Poly& p, int axis, double pos, double sign, Poly& out)
{
    out.n = 0;
    for (int i = 0; i < p.n; i++) {
        const double* a = p.v[i]; const double* b = p.v[(i + 1) % p.n];
        const double da = sign * (a[axis] - pos), db = sign * (b[axis] - pos);
        if (da <= 0) { if (out.n < 10) { out.v[out.n][0] = a[0]; out.v[out.n][1] = a[1]; out.v[out.n][2] = a[2]; out.n++; } }
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
            const double t = da / (da - db);
            if (out.n < 10) { for (int k = 0; k < 3; k++) out.v[out.n][k] = a[k] + t * (b[k] - a[k]); out.v[out.n][axis] = pos; out.n++; }
        }
    }
}

struct Ref { Poly poly; Box box; uint32_t prim; };

// splits ref r at the middle of its longest axis; returns the surface area saved (<= 0: not worth it)
inline double ref_split(const Ref& r, Ref& lo, Ref& hi)
{
    int ax = 0; double ext = r.box.hi[0] - r.box.lo[0];
    for (int k = 1; k < 3; k++) if (r.box.hi[k] - r.box.lo[k] > ext) { ext = r.box.hi[k] - r.box.lo[k]; ax = k; }
    if (!(ext > 0)) return 0.0;
    const double pos = 0.5 * r.box.lo[ax] + 0.5 * r.box.hi[ax];
    poly_clip(r.poly, ax, pos, 1.0, lo.poly); poly_clip(r.poly, ax, pos, -1.0, hi.poly);
    if (lo.poly.n < 3 || hi.poly.n < 3) return 0.0;
    poly_box(lo.poly, lo.box); poly_box(hi.poly, hi.box);
    for (int k = 0; k < 3; k++) {                                   // never outside the parent's box (clipping rounds)
        lo.box.lo[k] = std::max(lo.box.lo[k], r.box.lo[k]); lo.box.hi[k] = std::min(lo.box.hi[k], r.box.hi[k]);
        hi.box.lo[k] = std::max(hi.box.lo[k], r.box.lo[k]); hi.box.hi[k] = std::min(hi.box.hi[k], r.box.hi[k]);
    }
    lo.box.hi[ax] = pos; hi.box.lo[ax] = pos;                       // both halves contain the cut itself
    lo.prim = hi.prim = r.prim;
    return box_area(r.box) - (box_area(lo.box) + box_area(hi.box));
}

}  // namespace

// verts: [n_verts][3] target-space vertices of the mesh; tris: [n_tris][3] indices into verts.
// Appends the mesh's nodes to `nodes` (child links are indices into that shared array) and its leaf order to
// `leaf_prim` (LOCAL triangle index per leaf slot; a leaf link is ~slot, slot counted over the shared array).
// split_budget: extra references allowed, as a fraction of the triangle count (0: one reference per triangle).
int rts_sah_build(const double* verts, const uint32_t* tris, uint32_t n_tris, double split_budget, std::vector<RtsNode4>& nodes, std::vector<uint32_t>& leaf_prim, RtsBlasInfo& out)
{
    out.root = -1; out.n_nodes = 0; out.n_leaves = 0; out.depth = 0;
    for (int k = 0; k < 3; k++) { out.lo[k] = 0; out.hi[k] = 0; }
    out.max_abs = 0;
    std::vector<Ref> refs; refs.reserve(n_tris + n_tris / 2);
    double area_sum = 0;
    for (uint32_t i = 0; i < n_tris; i++) {
        Ref r; r.poly.n = 3; r.prim = i; bool finite = true;
        for (int k = 0; k < 3; k++) {
            const double* p = verts + 3*(size_t)tris[3*(size_t)i + k];
            finite = finite && std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
            r.poly.v[k][0] = p[0]; r.poly.v[k][1] = p[1]; r.poly.v[k][2] = p[2];
        }
        if (!finite) continue;
        poly_box(r.poly, r.box); area_sum += box_area(r.box);
        refs.push_back(r);
    }
    if (split_budget > 0 && !refs.empty()) {
        const size_t n_orig = refs.size();
        size_t extra = (size_t)(split_budget * (double)n_orig);
        Ref lo, hi;
        struct Cand { double gain; uint32_t ref; bool operator<(const Cand& o) const { return gain < o.gain; } };
        {   // round 0 (half the budget), greedy by box area saved: always split the reference that saves the most; a split
            // must save a quarter of its reference's area and a little of the mean reference area
            const double mean_area = area_sum / (double)n_orig;
            std::priority_queue<Cand> heap;
            auto consider = [&](uint32_t ri) {
                const double g = ref_split(refs[ri], lo, hi);
                if (g > 0.25 * box_area(refs[ri].box) && g > 0.02 * mean_area) heap.push(Cand{g, ri});
            };
            for (uint32_t i = 0; i < (uint32_t)refs.size(); i++) consider(i);
            size_t quota = extra / 2;
            while (quota > 0 && !heap.empty()) {
                const Cand c = heap.top(); heap.pop();
                if (!(ref_split(refs[c.ref], lo, hi) > 0)) continue;
                refs[c.ref] = lo; refs.push_back(hi); quota--; extra--;
                consider(c.ref); consider((uint32_t)refs.size() - 1);
            }
        }
        // rounds 1..3 (the rest), aimed at where boxes pile up: area saved says nothing about HOW MANY references cover
        // the same spot (a ray through the hub of a fan of N slivers tests all N of them, however small they are).  Build
        // a provisional tree, count for every reference how many reference boxes contain its centre, and split the
        // references in crowded spots, most crowded and most wasteful first; recount and repeat.
        for (int round = 0; round < 3 && extra > 0; round++) {
            const uint32_t nr = (uint32_t)refs.size();
            std::vector<Box> tb(nr); std::vector<double> tc(3 * (size_t)nr); std::vector<uint32_t> tidx(nr);
            for (uint32_t i = 0; i < nr; i++) { tb[i] = refs[i].box; tidx[i] = i; for (int k = 0; k < 3; k++) tc[3*(size_t)i + k] = 0.5 * refs[i].box.lo[k] + 0.5 * refs[i].box.hi[k]; }
            Builder T{tb, tc, tidx, {}};
            T.nodes.reserve(2 * (size_t)nr);
            T.build(0, nr);
            std::priority_queue<Cand> heap;
            std::vector<int> stack;
            for (uint32_t i = 0; i < nr; i++) {
                const double* c = &tc[3*(size_t)i];
                int count = 0; stack.clear(); stack.push_back(0);
                while (!stack.empty() && count < 4096) {
                    const Node2& nd = T.nodes[stack.back()]; stack.pop_back();
                    if (c[0] < nd.box.lo[0] || c[0] > nd.box.hi[0] || c[1] < nd.box.lo[1] || c[1] > nd.box.hi[1] || c[2] < nd.box.lo[2] || c[2] > nd.box.hi[2]) continue;
                    if (nd.left < 0) count++; else { stack.push_back(nd.left); stack.push_back(nd.right); }
                }
                if (count < 12) continue;                             // a dozen overlapping neighbours is normal on a curved surface
                const double g = ref_split(refs[i], lo, hi);
                if (g > 0.10 * box_area(refs[i].box)) heap.push(Cand{(double)count * g / box_area(refs[i].box), i});
            }
            if (heap.empty()) break;
            size_t quota = std::max<size_t>(extra / (size_t)(3 - round), 1);
            while (quota > 0 && extra > 0 && !heap.empty()) {
                const Cand c = heap.top(); heap.pop();
                if (!(ref_split(refs[c.ref], lo, hi) > 0)) continue;
                refs[c.ref] = lo; refs.push_back(hi); quota--; extra--;
            }
        }
    }
    std::vector<Box> pb(refs.size()); std::vector<double> cen(3 * refs.size()); std::vector<uint32_t> prim_of(refs.size());
    for (size_t i = 0; i < refs.size(); i++) {
        pb[i] = refs[i].box; prim_of[i] = refs[i].prim;
        for (int k = 0; k < 3; k++) cen[3*i + k] = 0.5 * refs[i].box.lo[k] + 0.5 * refs[i].box.hi[k];
    }
    { std::vector<Ref>().swap(refs); }
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
    // unused slots: the degenerate box lo = hi = 3e38 in every axis.  The traversal takes min / max of the two plane
    // parameters of an axis (rts_trace.hip), so an INVERTED box would read as a valid one; a point at 3e38 gives entry = exit
    // = +-inf (or 3e38 |1/d|), which fails entry <= min(exit, t_prune) for every ray.
    auto new_node = [&]() { RtsNode4 o; for (int k = 0; k < 4; k++) { o.lox[k] = o.loy[k] = o.loz[k] = 3.0e38f; o.hix[k] = o.hiy[k] = o.hiz[k] = 3.0e38f; o.child[k] = 0x7fffffff; o.pad[k] = 0; }
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

// The host builder on its own, without a device (tests: the invariants of the hierarchy, and the builder under the sanitizers):
// nodes as rts_get_bvh returns them (128-byte records, leaf children as ~slot), leaf_prim[slot] = triangle index.  Two calls:
// null outputs give the sizes.
extern "C" int rts_build_hierarchy_host(const double* vertices, const uint32_t* triangles, uint32_t n_triangles, double split_budget, void* nodes128, uint32_t node_capacity,
                                        uint32_t* leaf_prim, uint32_t leaf_capacity, uint32_t* n_nodes, uint32_t* n_leaves, int32_t* root)
{
    if ((n_triangles && (!vertices || !triangles)) || !n_nodes || !n_leaves) { rts_set_error("rts_build_hierarchy_host: null argument"); return RTS_ERR_INVALID; }
    std::vector<RtsNode4> nodes; std::vector<uint32_t> lp; RtsBlasInfo info; memset(&info, 0, sizeof(info));
    const int rc = rts_sah_build(vertices, triangles, n_triangles, split_budget, nodes, lp, info);
    if (rc != RTS_OK) return rc;
    *n_nodes = (uint32_t)nodes.size(); *n_leaves = (uint32_t)lp.size();
    if (root) *root = info.root;
    if (!nodes128 && !leaf_prim) return RTS_OK;
    if (!nodes128 || !leaf_prim || node_capacity < nodes.size() || leaf_capacity < lp.size()) { rts_set_error("rts_build_hierarchy_host: capacity too small (%zu nodes, %zu leaf slots)", nodes.size(), lp.size()); return RTS_ERR_CAPACITY; }
    if (!nodes.empty()) memcpy(nodes128, nodes.data(), sizeof(RtsNode4) * nodes.size());
    if (!lp.empty()) memcpy(leaf_prim, lp.data(), sizeof(uint32_t) * lp.size());
    return RTS_OK;
}
