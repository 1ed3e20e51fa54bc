// =====================================================================================
// rts_oracle.cpp -- CPU ORACLE for the RTS hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is a plain C++ restatement of the arithmetic of the reference
// (ymartin101/RTS) for the path  ray launch -> closest triangle hit -> reflect/refract
// shading -> receiver capture -> host finalise -> per-receiver path aggregation.
// It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// CHECK the HIP product path.  Nothing under rts_amd/ may include, link or call it.
//
// PARITY STATUS: *parity unpinned*.  The reference ships no tests, golden vectors or
// fixtures and cannot be compiled here (needs NVIDIA OptiX <= 6.x, nvcc and the SOARS
// rs*.cuh headers).  The oracle is pinned only by (1) the closed-form known-answer tests
// derivable from the cited formulas (tests/test_oracle_kat.py) and (2) the source text.
//
// Third-party arithmetic that is NOT under /root/reference (NVIDIA OptiX SDK, legacy rt*
// API, version unpinned by the reference) is restated from its published semantics:
//   reflect(i, n)  = i - 2*n*dot(n, i)                        (f32)
//   refract(r,i,n,ior): c = dot(i,n); if c > 0 { eta = ior; n = -n; c = -c } else eta = 1/ior;
//                       k = 1 - eta^2 (1 - c^2); k < 0 -> r = 0, false;
//                       else r = normalize(eta*i - (eta*c + sqrtf(k))*n), true   (f32)
//   RT_DEFAULT_MAX = 1e27f;  rtPotentialIntersection(t): tmin < t < current tmax  (f32)
// The OptiX "Bvh" builder/traverser is closed source; the oracle's definition of the
// closest hit is the BVH-free brute force over all primitives, ties in f32 t resolved to
// the lowest (target index, primitive index).
//
// Every function cites the reference file:line it follows.  Compile with
//   g++ -O2 -ffp-contract=off  (no FMA contraction: expression trees are literal).
//
// Deliberate deviations from the literal text, each flagged where it occurs:
//   [D1] atan2f in the capture test (ray_tracer.cu:326-329) is CUDA libm there (<= 2 ulp,
//        bits unknowable).  The oracle uses orc_atan2f_cr(): atan2 evaluated in f64 from
//        basic IEEE operations only and rounded once to f32, so that a HIP kernel built
//        from the same operations reproduces it bit-for-bit.  tests check it against
//        glibc atan2f to <= 1 ulp.
//   [D2] pow(x, 2) (aggregation.cu:89) is restated as x*x.
//   [D3] when only a subset of the W^3 launch indices is traced (ray_first/ray_stride),
//        output rows are indexed by the local sample number and the refraction row stride
//        W^3 (normal_shader.cu:214) becomes the local ray count.  With the full launch the
//        layout is exactly the reference's.
//   [D4] sin / cos (ray_generation's launch-constant trigonometry, the mesh and receiver builders) are CUDA device
//        functions there (bits unknowable); here they are glibc's, called ONE AT A TIME (orc_sin ...): a compiler that
//        merges sin(x), cos(x) into sincos(x) changes last bits, and the product's host code calls them separately.
// =====================================================================================
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <set>
#include <string>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <vector>

// ------------------------------------------------------------------ ray_tracer.h:9-28
#define SCENE_EPS 0.005f
#define SCENE_EPS_R 0.005f
#define RT_DEFAULT_MAX 1e27f

struct d2 { double x, y; } __attribute__((aligned(16)));
struct d3 { double x, y, z; };
struct f3 { float x, y, z; };

struct PerRayData {              // ray_tracer.h:13-28 ; sizeof == 144, alignof == 16
    double rayLength;
    d2 refrIndex;
    unsigned int reflDepth;
    unsigned int refrDepth;
    unsigned int maxRayIndex;
    d3 rayDirection;
    d3 firstHitPoint;
    d3 prevHitPoint;
    double power;
    double doppler;
    int received;
    bool end;
};
static_assert(sizeof(PerRayData) == 144, "PerRayData layout");
static_assert(alignof(PerRayData) == 16, "PerRayData alignment");

struct Ray { f3 origin; f3 direction; float tmin; float tmax; };

// ------------------------------------------------------------------ small vector helpers
// ray_tracer.cu:72-122, normal_shader.cu:48-115, triangle_mesh.cu:39-94
// libm's sin and cos, ONE call per function: GCC merges sin(x) and cos(x) of the same argument into a single sincos(x)
// (also through libstdc++'s __builtin_sinf / __builtin_cosf, which -fno-builtin-* does not reach), and glibc's sincos does
// not promise the bits of its separate sin and cos.  The product's host code (clang) calls them separately; so does this.
static __attribute__((noinline)) double orc_sin(double x) { return sin(x); }
static __attribute__((noinline)) double orc_cos(double x) { return cos(x); }
static __attribute__((noinline)) float orc_sinf(float x) { return sinf(x); }
static __attribute__((noinline)) float orc_cosf(float x) { return cosf(x); }
static inline d3 to_double3(double x, double y, double z) { d3 o; o.x = x; o.y = y; o.z = z; return o; }
static inline d3 operator+(d3 a, d3 b) { return to_double3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline d3 operator-(d3 a, d3 b) { return to_double3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline d3 mul(double a, d3 b) { return to_double3(a * b.x, a * b.y, a * b.z); }      // triangle_mesh.cu:66
static inline d3 crossd3(d3 a, d3 b) { return to_double3(a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }
static inline double dotd3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double magsquared3(d3 a) { return (a.x*a.x + a.y*a.y + a.z*a.z); }
static inline double lengthd3(d3 in) { return std::sqrt(in.x*in.x + in.y*in.y + in.z*in.z); }
static inline d3 normalised3(d3 in) { double norm = lengthd3(in); return to_double3(in.x/norm, in.y/norm, in.z/norm); }
static inline f3 make_float3(float x, float y, float z) { f3 o; o.x = x; o.y = y; o.z = z; return o; }
static inline f3 normalise_float3(double in1, double in2, double in3) {   // ray_tracer.cu:125-129
    double norm = lengthd3(to_double3(in1, in2, in3));
    return make_float3((float)(in1/norm), (float)(in2/norm), (float)(in3/norm));
}
static inline d3 float3_to_double3(f3 in) { return to_double3(in.x, in.y, in.z); }

// ------------------------------------------------------------------ [D1] atan2 from basic ops
// atan(x) for x in [0, 1]: argument reduction x -> (x - c)/(1 + x c) about c = k/8 grid
// points with tabulated atan(c), then an odd Taylor series; accurate to ~1e-16, which is
// all the f32 rounding needs.  Only + - * / and comparisons are used.
static const double ORC_ATAN_TAB[9] = {
    0.0,
    0.12435499454676143503,   // atan(1/8)
    0.24497866312686415417,   // atan(2/8)
    0.35877067027057222040,   // atan(3/8)
    0.46364760900080611621,   // atan(4/8)
    0.55859931534356243597,   // atan(5/8)
    0.64350110879328438680,   // atan(6/8)
    0.71882999962162450542,   // atan(7/8)
    0.78539816339744830962    // atan(1)
};
static double orc_atan_unit(double x) {           // 0 <= x <= 1
    int k = (int)(x * 8.0 + 0.5);
    double c = (double)k * 0.125;
    double r = (x - c) / (1.0 + x * c);          // |r| <= ~1/16
    double r2 = r * r;
    // odd series r - r^3/3 + r^5/5 - ... to r^17 (|r|^19/19 < 1e-24)
    double s = 1.0/17.0;
    s = 1.0/15.0 - r2 * s;
    s = 1.0/13.0 - r2 * s;
    s = 1.0/11.0 - r2 * s;
    s = 1.0/9.0 - r2 * s;
    s = 1.0/7.0 - r2 * s;
    s = 1.0/5.0 - r2 * s;
    s = 1.0/3.0 - r2 * s;
    s = 1.0 - r2 * s;
    return ORC_ATAN_TAB[k] + r * s;
}
static double orc_atan2_f64(double y, double x) {
    const double PI = 3.14159265358979323846, PI_2 = 1.57079632679489661923;
    if (x != x || y != y) return x + y;
    double ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax) a = orc_atan_unit(ay / ax);
    else a = PI_2 - orc_atan_unit(ax / ay);
    bool xneg = std::signbit(x);
    if (xneg) a = PI - a;
    return std::signbit(y) ? -a : a;
}
static float orc_atan2f_cr(float y, float x) { return (float)orc_atan2_f64((double)y, (double)x); }

// ------------------------------------------------------------------ scene containers
struct OMesh {
    std::vector<uint32_t> tris;     // dbuf_triangles (uint3)
    std::vector<d3> verts;          // dbuf_triVertices (world space, after += position)
    std::vector<d3> normals;        // dbuf_normals
    double reflCoeff;               // d_targReflCoeff
    double refrIndex;               // d_targRefrIndex
    d3 vel;                         // dbuf_targ_vel[targ]
};
struct OBvhNode { float lo[3], hi[3]; int left, right, first, count; };
struct OScene {
    std::vector<OMesh> meshes;
    std::vector<d3> sphCentre; std::vector<double> sphRadius, minTheta, maxTheta, minPhi, maxPhi;
    // test-only acceleration structure (median split, own code, not the product's LBVH)
    std::vector<OBvhNode> nodes; std::vector<uint32_t> primTarg, primIdx;
    bool bvhBuilt = false;
};

struct OPulse {                      // launch constants, ray_tracer.cu:39-45, normal_shader.cu:35-42
    double rayOrigin[3];
    double txSpan[3];
    double txDir[2];
    uint32_t width;                  // d_width
    uint32_t maxRefl;                // h_maxReflDepth (user value; d_maxReflDepth = maxRefl + 1)
    uint32_t maxRefr;                // h_maxRefrDepth after the clamp to 2 (ray_tracer.cpp:604-605)
    uint32_t interpolate_smooth;     // d_interpolate_smooth
};

struct OTraceCtx {
    const OScene* sc; const OPulse* p;
    unsigned d_maxReflDepth, d_maxRefrDepth, depthTotal;
    uint64_t stride;                 // W^3 or local ray count [D3]
    bool useBvh;
    PerRayData* results; int* targ_intersect; d2* rcs_angle;
    // debug trace (oracle-only outputs): per launch ray, per segment of the REFLECTION chain
    int* hit_prim; float* hit_t; unsigned hitCols;
    uint64_t nodeVisits, triTests, segments, shaded;
    unsigned long long cov[24] = {0};   // branch coverage of the miss program, private to the worker (COV_* below)
};

// ------------------------------------------------------------------ triangle_mesh.cu:121-137
static inline bool intersect_triangle_doubles(const Ray& ray, const PerRayData& prd, const d3& p0, const d3& p1,
                                              const d3& p2, d3& n, double& t, double& beta, double& gamma)
{
    const d3 e0 = p1 - p0;
    const d3 e1 = p0 - p2;
    n = crossd3(e1, e0);
    const d3 e2 = mul((1/dotd3(n, prd.rayDirection)), (p0 - prd.prevHitPoint));
    const d3 i = crossd3(prd.rayDirection, e2);
    beta = dotd3(i, e1);
    gamma = dotd3(i, e0);
    t = dotd3(n, e2);
    return ( (t < ray.tmax) & (t > ray.tmin) & (beta >= 0.0f) & (gamma >= 0.0f) & (beta + gamma <= 1) );
}

// triangle_mesh.cu:169-194 -- attribute normal for an accepted hit
static inline d3 shading_normal(const OMesh& m, unsigned prim_index, bool smooth, const d3& n, double beta, double gamma)
{
    unsigned v0 = m.tris[3*prim_index], v1 = m.tris[3*prim_index+1], v2 = m.tris[3*prim_index+2];
    d3 normal;
    if (smooth) {
        if (m.normals.size() > m.verts.size()) {            // "rect": per-face normals  :178-180
            normal = m.normals[prim_index];
        } else {                                            // :182-184
            d3 n0 = m.normals[v0], n1 = m.normals[v1], n2 = m.normals[v2];
            normal = to_double3(n1.x*beta + n2.x*gamma + n0.x*(1.0f - beta - gamma),
                                n1.y*beta + n2.y*gamma + n0.y*(1.0f - beta - gamma),
                                n1.z*beta + n2.z*gamma + n0.z*(1.0f - beta - gamma));
        }
        normal = normalised3(normal);                       // :188
    } else {
        normal = normalised3(n);                            // :193
    }
    return normal;
}

// triangle_mesh.cu:204-233 -- per-primitive AABB, f64 -> f32 rounded outward
static inline float d2f_rd(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; }
static inline float d2f_ru(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; }
static bool prim_bound(const d3& v0, const d3& v1, const d3& v2, float out[6])
{
    const double area = lengthd3(crossd3(v1 - v0, v2 - v0));
    if (area > 0.0f && !std::isinf(area)) {
        d3 mn = to_double3(std::min(std::min(v0.x, v1.x), v2.x), std::min(std::min(v0.y, v1.y), v2.y), std::min(std::min(v0.z, v1.z), v2.z));
        d3 mx = to_double3(std::max(std::max(v0.x, v1.x), v2.x), std::max(std::max(v0.y, v1.y), v2.y), std::max(std::max(v0.z, v1.z), v2.z));
        out[0] = d2f_rd(mn.x); out[1] = d2f_rd(mn.y); out[2] = d2f_rd(mn.z);
        out[3] = d2f_ru(mx.x); out[4] = d2f_ru(mx.y); out[5] = d2f_ru(mx.z);
        return true;
    }
    // optix::Aabb::invalidate(): min = +1e37f, max = -1e37f
    out[0] = out[1] = out[2] = 1e37f; out[3] = out[4] = out[5] = -1e37f;
    return false;
}

// ------------------------------------------------------------------ oracle-only BVH (median split)
// Used for the cpu_baseline leg and for large sampled parity runs, validated against the
// brute force in tests.  Boxes are padded so that the f64 slab test below is conservative
// with respect to intersect_triangle_doubles().
static void bvh_build(OScene& sc)
{
    sc.nodes.clear(); sc.primTarg.clear(); sc.primIdx.clear();
    std::vector<std::array<float,6>> boxes; std::vector<std::array<double,3>> cent;
    for (size_t ti = 0; ti < sc.meshes.size(); ti++) {
        const OMesh& m = sc.meshes[ti];
        for (size_t pi = 0; pi < m.tris.size()/3; pi++) {
            const d3& a = m.verts[m.tris[3*pi]]; const d3& b = m.verts[m.tris[3*pi+1]]; const d3& c = m.verts[m.tris[3*pi+2]];
            double lo[3] = { std::min(std::min(a.x,b.x),c.x), std::min(std::min(a.y,b.y),c.y), std::min(std::min(a.z,b.z),c.z) };
            double hi[3] = { std::max(std::max(a.x,b.x),c.x), std::max(std::max(a.y,b.y),c.y), std::max(std::max(a.z,b.z),c.z) };
            bool finite = true; double s = 0;
            for (int k = 0; k < 3; k++) { finite = finite && std::isfinite(lo[k]) && std::isfinite(hi[k]); s = std::max(s, std::max(std::fabs(lo[k]), std::fabs(hi[k]))); }
            if (!finite) continue;
            double pad = s * 2.4e-7 + 1e-30;
            std::array<float,6> bx;
            for (int k = 0; k < 3; k++) { bx[k] = d2f_rd(lo[k] - pad); bx[3+k] = d2f_ru(hi[k] + pad); }
            boxes.push_back(bx); cent.push_back({ (lo[0]+hi[0])*0.5, (lo[1]+hi[1])*0.5, (lo[2]+hi[2])*0.5 });
            sc.primTarg.push_back((uint32_t)ti); sc.primIdx.push_back((uint32_t)pi);
        }
    }
    size_t n = boxes.size();
    std::vector<uint32_t> order(n); for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    struct Job { int node; size_t lo, hi; };
    std::vector<Job> jobs;
    sc.nodes.push_back(OBvhNode{}); jobs.push_back({0, 0, n});
    while (!jobs.empty()) {
        Job j = jobs.back(); jobs.pop_back();
        OBvhNode nd; for (int k = 0; k < 3; k++) { nd.lo[k] = INFINITY; nd.hi[k] = -INFINITY; }
        double clo[3] = {INFINITY,INFINITY,INFINITY}, chi[3] = {-INFINITY,-INFINITY,-INFINITY};
        for (size_t i = j.lo; i < j.hi; i++) {
            const auto& b = boxes[order[i]];
            for (int k = 0; k < 3; k++) { nd.lo[k] = std::min(nd.lo[k], b[k]); nd.hi[k] = std::max(nd.hi[k], b[3+k]);
                clo[k] = std::min(clo[k], cent[order[i]][k]); chi[k] = std::max(chi[k], cent[order[i]][k]); }
        }
        nd.left = nd.right = -1; nd.first = (int)j.lo; nd.count = (int)(j.hi - j.lo);
        if (j.hi - j.lo > 4) {
            int ax = 0; if (chi[1]-clo[1] > chi[ax]-clo[ax]) ax = 1; if (chi[2]-clo[2] > chi[ax]-clo[ax]) ax = 2;
            size_t mid = (j.lo + j.hi)/2;
            std::nth_element(order.begin()+j.lo, order.begin()+mid, order.begin()+j.hi,
                             [&](uint32_t a, uint32_t b){ return cent[a][ax] < cent[b][ax]; });
            nd.left = (int)sc.nodes.size(); sc.nodes.push_back(OBvhNode{});
            nd.right = (int)sc.nodes.size(); sc.nodes.push_back(OBvhNode{});
            nd.count = 0;
            jobs.push_back({nd.left, j.lo, mid}); jobs.push_back({nd.right, mid, j.hi});
        }
        sc.nodes[j.node] = nd;
    }
    std::vector<uint32_t> pt(n), pi(n);
    for (size_t i = 0; i < n; i++) { pt[i] = sc.primTarg[order[i]]; pi[i] = sc.primIdx[order[i]]; }
    sc.primTarg.swap(pt); sc.primIdx.swap(pi);
    sc.bvhBuilt = true;
}

struct Hit { bool found; float t; unsigned targ, prim; d3 normal; };

static inline void test_prim(OTraceCtx& cx, const Ray& ray, const PerRayData& prd, unsigned ti, unsigned pi, float& cur_tmax, Hit& h)
{
    const OMesh& m = cx.sc->meshes[ti];
    const d3& p0 = m.verts[m.tris[3*pi]]; const d3& p1 = m.verts[m.tris[3*pi+1]]; const d3& p2 = m.verts[m.tris[3*pi+2]];
    d3 n; double t, beta, gamma;
    cx.triTests++;
    if (intersect_triangle_doubles(ray, prd, p0, p1, p2, n, t, beta, gamma)) {      // triangle_mesh.cu:166
        float tf = (float)t;                                                         // rtPotentialIntersection(t) takes float  :167
        bool better = (tf > ray.tmin) && (tf < cur_tmax);
        bool tie = h.found && (tf == cur_tmax) && (ti < h.targ || (ti == h.targ && pi < h.prim));
        if (better || tie) {
            cur_tmax = tf; h.found = true; h.t = tf; h.targ = ti; h.prim = pi;
            h.normal = shading_normal(m, pi, cx.p->interpolate_smooth != 0, n, beta, gamma);
        }
    }
}

// closest hit of rtTrace: brute force (definition) or oracle BVH (validated equal)
static Hit find_closest(OTraceCtx& cx, const Ray& ray, const PerRayData& prd)
{
    Hit h; h.found = false; h.t = 0; h.targ = h.prim = 0; h.normal = to_double3(0,0,0);
    float cur_tmax = ray.tmax;
    cx.segments++;
    if (!cx.useBvh) {
        for (unsigned ti = 0; ti < cx.sc->meshes.size(); ti++) {
            unsigned np = (unsigned)(cx.sc->meshes[ti].tris.size()/3);
            for (unsigned pi = 0; pi < np; pi++) test_prim(cx, ray, prd, ti, pi, cur_tmax, h);
        }
        return h;
    }
    const OScene& sc = *cx.sc;
    if (sc.nodes.empty() || sc.primIdx.empty()) return h;
    const d3 o = prd.prevHitPoint, d = prd.rayDirection;
    const double inv[3] = { 1.0/d.x, 1.0/d.y, 1.0/d.z };
    const double oo[3] = { o.x, o.y, o.z };
    int stack[128]; int sp = 0; stack[sp++] = 0;
    while (sp > 0) {
        const OBvhNode& nd = sc.nodes[stack[--sp]];
        cx.nodeVisits++;
        double tn = 0.0, tf = (double)std::nextafterf(cur_tmax, INFINITY);
        bool ok = true;
        for (int k = 0; k < 3; k++) {
            double t1 = ((double)nd.lo[k] - oo[k]) * inv[k], t2 = ((double)nd.hi[k] - oo[k]) * inv[k];
            if (t1 != t1 || t2 != t2) {           // 0 * inf: origin on the slab plane with zero direction component
                if (oo[k] < (double)nd.lo[k] || oo[k] > (double)nd.hi[k]) ok = false;
                continue;
            }
            double a = std::min(t1, t2), b = std::max(t1, t2);
            tn = std::max(tn, a); tf = std::min(tf, b);
        }
        if (!ok || tn > tf * (1.0 + 1e-12) + 1e-300) continue;
        if (nd.left < 0) {
            for (int i = 0; i < nd.count; i++) test_prim(cx, ray, prd, sc.primTarg[nd.first+i], sc.primIdx[nd.first+i], cur_tmax, h);
        } else {
            if (sp + 2 > 128) { fprintf(stderr, "oracle bvh stack overflow\n"); abort(); }
            stack[sp++] = nd.left; stack[sp++] = nd.right;
        }
    }
    return h;
}

// ------------------------------------------------------------------ ray_tracer.cu:53-69
static void normalise_angle(double& angle) { while (angle < -M_PI) angle += 2*M_PI; while (angle > M_PI) angle -= 2*M_PI; }
static bool angle_in_range(double testAngle, double a, double b)
{
    a -= testAngle; b -= testAngle;
    normalise_angle(a); normalise_angle(b);
    if (a * b >= 0) return false;
    return std::fabs(a - b) < M_PI;
}
static inline d3 sph_to_cart(double azi, double ele) {      // ray_tracer.cu:132-139
    d3 cart; cart.x = orc_cos(azi)*orc_cos(ele); cart.y = orc_sin(azi)*orc_cos(ele); cart.z = orc_sin(ele); return cart;
}
static inline d2 cart_to_sph(d3 in) {                       // normal_shader.cu:118-124
    d2 sph; sph.x = std::atan2(in.y, in.x); sph.y = std::atan2(in.z, std::sqrt(in.x*in.x + in.y*in.y)); return sph;
}

// Branch coverage of the miss program (observers only: they never feed back into the arithmetic).  Tests use them to
// prove that a scene really drives the oracle through the branch it claims to cover (orc_coverage).
enum { COV_PHI_LOW = 0,       // :332-335  phi < -pi/2 correction taken
       COV_PHI_HIGH,          // :337-340  phi > +pi/2 correction taken
       COV_WIN_MINPHI,        // :354-360  window crosses -pi/2 (second region built from minPhi)
       COV_WIN_MAXPHI,        // :362-368  window crosses +pi/2 (second region built from maxPhi)
       COV_CAPTURE_REGION1,   // :374      captured by the first (theta, phi) region
       COV_CAPTURE_REGION2,   // :375      captured ONLY by the second region
       COV_BOTH_ROOTS,        // :378-381  both roots captured, nearest chosen
       COV_SECOND_ROOT_ONLY,  //           only root 1 captured (root 0 invalid or outside the window)
       COV_RECAPTURE,         // :272,393-426 quirk 4: a ray already received is captured again by a later receiver
       COV_RECAPTURE_REFLECTED, //         ... and it is a reflected ray, so power is multiplied a second time (:419-425)
       COV_DIRECT_CAPTURE,    // :410-417
       COV_EARTH_TESTED,      // :438      Earth block entered (prd.end == false)
       COV_EARTH_ROOT0,       // :462-471  t0 >= 0 && rayLength > 0
       COV_EARTH_ROOT1,       //           t1 >= 0 && rayLength > 0
       COV_EARTH_BOTH,        //           both roots added to rayLength (quirk 6)
       COV_EARTH_MISS,        //           discriminant <= 0 or no valid root
       COV_N };
static_assert(COV_N <= 24, "OTraceCtx::cov is too small");
static std::atomic<unsigned long long> g_cov[COV_N];            // totals since the last reset; the workers count privately
#define COV(k) (cx.cov[k]++)                                      // (a shared atomic per ray made 256 threads crawl) and add up once

// ------------------------------------------------------------------ ray_tracer.cu:260-478  miss program
static void miss_program(OTraceCtx& cx, PerRayData& prd)
{
    const OScene& sc = *cx.sc;
    const d3 d_rayOrigin = to_double3(cx.p->rayOrigin[0], cx.p->rayOrigin[1], cx.p->rayOrigin[2]);
    if (prd.end == false) {
        double A, B, C, discriminant;
        double t[2] = {0, 0};
        for (unsigned int Rx_i = 0; Rx_i < sc.sphCentre.size(); Rx_i++) {
            const d3 c = sc.sphCentre[Rx_i];
            A = ((prd.rayDirection).x)*((prd.rayDirection).x) + ((prd.rayDirection).y)*((prd.rayDirection).y) + ((prd.rayDirection).z)*((prd.rayDirection).z);
            B = 2*((((prd.prevHitPoint).x - c.x)*(prd.rayDirection).x) +
                   (((prd.prevHitPoint).y - c.y)*(prd.rayDirection).y) +
                   (((prd.prevHitPoint).z - c.z)*(prd.rayDirection).z));
            C = (prd.prevHitPoint).x*(prd.prevHitPoint).x + (prd.prevHitPoint).y*(prd.prevHitPoint).y + (prd.prevHitPoint).z*(prd.prevHitPoint).z +
                (c.x*c.x) +
                (c.y*c.y) +
                (c.z*c.z) -
                2*((c.x*(prd.prevHitPoint).x) + (c.y*(prd.prevHitPoint).y) + (c.z*(prd.prevHitPoint).z)) -
                sc.sphRadius[Rx_i]*sc.sphRadius[Rx_i];
            discriminant = B*B - 4*A*C;
            if (discriminant > 0.f) {
                discriminant = std::sqrt(discriminant);
                t[0] = (-B - discriminant)/(2*A);
                t[1] = (-B + discriminant)/(2*A);
                unsigned int received_root = 2;
                for (int i = 0; i < 2; i++) {
                    if ((t[i] >= 0) && ((prd.rayLength + t[i]) > SCENE_EPS) && ((prd.rayLength + t[i]) > SCENE_EPS_R)) {
                        d3 endPoint;
                        endPoint.x = (prd.prevHitPoint).x + t[i]*(prd.rayDirection).x;
                        endPoint.y = (prd.prevHitPoint).y + t[i]*(prd.rayDirection).y;
                        endPoint.z = (prd.prevHitPoint).z + t[i]*(prd.rayDirection).z;
                        // [D1] atan2f: arguments are narrowed to float first (float atan2f(float,float))
                        double theta = orc_atan2f_cr((float)(endPoint.y - c.y), (float)(endPoint.x - c.x));
                        double phi = orc_atan2f_cr((float)(endPoint.z - c.z), (float)std::sqrt(((endPoint.y - c.y) *
                                            (endPoint.y - c.y)) + ((endPoint.x - c.x) *
                                            (endPoint.x - c.x))));
                        if ((phi < -M_PI/2)) { theta += M_PI; phi = -M_PI - phi; COV(COV_PHI_LOW); }
                        if ((phi > M_PI/2)) { theta += M_PI; phi = M_PI - phi; COV(COV_PHI_HIGH); }
                        double d_maxTheta1 = sc.maxTheta[Rx_i];
                        double d_minTheta1 = sc.minTheta[Rx_i];
                        double d_maxTheta2 = d_maxTheta1;
                        double d_minTheta2 = d_minTheta1;
                        double d_maxPhi1 = sc.maxPhi[Rx_i];
                        double d_minPhi1 = sc.minPhi[Rx_i];
                        double d_maxPhi2 = d_maxPhi1;
                        double d_minPhi2 = d_minPhi1;
                        if ((d_minPhi1 < -M_PI/2)) {
                            d_maxTheta2 += M_PI; d_minTheta2 += M_PI;
                            d_maxPhi2 = -M_PI - d_minPhi1; d_minPhi2 = -M_PI/2; d_minPhi1 = -M_PI/2; COV(COV_WIN_MINPHI);
                        }
                        if ((d_maxPhi1 > M_PI/2)) {
                            d_maxTheta2 += M_PI; d_minTheta2 += M_PI;
                            d_minPhi2 = M_PI - d_maxPhi1; d_maxPhi2 = M_PI/2; d_maxPhi1 = M_PI/2; COV(COV_WIN_MAXPHI);
                        }
                        if (((angle_in_range(theta, d_minTheta1, d_maxTheta1)) && (angle_in_range(phi, d_minPhi1, d_maxPhi1))) ||
                            ((angle_in_range(theta, d_minTheta2, d_maxTheta2)) && (angle_in_range(phi, d_minPhi2, d_maxPhi2))))
                        {
                            if ((angle_in_range(theta, d_minTheta1, d_maxTheta1)) && (angle_in_range(phi, d_minPhi1, d_maxPhi1))) COV(COV_CAPTURE_REGION1); else COV(COV_CAPTURE_REGION2);
                            if (received_root == 2) { received_root = i; if (i == 1) COV(COV_SECOND_ROOT_ONLY); }
                            else { COV(COV_BOTH_ROOTS); if (t[received_root] > t[i]) received_root = i; }
                        }
                    }
                }
                if (received_root < 2) {
                    if (prd.received >= 0) { COV(COV_RECAPTURE); if (!((prd.reflDepth == 0) && (prd.refrDepth == 0))) COV(COV_RECAPTURE_REFLECTED); }
                    prd.end = true;
                    unsigned int i = received_root;
                    d3 endPoint;
                    endPoint.x = (prd.prevHitPoint).x + t[i]*(prd.rayDirection).x;
                    endPoint.y = (prd.prevHitPoint).y + t[i]*(prd.rayDirection).y;
                    endPoint.z = (prd.prevHitPoint).z + t[i]*(prd.rayDirection).z;
                    d3 RxRange;
                    if ((prd.reflDepth == 0) && (prd.refrDepth == 0)) {
                        RxRange = endPoint - d_rayOrigin;
                        if (lengthd3(RxRange) >= SCENE_EPS) {
                            prd.power = 1/(4*M_PI*4*M_PI*(magsquared3(RxRange)));
                            COV(COV_DIRECT_CAPTURE);
                            prd.doppler = 0;
                            prd.rayLength += t[i];
                            prd.received = Rx_i;
                        }
                    } else {
                        RxRange = endPoint - prd.prevHitPoint;
                        if (lengthd3(RxRange) >= SCENE_EPS_R) {
                            prd.power *= 1/((magsquared3(RxRange))*4*M_PI*4*M_PI);
                            prd.rayLength += t[i];
                            prd.received = Rx_i;
                        }
                    }
                }
            }
        }
    }
    if (prd.end == false) {                                   // Earth sphere  :438-476
        double d_earthRadius = 6378136;
        double A = ((prd.rayDirection).x)*((prd.rayDirection).x) + ((prd.rayDirection).y)*((prd.rayDirection).y) + ((prd.rayDirection).z)*((prd.rayDirection).z);
        double B = 2*((prd.prevHitPoint).x*(prd.rayDirection).x + (prd.prevHitPoint).y*(prd.rayDirection).y + (prd.prevHitPoint).z*(prd.rayDirection).z);
        double C = (prd.prevHitPoint).x*(prd.prevHitPoint).x + (prd.prevHitPoint).y*(prd.prevHitPoint).y + (prd.prevHitPoint).z*(prd.prevHitPoint).z - d_earthRadius*d_earthRadius;
        double discriminant = B*B - 4*A*C;
        double t[2] = {0, 0};
        COV(COV_EARTH_TESTED);
        int n_roots = 0;
        if (discriminant > 0.f) {
            discriminant = std::sqrt(discriminant);
            t[0] = (-B - discriminant)/(2*A);
            t[1] = (-B + discriminant)/(2*A);
            for (int i = 0; i < 2; i++) {
                if ((t[i] >= 0) && (prd.rayLength > 0)) { prd.end = true; prd.rayLength += t[i]; n_roots++; COV(i == 0 ? COV_EARTH_ROOT0 : COV_EARTH_ROOT1); }
            }
        }
        if (n_roots == 2) COV(COV_EARTH_BOTH);
        if (n_roots == 0) COV(COV_EARTH_MISS);
    }
}

// OptiX SDK optixu_math: reflect / refract (published semantics, see header)
static inline float dotf3(f3 a, f3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }
static inline f3 reflect_f3(f3 i, f3 n) {
    float d = dotf3(n, i);
    f3 tn = make_float3(2.0f*n.x, 2.0f*n.y, 2.0f*n.z);
    return make_float3(i.x - tn.x*d, i.y - tn.y*d, i.z - tn.z*d);
}
static inline bool refract_f3(f3& r, f3 i, f3 n, float ior) {
    f3 nn = n;
    float negNdotV = dotf3(i, nn);
    float eta;
    if (negNdotV > 0.0f) { eta = ior; nn = make_float3(-n.x, -n.y, -n.z); negNdotV = -negNdotV; }
    else { eta = 1.f / ior; }
    const float k = 1.f - eta*eta * (1.f - negNdotV * negNdotV);
    if (k < 0.0f) { r = make_float3(0.f, 0.f, 0.f); return false; }
    float s = eta*negNdotV + std::sqrt(k);
    f3 v = make_float3(eta*i.x - s*nn.x, eta*i.y - s*nn.y, eta*i.z - s*nn.z);
    float inv = 1.0f / std::sqrt(dotf3(v, v));
    r = make_float3(v.x*inv, v.y*inv, v.z*inv);
    return true;
}

static void rtTrace(OTraceCtx& cx, const Ray& ray, PerRayData& prd, uint64_t rayIndex, int chainCol);

// ------------------------------------------------------------------ normal_shader.cu:128-340  closest hit
static void closest_hit(OTraceCtx& cx, const Ray& ray, PerRayData& prd, uint64_t rayIndex, const Hit& hit, int chainCol)
{
    const unsigned d_maxReflDepth = cx.d_maxReflDepth, d_maxRefrDepth = cx.d_maxRefrDepth;
    const OMesh& tm = cx.sc->meshes[hit.targ];
    const unsigned d_targIndex = hit.targ;
    const double d_targReflCoeff = tm.reflCoeff, d_targRefrIndex = tm.refrIndex;
    const d3 d_rayOrigin = to_double3(cx.p->rayOrigin[0], cx.p->rayOrigin[1], cx.p->rayOrigin[2]);
    const d3 normal = hit.normal; const float hit_t = hit.t;
    const uint64_t W3 = cx.stride;
    const unsigned D = cx.depthTotal;

    if ((prd.end == false) && ((prd.refrDepth < d_maxRefrDepth) || (prd.reflDepth < (d_maxReflDepth - 1)))) {
        cx.shaded++;
        if (prd.refrDepth != 1) {                                            // :140-146
            uint64_t row = rayIndex + prd.maxRayIndex;
            unsigned col = prd.reflDepth + prd.refrDepth;
            if (col < (d_maxRefrDepth + d_maxReflDepth - 1))
                cx.targ_intersect[row*D + col] = (int)(d_targIndex);
        }
        d3 hitPoint;                                                         // :149-153
        hitPoint.x = (prd.prevHitPoint).x + (double)hit_t*(prd.rayDirection).x;
        hitPoint.y = (prd.prevHitPoint).y + (double)hit_t*(prd.rayDirection).y;
        hitPoint.z = (prd.prevHitPoint).z + (double)hit_t*(prd.rayDirection).z;
        prd.rayLength += hit_t;
        if ((prd.reflDepth == 0) && (prd.refrDepth == 0)) {                  // :159-166
            prd.firstHitPoint = hitPoint;
            d3 TxRange = prd.firstHitPoint - d_rayOrigin;
            if (lengthd3(TxRange) >= SCENE_EPS) prd.power = 1/((magsquared3(TxRange))*4*M_PI);
            else prd.end = true;
        } else {                                                             // :167-173
            d3 TargRange = hitPoint - prd.prevHitPoint;
            if (lengthd3(TargRange) >= SCENE_EPS_R) prd.power *= 1/((magsquared3(TargRange))*4*M_PI);
            else prd.end = true;
        }
        prd.prevHitPoint = hitPoint;                                         // :176
        f3 hitPoint_f3 = make_float3((float)hitPoint.x, (float)hitPoint.y, (float)hitPoint.z);
        f3 new_direction;
        d3 V_targ = tm.vel;                                                  // :187
        d3 k1, k0;
        PerRayData prd_refr = prd;                                           // :191
        prd_refr.refrIndex.x = prd_refr.refrIndex.y;                         // :194

        if ( (std::fabs(d_targReflCoeff) != 1.00000f) && (prd_refr.refrDepth < d_maxRefrDepth) && (prd_refr.reflDepth == 0)) {   // :198
            if (prd_refr.refrIndex.x == 1) prd_refr.refrIndex.y = d_targRefrIndex;
            else prd_refr.refrIndex.y = 1;
            float refr_index_ratio = (float)(prd_refr.refrIndex.y/prd_refr.refrIndex.x);
            if ( refract_f3( new_direction, ray.direction, normalise_float3(normal.x, normal.y, normal.z), refr_index_ratio ) ) {
                uint64_t currentRayIndex = prd_refr.maxRayIndex + W3;        // :214 [D3]
                prd_refr.maxRayIndex = (unsigned)currentRayIndex;
                if ((prd_refr.refrDepth == 0) && (currentRayIndex == W3)) {  // :221-239
                    for (unsigned int i = 0; i < (d_maxReflDepth + d_maxRefrDepth - 1); i++)
                        cx.targ_intersect[(rayIndex + currentRayIndex)*D + i] = (int)(d_targIndex);
                    for (unsigned int j = 0; j < d_maxReflDepth; j++)
                        for (unsigned int i = 0; i < (j + 2); i++)
                            cx.targ_intersect[(rayIndex + (j + 2)*currentRayIndex)*D + i] = (int)(d_targIndex);
                }
                Ray refr_ray; refr_ray.origin = hitPoint_f3; refr_ray.direction = new_direction; refr_ray.tmin = SCENE_EPS; refr_ray.tmax = RT_DEFAULT_MAX;
                if ((prd_refr.reflDepth + 1) < d_maxReflDepth) prd_refr.power *= (1 - std::fabs(d_targReflCoeff));   // :245-246
                prd_refr.refrDepth++;
                k0 = normalised3(prd_refr.rayDirection);
                prd_refr.rayDirection = float3_to_double3(new_direction);
                k1 = normalised3(prd_refr.rayDirection);
                prd_refr.doppler += dotd3(V_targ, (k1 - k0));
                uint64_t rrow = rayIndex + currentRayIndex;                  // :259-265
                unsigned rcol = prd_refr.reflDepth + (prd_refr.refrDepth - 1);
                d2 k0_sph = cart_to_sph(k0);
                d2 k1_sph = cart_to_sph(to_double3(-k1.x, -k1.y, -k1.z));
                cx.rcs_angle[rrow*D + rcol].x = k0_sph.x + k1_sph.x;
                cx.rcs_angle[rrow*D + rcol].y = k0_sph.y + k1_sph.y;
                rtTrace(cx, refr_ray, prd_refr, rayIndex, -1);               // :268
                PerRayData& o = cx.results[rayIndex + currentRayIndex];      // :272-279
                o.reflDepth = prd_refr.reflDepth; o.refrDepth = prd_refr.refrDepth; o.rayLength = prd_refr.rayLength;
                o.firstHitPoint = prd_refr.firstHitPoint; o.prevHitPoint = prd_refr.prevHitPoint;
                o.power = prd_refr.power; o.doppler = prd_refr.doppler; o.received = prd_refr.received;
            }
        }
        prd.reflDepth++;                                                     // :286
        prd.refrIndex.y = prd_refr.refrIndex.x;                              // :289-290
        prd.refrIndex.x = prd_refr.refrIndex.x;
        if (prd.reflDepth < d_maxReflDepth) {                                // :293
            new_direction = reflect_f3( ray.direction, normalise_float3(normal.x, normal.y, normal.z) );
            Ray refl_ray; refl_ray.origin = hitPoint_f3; refl_ray.direction = new_direction; refl_ray.tmin = SCENE_EPS_R; refl_ray.tmax = RT_DEFAULT_MAX;
            prd.power *= d_targReflCoeff;
            k0 = normalised3(prd.rayDirection);
            prd.rayDirection = float3_to_double3(new_direction);
            k1 = normalised3(prd.rayDirection);
            prd.doppler += dotd3(V_targ, (k1 - k0));
            uint64_t rrow = rayIndex + prd.maxRayIndex;                      // :320-326
            unsigned rcol = (prd.reflDepth - 1) + prd.refrDepth;
            d2 k0_sph = cart_to_sph(k0);
            d2 k1_sph = cart_to_sph(to_double3(-k1.x, -k1.y, -k1.z));
            cx.rcs_angle[rrow*D + rcol].x = k0_sph.x + k1_sph.x;
            cx.rcs_angle[rrow*D + rcol].y = k0_sph.y + k1_sph.y;
            rtTrace(cx, refl_ray, prd, rayIndex, chainCol < 0 ? -1 : chainCol + 1);   // :332
        }
        if ((prd.reflDepth + 1 >= d_maxReflDepth) && (prd.refrDepth >= d_maxRefrDepth)) prd.end = true;   // :336-338
    }
}

static void rtTrace(OTraceCtx& cx, const Ray& ray, PerRayData& prd, uint64_t rayIndex, int chainCol)
{
    Hit h = find_closest(cx, ray, prd);
    if (chainCol >= 0 && cx.hit_prim && (unsigned)chainCol < cx.hitCols) {
        // oracle-only debug record of the reflection chain: global prim id and f32 t per segment
        int gid = -1;
        if (h.found) { gid = 0; for (unsigned k = 0; k < h.targ; k++) gid += (int)(cx.sc->meshes[k].tris.size()/3); gid += (int)h.prim; }
        cx.hit_prim[rayIndex*cx.hitCols + chainCol] = gid;
        cx.hit_t[rayIndex*cx.hitCols + chainCol] = h.found ? h.t : 0.0f;
    }
    if (h.found) closest_hit(cx, ray, prd, rayIndex, h, chainCol);
    else miss_program(cx, prd);
}

// ------------------------------------------------------------------ ray_tracer.cu:144-255  ray generation
static void ray_generation(OTraceCtx& cx, uint64_t localIndex, unsigned lx, unsigned ly, unsigned lz)
{
    const unsigned d_width = cx.p->width;
    const d3 d_txSpan = to_double3(cx.p->txSpan[0], cx.p->txSpan[1], cx.p->txSpan[2]);
    const d2 d_txDir = { cx.p->txDir[0], cx.p->txDir[1] };
    const d3 d_rayOrigin = to_double3(cx.p->rayOrigin[0], cx.p->rayOrigin[1], cx.p->rayOrigin[2]);
    d3 beamStart = sph_to_cart(-d_txSpan.x/2, -d_txSpan.y/2);
    d3 beamEnd = sph_to_cart(d_txSpan.x/2, d_txSpan.y/2);
    d3 rayDir_d3;
    if (d_width == 1) {
        rayDir_d3 = sph_to_cart(d_txDir.x, d_txDir.y);
    } else {
        rayDir_d3.x = beamStart.x + (((beamEnd.x*(1 + d_txSpan.z)) - beamStart.x)/(d_width - 1)) * (lx);
        rayDir_d3.y = beamStart.y + ((beamEnd.y - beamStart.y)/(d_width - 1)) * (ly);
        rayDir_d3.z = beamStart.z + ((beamEnd.z - beamStart.z)/(d_width - 1)) * (lz);
        rayDir_d3 = normalised3(rayDir_d3);
        double Rot[3][3] = {{orc_cos(d_txDir.x), -orc_sin(d_txDir.x), 0},
                            {orc_sin(d_txDir.x), orc_cos(d_txDir.x), 0},
                            {0, 0, 1}};
        d3 rotated; rotated.x = 0; rotated.y = 0; rotated.z = 0;
        rotated.x += Rot[0][0]*rayDir_d3.x + Rot[0][1]*rayDir_d3.y + Rot[0][2]*rayDir_d3.z;
        rotated.y += Rot[1][0]*rayDir_d3.x + Rot[1][1]*rayDir_d3.y + Rot[1][2]*rayDir_d3.z;
        rotated.z += Rot[2][0]*rayDir_d3.x + Rot[2][1]*rayDir_d3.y + Rot[2][2]*rayDir_d3.z;
        rayDir_d3 = normalised3(rotated);
        rotated.x = 0; rotated.y = 0; rotated.z = 0;
        rotated.x += Rot[0][1];
        rotated.y += Rot[1][1];
        rotated.z += Rot[2][1];
        d3 orth_vec = normalised3(rotated);
        double Rot1[3][3] = {{orc_cos(d_txDir.y) + orth_vec.x*orth_vec.x*(1 - orc_cos(d_txDir.y)), orth_vec.x*orth_vec.y*(1 - orc_cos(d_txDir.y)) + orth_vec.z*orc_sin(d_txDir.y), orth_vec.x*orth_vec.z*(1 - orc_cos(d_txDir.y)) - orth_vec.y*orc_sin(d_txDir.y)},
                             {orth_vec.y*orth_vec.x*(1 - orc_cos(d_txDir.y)) - orth_vec.z*orc_sin(d_txDir.y), orc_cos(d_txDir.y) + orth_vec.y*orth_vec.y*(1 - orc_cos(d_txDir.y)), orth_vec.y*orth_vec.z*(1 - orc_cos(d_txDir.y)) + orth_vec.x*orc_sin(d_txDir.y)},
                             {orth_vec.z*orth_vec.x*(1 - orc_cos(d_txDir.y)) + orth_vec.y*orc_sin(d_txDir.y), orth_vec.z*orth_vec.y*(1 - orc_cos(d_txDir.y)) - orth_vec.x*orc_sin(d_txDir.y), orc_cos(d_txDir.y) + orth_vec.z*orth_vec.z*(1 - orc_cos(d_txDir.y))}};
        rotated.x = 0; rotated.y = 0; rotated.z = 0;
        rotated.x += Rot1[0][0]*rayDir_d3.x + Rot1[0][1]*rayDir_d3.y + Rot1[0][2]*rayDir_d3.z;
        rotated.y += Rot1[1][0]*rayDir_d3.x + Rot1[1][1]*rayDir_d3.y + Rot1[1][2]*rayDir_d3.z;
        rotated.z += Rot1[2][0]*rayDir_d3.x + Rot1[2][1]*rayDir_d3.y + Rot1[2][2]*rayDir_d3.z;
        rayDir_d3 = rotated;
    }
    f3 rayDir_f3 = normalise_float3(rayDir_d3.x, rayDir_d3.y, rayDir_d3.z);                     // :208
    Ray ray; ray.origin = make_float3((float)d_rayOrigin.x, (float)d_rayOrigin.y, (float)d_rayOrigin.z);
    ray.direction = rayDir_f3; ray.tmin = SCENE_EPS; ray.tmax = RT_DEFAULT_MAX;                // :209
    PerRayData prd;                                                                             // :212-224
    memset(&prd, 0, sizeof(prd));
    prd.reflDepth = 0; prd.refrDepth = 0; prd.maxRayIndex = 0; prd.rayLength = 0;
    prd.rayDirection = rayDir_d3;
    prd.firstHitPoint = to_double3(0.f, 0.f, 0.f);
    prd.prevHitPoint = d_rayOrigin;
    prd.refrIndex.x = 1; prd.refrIndex.y = 1;
    prd.power = 0; prd.doppler = 0; prd.received = -1; prd.end = false;
    // output slots are pre-initialised by the caller (ray_tracer.cu:227-240)
    rtTrace(cx, ray, prd, localIndex, 0);                                                       // :243
    PerRayData& o = cx.results[localIndex];                                                     // :246-253
    o.reflDepth = prd.reflDepth; o.refrDepth = prd.refrDepth; o.rayLength = prd.rayLength;
    o.firstHitPoint = prd.firstHitPoint; o.prevHitPoint = prd.prevHitPoint;
    o.power = prd.power; o.doppler = prd.doppler; o.received = prd.received;
}

// =====================================================================================
//                                   extern "C" API (ctypes)
// =====================================================================================
extern "C" {

// branch coverage of the miss program since the last reset (see the COV_* list above); out[COV_N]
unsigned int orc_coverage(unsigned long long* out, int reset)
{
    for (int k = 0; k < COV_N; k++) { if (out) out[k] = g_cov[k].load(); if (reset) g_cov[k].store(0); }
    return COV_N;
}

void* orc_scene_create() { return new OScene(); }
void orc_scene_destroy(void* s) { delete (OScene*)s; }
void orc_scene_clear_meshes(void* s) { ((OScene*)s)->meshes.clear(); ((OScene*)s)->bvhBuilt = false; }

// vertices are WORLD space (ray_tracer.cpp:1010-1014 already applied by the caller)
void orc_scene_add_mesh(void* s, const uint32_t* tris, uint32_t ntris, const double* verts, uint32_t nverts,
                        const double* normals, uint32_t nnormals, double reflCoeff, double refrIndex, const double* vel)
{
    OScene* sc = (OScene*)s; OMesh m;
    m.tris.assign(tris, tris + 3*(size_t)ntris);
    m.verts.resize(nverts); for (uint32_t i = 0; i < nverts; i++) m.verts[i] = to_double3(verts[3*i], verts[3*i+1], verts[3*i+2]);
    m.normals.resize(nnormals); for (uint32_t i = 0; i < nnormals; i++) m.normals[i] = to_double3(normals[3*i], normals[3*i+1], normals[3*i+2]);
    m.reflCoeff = reflCoeff; m.refrIndex = refrIndex; m.vel = to_double3(vel[0], vel[1], vel[2]);
    sc->meshes.push_back(std::move(m)); sc->bvhBuilt = false;
}

// receiver sphere buffers, ray_tracer.cu:33-38
void orc_set_receivers(void* s, uint32_t n, const double* centre, const double* radius, const double* minTheta,
                       const double* maxTheta, const double* minPhi, const double* maxPhi)
{
    OScene* sc = (OScene*)s;
    sc->sphCentre.resize(n); sc->sphRadius.assign(radius, radius+n); sc->minTheta.assign(minTheta, minTheta+n);
    sc->maxTheta.assign(maxTheta, maxTheta+n); sc->minPhi.assign(minPhi, minPhi+n); sc->maxPhi.assign(maxPhi, maxPhi+n);
    for (uint32_t i = 0; i < n; i++) sc->sphCentre[i] = to_double3(centre[3*i], centre[3*i+1], centre[3*i+2]);
}

// rows per launch ray: 1 without refraction, maxRefl + 3 with (ray_tracer.cpp:608-626)
uint32_t orc_rows_per_ray(uint32_t maxRefl, uint32_t maxRefr) { return maxRefr == 2 ? (maxRefl + 1) + 1 + 1 : 1; }

// Trace launch indices  ray_first + k*ray_stride (k < n_rays) of the W^3 launch.
//   results        [rows_per_ray * n_rays]       PerRayData
//   targ_intersect [rows_per_ray * n_rays * D]   int     (D = maxRefr + maxRefl)
//   rcs_angle      [rows_per_ray * n_rays * D]   double2
//   hit_prim/hit_t [n_rays * (maxRefl+1)]        oracle-only debug (may be NULL)
//   counters       [4]: node visits, triangle tests, segments, shaded hits (may be NULL)
int orc_trace(void* s, const OPulse* p, uint64_t ray_first, uint64_t ray_stride, uint64_t n_rays, int use_bvh, int n_threads,
              PerRayData* results, int* targ_intersect, double* rcs_angle, int* hit_prim, float* hit_t, uint64_t* counters)
{
    OScene* sc = (OScene*)s;
    if (use_bvh && !sc->bvhBuilt) bvh_build(*sc);
    const unsigned D = p->maxRefr + p->maxRefl;
    const uint64_t rows = (uint64_t)orc_rows_per_ray(p->maxRefl, p->maxRefr) * n_rays;
    const unsigned hitCols = p->maxRefl + 1;
    if (n_threads < 1) n_threads = 1;
    // host pre-fill: ray_tracer.cpp:854-868 and ray_tracer.cu:227-240 (done by the workers, chunk by chunk, so that a
    // many-thread run is not dominated by one thread touching gigabytes of output)
    auto prefill = [&](uint64_t r0, uint64_t r1) {
        for (uint64_t i = r0; i < r1; i++) {
            PerRayData& o = results[i]; memset(&o, 0, sizeof(o));
            o.refrIndex.x = 1; o.refrIndex.y = 1; o.received = -1; o.end = false;
        }
        for (uint64_t i = r0*D; i < r1*D; i++) { targ_intersect[i] = -1; rcs_angle[2*i] = -1000000; rcs_angle[2*i+1] = -1000000; }
    };
    std::vector<OTraceCtx> ctxs(n_threads);
    // RTS_ORACLE_PIN=1 (the timed CPU baseline of bench.py): worker t runs on the t-th CPU of the process's affinity mask -- one thread per
    // hardware thread, no migration -- so that a pulse's time does not depend on where the scheduler happened to put 256 fresh threads
    std::vector<int> pin_cpus;
    if (const char* e = getenv("RTS_ORACLE_PIN")) if (e[0] == '1') { cpu_set_t m; CPU_ZERO(&m); if (sched_getaffinity(0, sizeof(m), &m) == 0) for (int k = 0; k < CPU_SETSIZE; k++) if (CPU_ISSET(k, &m)) pin_cpus.push_back(k); }
    const uint64_t CHUNK = 2048;                                   // launch indices per work unit, handed out dynamically:
    std::atomic<uint64_t> next_fill(0), next_ray(0), filled(0);    // rays that hit cluster in launch-index space
    auto worker = [&](int tid) {
        if (!pin_cpus.empty() && n_threads > 1) { cpu_set_t m; CPU_ZERO(&m); CPU_SET(pin_cpus[(size_t)tid % pin_cpus.size()], &m); (void)pthread_setaffinity_np(pthread_self(), sizeof(m), &m); }
        OTraceCtx& cx = ctxs[tid];
        cx.sc = sc; cx.p = p; cx.d_maxReflDepth = p->maxRefl + 1; cx.d_maxRefrDepth = p->maxRefr; cx.depthTotal = D;
        cx.stride = n_rays; cx.useBvh = use_bvh != 0; cx.results = results; cx.targ_intersect = targ_intersect; cx.rcs_angle = (d2*)rcs_angle;
        cx.hit_prim = hit_prim; cx.hit_t = hit_t; cx.hitCols = hitCols;
        cx.nodeVisits = cx.triTests = cx.segments = cx.shaded = 0;
        const uint64_t W = p->width;
        for (;;) {                                                 // phase 1: pre-fill every output row
            const uint64_t r0 = next_fill.fetch_add(CHUNK); if (r0 >= rows) break;
            const uint64_t r1 = std::min(rows, r0 + CHUNK);
            prefill(r0, r1);
            if (hit_prim) for (uint64_t i = r0; i < std::min(r1, n_rays); i++) for (unsigned c = 0; c < hitCols; c++) { hit_prim[i*hitCols + c] = -2; hit_t[i*hitCols + c] = 0.0f; }
            filled.fetch_add(r1 - r0);
        }
        while (filled.load() < rows) std::this_thread::yield();    // a refracted child writes rows of other chunks: all rows first
        for (;;) {                                                 // phase 2: trace
            const uint64_t lo = next_ray.fetch_add(CHUNK); if (lo >= n_rays) break;
            const uint64_t hi = std::min(n_rays, lo + CHUNK);
            for (uint64_t k = lo; k < hi; k++) {
                uint64_t g = ray_first + k*ray_stride;                         // rayIndex = z*W*W + y*W + x  (ray_tracer.cu:151)
                unsigned lx = (unsigned)(g % W), ly = (unsigned)((g / W) % W), lz = (unsigned)(g / (W*W));
                ray_generation(cx, k, lx, ly, lz);
            }
        }
    };
    if (n_threads == 1) worker(0);
    else { std::vector<std::thread> th; for (int t = 0; t < n_threads; t++) th.emplace_back(worker, t); for (auto& t : th) t.join(); }
    for (auto& c : ctxs) for (int k = 0; k < COV_N; k++) if (c.cov[k]) g_cov[k].fetch_add(c.cov[k], std::memory_order_relaxed);
    if (counters) { counters[0] = counters[1] = counters[2] = counters[3] = 0;
        for (auto& c : ctxs) { counters[0] += c.nodeVisits; counters[1] += c.triTests; counters[2] += c.segments; counters[3] += c.shaded; } }
    return 0;
}

// ---------------------------------------------------------------- triangle_mesh.cu:204-233
int orc_bound(const double* v0, const double* v1, const double* v2, float* out6)
{
    return prim_bound(to_double3(v0[0],v0[1],v0[2]), to_double3(v1[0],v1[1],v1[2]), to_double3(v2[0],v2[1],v2[2]), out6) ? 1 : 0;
}

float orc_atan2f(float y, float x) { return orc_atan2f_cr(y, x); }
float orc_libm_atan2f(float y, float x) { return atan2f(y, x); }

// ---------------------------------------------------------------- ray_tracer.cpp:894-918  receiver sphere set-up (host, glibc float trig)
void orc_rx_sphere(const double* repos, double az, double el, double radius, double thetaSpan, double phiSpan, double* out9)
{
    double h_Rx_azimuth = az, h_Rx_elevation = el;
    double cx = repos[0] + (radius * orc_cosf(h_Rx_elevation) * orc_cosf(h_Rx_azimuth));
    double cy = repos[1] + (radius * orc_cosf(h_Rx_elevation) * orc_sinf(h_Rx_azimuth));
    double cz = repos[2] + (radius * orc_sinf(h_Rx_elevation));
    h_Rx_azimuth = atan2f((repos[1] - cy), (repos[0] - cx));
    h_Rx_elevation = atan2f((repos[2] - cz), sqrt((repos[0] - cx)*(repos[0] - cx) + (repos[1] - cy)*(repos[1] - cy)));
    out9[0] = cx; out9[1] = cy; out9[2] = cz; out9[3] = radius;
    out9[4] = h_Rx_azimuth - thetaSpan/2; out9[5] = h_Rx_azimuth + thetaSpan/2;
    out9[6] = h_Rx_elevation - phiSpan/2; out9[7] = h_Rx_elevation + phiSpan/2;
    out9[8] = 0;
}

// ---------------------------------------------------------------- ray_tracer.cpp:120-170  rotations
typedef std::vector<std::vector<double>> Mat;
static Mat matrix_multiply(Mat M1, Mat M2) {
    Mat M3(M1.size(), std::vector<double>(M2[0].size(), 0));
    for (unsigned i = 0; i < M1.size(); i++) for (unsigned j = 0; j < M2[0].size(); j++) {
        M3[i][j] = 0; for (unsigned k = 0; k < M2.size(); k++) M3[i][j] += M1[i][k] * M2[k][j]; }
    return M3;
}
static Mat matrix_transpose(Mat M1) {
    Mat M2(M1[0].size(), std::vector<double>(M1.size(), 0));
    for (unsigned i = 0; i < M1.size(); i++) for (unsigned j = 0; j < M1[0].size(); j++) M2[j][i] = M1[i][j];
    return M2;
}
static Mat vertex_rotation(Mat vertices, float yaw, float pitch, float roll) {
    Mat Rx = {{1, 0, 0}, {0, orc_cosf(roll), -orc_sinf(roll)}, {0, orc_sinf(roll), orc_cosf(roll)}};
    Mat Ry = {{orc_cosf(pitch), 0, orc_sinf(pitch)}, {0, 1, 0}, {-orc_sinf(pitch), 0, orc_cosf(pitch)}};
    Mat Rz = {{orc_cosf(yaw), -orc_sinf(yaw), 0}, {orc_sinf(yaw), orc_cosf(yaw), 0}, {0, 0, 1}};
    Mat R_total = matrix_multiply(Rz, matrix_multiply(Ry, Rx));
    if (vertices.empty()) return vertices;
    return matrix_transpose(matrix_multiply(R_total, matrix_transpose(vertices)));
}
void orc_vertex_rotation(double* verts, uint32_t n, float yaw, float pitch, float roll) {
    Mat v(n, std::vector<double>(3)); for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) v[i][k] = verts[3*i+k];
    v = vertex_rotation(v, yaw, pitch, roll);
    for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) verts[3*i+k] = v[i][k];
}

// ---------------------------------------------------------------- ray_tracer.cpp:226-297  rect mesh
void orc_rect_mesh(float w, float h, float d, float yaw, float pitch, float roll, double* verts24, uint32_t* tris36, double* normals36)
{
    Mat vertices(8, std::vector<double>(3));
    vertices[0][0] = w*+0.5f; vertices[0][1] = h*-0.5f; vertices[0][2] = d*-0.5f;
    vertices[1][0] = w*+0.5f; vertices[1][1] = h*+0.5f; vertices[1][2] = d*-0.5f;
    vertices[2][0] = w*+0.5f; vertices[2][1] = h*-0.5f; vertices[2][2] = d*+0.5f;
    vertices[3][0] = w*+0.5f; vertices[3][1] = h*+0.5f; vertices[3][2] = d*+0.5f;
    vertices[4][0] = w*-0.5f; vertices[4][1] = h*-0.5f; vertices[4][2] = d*-0.5f;
    vertices[5][0] = w*-0.5f; vertices[5][1] = h*+0.5f; vertices[5][2] = d*-0.5f;
    vertices[6][0] = w*-0.5f; vertices[6][1] = h*-0.5f; vertices[6][2] = d*+0.5f;
    vertices[7][0] = w*-0.5f; vertices[7][1] = h*+0.5f; vertices[7][2] = d*+0.5f;
    static const unsigned T[12][3] = {{0,1,2},{1,3,2},{2,3,7},{2,7,6},{1,7,3},{1,5,7},{6,7,4},{7,5,4},{0,4,1},{1,4,5},{2,6,4},{0,2,4}};
    vertices = vertex_rotation(vertices, yaw, pitch, roll);
    for (int i = 0; i < 12; i++) {
        double v1[3], v2[3], f[3];
        for (int k = 0; k < 3; k++) { v1[k] = vertices[T[i][1]][k] - vertices[T[i][0]][k]; v2[k] = vertices[T[i][2]][k] - vertices[T[i][0]][k]; }
        f[0] = (v1[1]*v2[2] - v1[2]*v2[1]); f[1] = (v1[2]*v2[0] - v1[0]*v2[2]); f[2] = (v1[0]*v2[1] - v1[1]*v2[0]);
        double norm = sqrt(f[0]*f[0] + f[1]*f[1] + f[2]*f[2]);
        for (int k = 0; k < 3; k++) { normals36[3*i+k] = f[k]/norm; tris36[3*i+k] = T[i][k]; }
    }
    for (int i = 0; i < 8; i++) for (int k = 0; k < 3; k++) verts24[3*i+k] = vertices[i][k];
}

// ---------------------------------------------------------------- ray_tracer.cpp:85-101,300-426  icosphere
// Literal restatement, including the std::set de-duplication that fixes vertex ORDER
// (lexicographic on exact doubles) and triangle ORDER (lexicographic on index triples).
// Sizes: 20*4^n triangles, 10*4^n + 2 vertices.  Call with NULL outputs to get sizes.
void orc_sphere_mesh(uint32_t n, float radius, float yaw, float pitch, float roll,
                     double* verts, uint32_t* nverts, uint32_t* tris, uint32_t* ntris, double* normals)
{
    double t = (1 + sqrt(5)) / 2;
    Mat v = {{-1, t, 0},{1, t, 0},{-1, -t, 0},{1, -t, 0},{0, -1, t},{0, 1, t},{0, -1, -t},{0, 1, -t},{t, 0, -1},{t, 0, 1},{-t, 0, -1},{-t, 0, 1}};
    for (unsigned i = 0; i < v.size(); i++) {
        double norm = sqrt(v[i][0]*v[i][0] + v[i][1]*v[i][1] + v[i][2]*v[i][2]);
        v[i][0] = v[i][0]/norm; v[i][1] = v[i][1]/norm; v[i][2] = v[i][2]/norm;
    }
    std::vector<std::vector<unsigned>> f = {{0,11,5},{0,5,1},{0,1,7},{0,7,10},{0,10,11},{1,5,9},{5,11,4},{11,10,2},{10,7,6},{7,1,8},
                                            {3,9,4},{3,4,2},{3,2,6},{3,6,8},{3,8,9},{4,9,5},{2,4,11},{6,2,10},{8,6,7},{9,8,1}};
    auto getMidPoint = [&](int t1, int t2) {
        std::vector<double> pm(3, 0);
        pm[0] = (v[t1][0] + v[t2][0])/2; pm[1] = (v[t1][1] + v[t2][1])/2; pm[2] = (v[t1][2] + v[t2][2])/2;
        double norm = sqrt(pm[0]*pm[0] + pm[1]*pm[1] + pm[2]*pm[2]);
        pm[0] = pm[0]/norm; pm[1] = pm[1]/norm; pm[2] = pm[2]/norm;
        v.push_back(pm);
    };
    for (unsigned gen = 0; gen < n; gen++) {
        std::vector<std::vector<unsigned>> f_(f.size()*4, std::vector<unsigned>(3, 0));
        for (unsigned i = 0; i < f.size(); i++) {
            int tri[3] = {(int)f[i][0], (int)f[i][1], (int)f[i][2]};
            int a = v.size(); getMidPoint(tri[0], tri[1]);
            int b = v.size(); getMidPoint(tri[1], tri[2]);
            int c = v.size(); getMidPoint(tri[2], tri[0]);
            int nfc[4][3] = {{tri[0], a, c},{tri[1], b, a},{tri[2], c, b},{a, b, c}};
            for (unsigned j = 0; j < 4; j++) { int idx = (4*i) + j; f_[idx][0] = nfc[j][0]; f_[idx][1] = nfc[j][1]; f_[idx][2] = nfc[j][2]; }
        }
        f = f_;
    }
    std::set<std::vector<double>> v_unique(v.begin(), v.end());
    // index of each vertex in the sorted unique set (the reference does this with an O(V^2)
    // std::find over the set; a binary search over the same ordering gives the same index)
    std::vector<std::vector<double>> verts_sorted(v_unique.begin(), v_unique.end());
    std::vector<int> ix(v.size());
    for (size_t i = 0; i < v.size(); i++)
        ix[i] = (int)(std::lower_bound(verts_sorted.begin(), verts_sorted.end(), v[i]) - verts_sorted.begin());
    Mat vertices = vertex_rotation(verts_sorted, yaw, pitch, roll);
    Mat vert_normals = vertices;
    for (unsigned i = 0; i < f.size(); i++) { f[i][0] = ix[f[i][0]]; f[i][1] = ix[f[i][1]]; f[i][2] = ix[f[i][2]]; }
    std::set<std::vector<unsigned>> f_unique(f.begin(), f.end());
    std::vector<std::vector<unsigned>> trisv(f_unique.begin(), f_unique.end());
    for (unsigned i = 0; i < vertices.size(); i++) { vertices[i][0] *= radius; vertices[i][1] *= radius; vertices[i][2] *= radius; }
    if (nverts) *nverts = (uint32_t)vertices.size();
    if (ntris) *ntris = (uint32_t)trisv.size();
    if (verts) for (size_t i = 0; i < vertices.size(); i++) for (int k = 0; k < 3; k++) { verts[3*i+k] = vertices[i][k]; normals[3*i+k] = vert_normals[i][k]; }
    if (tris) for (size_t i = 0; i < trisv.size(); i++) for (int k = 0; k < 3; k++) tris[3*i+k] = trisv[i][k];
}

// ---------------------------------------------------------------- ray_tracer.cpp:429-504  file mesh
// returns number of triangles (lines), or -1 on open failure (the reference exit()s).
int orc_file_mesh(const char* v_file, const char* n_file, float yaw, float pitch, float roll,
                  double* verts, uint32_t* tris, double* normals, uint32_t cap_tris)
{
    FILE* fp = fopen(v_file, "r"); if (!fp) return -1;
    unsigned nt = 0; int ch; while ((ch = fgetc(fp)) != EOF) if (ch == '\n') nt++;
    rewind(fp);
    if (!verts) { fclose(fp); return (int)nt; }
    if (nt > cap_tris) { fclose(fp); return -2; }
    Mat vertices(nt*3, std::vector<double>(3)), vert_normals(nt*3, std::vector<double>(3));
    for (unsigned i = 0; i < nt; i++) {
        if (fscanf(fp, "%lf %lf %lf, %lf %lf %lf, %lf %lf %lf,\n", &vertices[3*i][0], &vertices[3*i][1], &vertices[3*i][2],
                   &vertices[3*i+1][0], &vertices[3*i+1][1], &vertices[3*i+1][2],
                   &vertices[3*i+2][0], &vertices[3*i+2][1], &vertices[3*i+2][2]) == EOF) { fclose(fp); return -3; }
    }
    fclose(fp);
    vertices = vertex_rotation(vertices, yaw, pitch, roll);
    fp = fopen(n_file, "r"); if (!fp) return -1;
    for (unsigned i = 0; i < nt; i++) {
        if (fscanf(fp, "%lf %lf %lf, %lf %lf %lf, %lf %lf %lf,\n", &vert_normals[3*i][0], &vert_normals[3*i][1], &vert_normals[3*i][2],
                   &vert_normals[3*i+1][0], &vert_normals[3*i+1][1], &vert_normals[3*i+1][2],
                   &vert_normals[3*i+2][0], &vert_normals[3*i+2][1], &vert_normals[3*i+2][2]) == EOF) { fclose(fp); return -3; }
    }
    fclose(fp);
    vert_normals = vertex_rotation(vert_normals, yaw, pitch, roll);
    for (unsigned i = 0; i < nt*3; i++) for (int k = 0; k < 3; k++) { verts[3*i+k] = vertices[i][k]; normals[3*i+k] = vert_normals[i][k]; }
    for (unsigned i = 0; i < nt; i++) { tris[3*i] = i*3 + 0; tris[3*i+1] = i*3 + 1; tris[3*i+2] = i*3 + 2; }
    return (int)nt;
}

// ---------------------------------------------------------------- ray_tracer.cpp:1190-1258  host filter + finalise
// RCS, Gt, Gr come from SOARS virtual calls in the reference; here the caller supplies a
// constant RCS per target and constant gains (the synthetic scenes use isotropic = 1).
// Returns the number of received rays; outputs are the compacted arrays in ascending slot order.
uint64_t orc_filter_finalise(const PerRayData* results, const int* targ_intersect, uint64_t rayTotal, uint32_t D,
                             const double* rcs_per_target, double Wl, double Gt, double Gr, double carrier, double cspeed,
                             PerRayData* rx_results, int* rx_intersects, uint64_t* rx_slots)
{
    uint64_t receivedRays = 0;
    for (uint64_t i = 0; i < rayTotal; i++) {
        if (results[i].received >= 0) {
            PerRayData r = results[i];
            for (unsigned k = 0; k < D; k++) {
                uint64_t depth_ray_index = k + i*D;
                int targ_k = targ_intersect[depth_ray_index];
                rx_intersects[receivedRays*D + k] = targ_k;
                if (targ_k >= 0) { double targRCS = rcs_per_target[targ_k]; r.power *= targRCS; }
            }
            r.power *= (Wl*Wl*Gt*Gr);
            double Vr = r.doppler/2;
            r.doppler = carrier*(((1 + Vr/cspeed)/(1 - Vr/cspeed)) - 1);
            rx_results[receivedRays] = r;
            if (rx_slots) rx_slots[receivedRays] = i;
            receivedRays++;
        }
    }
    return receivedRays;
}

// ---------------------------------------------------------------- ray_tracer.cpp:1190-1258 with the simulator's callbacks
// The same loop with everything the reference asks SOARS for handed over as callbacks, so that tests can give antennas and
// targets patterns that depend on every argument: transvec / recvvec (:1204-1211; the direct-ray branch uses the Rx
// position, NOT the end point), GetRCS(rcs_angle.x, rcs_angle.y, Wl) per intersected depth (:1219-1230),
// GetGain(transvec, trans->GetRotation(time_t), Wl) and GetGain(recvvec, recv->GetRotation(delay + time_t), Wl) (:1233-1235).
// SVec3 is a SOARS type (absent from the reference): the callbacks get the Cartesian vector the reference constructs it
// from -- its length is overwritten with 1 (:1217-1218) -- and the time the rotation is asked at.
typedef double (*orc_rcs_fn)(void* user, int targ, double az, double el, double wl);
typedef double (*orc_gain_fn)(void* user, int is_rx, int index, double vx, double vy, double vz, double rot_time, double wl);
uint64_t orc_filter_finalise_cb(const PerRayData* results, const int* targ_intersect, const double* rcs_angle /* [rayTotal][D][2] */,
                                uint64_t rayTotal, uint32_t D, const double* h_rayOrigin, const double* rx_positions /* [n_rx][3] */,
                                int tx_index, double time_t, double Wl, double carrier, double cspeed,
                                orc_rcs_fn get_rcs, orc_gain_fn get_gain, void* user,
                                PerRayData* rx_results, int* rx_intersects, uint64_t* rx_slots)
{
    uint64_t receivedRays = 0;
    for (uint64_t i = 0; i < rayTotal; i++) {
        if (results[i].received >= 0) {
            PerRayData r = results[i];
            const double* repos = rx_positions + 3*(size_t)r.received;            // GetPosition(0)
            double transvec[3], recvvec[3];
            if ((r.reflDepth == 0) && (r.refrDepth == 0)) {
                for (int k = 0; k < 3; k++) { transvec[k] = h_rayOrigin[k] - repos[k]; recvvec[k] = repos[k] - h_rayOrigin[k]; }
            } else {
                transvec[0] = r.firstHitPoint.x - h_rayOrigin[0]; transvec[1] = r.firstHitPoint.y - h_rayOrigin[1]; transvec[2] = r.firstHitPoint.z - h_rayOrigin[2];
                recvvec[0] = r.prevHitPoint.x - repos[0]; recvvec[1] = r.prevHitPoint.y - repos[1]; recvvec[2] = r.prevHitPoint.z - repos[2];
            }
            double delay = (r.rayLength)/cspeed;
            for (unsigned k = 0; k < D; k++) {
                uint64_t depth_ray_index = k + i*D;
                int targ_k = targ_intersect[depth_ray_index];
                rx_intersects[receivedRays*D + k] = targ_k;
                if (targ_k >= 0) {
                    double targRCS = get_rcs(user, targ_k, rcs_angle[2*depth_ray_index], rcs_angle[2*depth_ray_index + 1], Wl);
                    r.power *= targRCS;
                }
            }
            double Gt = get_gain(user, 0, tx_index, transvec[0], transvec[1], transvec[2], time_t, Wl);
            double Gr = get_gain(user, 1, r.received, recvvec[0], recvvec[1], recvvec[2], delay + time_t, Wl);
            r.power *= (Wl*Wl*Gt*Gr);
            double Vr = r.doppler/2;
            r.doppler = carrier*(((1 + Vr/cspeed)/(1 - Vr/cspeed)) - 1);
            rx_results[receivedRays] = r;
            if (rx_slots) rx_slots[receivedRays] = i;
            receivedRays++;
        }
    }
    return receivedRays;
}

// ---------------------------------------------------------------- aggregation.cu:32-97  myKernel1 + myKernel2, literal O(R^2 D)
void orc_aggregate_literal(PerRayData* results_arr, const int* targ_intersect_arr, unsigned receivedRays, unsigned depthTotal,
                           double cspeed, double carrier, double* npath_arr, double* power_arr, double* doppler_arr,
                           double* delay_arr, double* phase_arr, int* pathMatch)
{
    for (unsigned i = 0; i < receivedRays; i++) {                       // myKernel1
        for (unsigned r = 0; r < receivedRays; r++) {
            if (results_arr[i].received == results_arr[r].received) {
                bool row_equal = true;
                for (unsigned k = 0; k < depthTotal; k++)
                    if (targ_intersect_arr[k + i*depthTotal] != targ_intersect_arr[k + r*depthTotal]) { row_equal = false; break; }
                if ((row_equal == true) || ((results_arr[i].reflDepth == 0) && (results_arr[i].refrDepth == 0))) {
                    double delay = (results_arr[r].rayLength)/cspeed;
                    double phase = -fmod(delay*2*M_PI*carrier, 2*M_PI);
                    npath_arr[i] += 1;
                    power_arr[i] += sqrt(results_arr[r].power);
                    delay_arr[i] += delay;
                    phase_arr[i] += phase;
                    doppler_arr[i] += results_arr[r].doppler;
                    if ((int)r < pathMatch[i]) pathMatch[i] = r;
                }
            }
        }
    }
    for (unsigned i = 0; i < receivedRays; i++) {                       // myKernel2
        if (npath_arr[i] > 0) {
            double v = power_arr[i]/npath_arr[i];
            results_arr[i].power = v*v;                                  // [D2] pow(v, 2)
            delay_arr[i] /= npath_arr[i];
            phase_arr[i] /= npath_arr[i];
            results_arr[i].doppler = doppler_arr[i]/npath_arr[i];
        }
    }
}

// ---------------------------------------------------------------- complex return cube (north-star product, NOT in the reference)
// Definition, two variants of cube[rx][pulse][bin] += A e^{j phi}, bin = floor((delay - t0) / dt):
//   mode 0, per received RAY (the coherent sum):  A = sqrt(power_i) of the FINALISED ray (ray_tracer.cpp:1247 applied),
//           delay_i = rayLength_i / c, phi_i = -fmod(2 pi fc delay_i, 2 pi)            -- the per-ray terms of aggregation.cu:59-60
//   mode 1, per UNIQUE PATH (one term per response the reference emits, ray_tracer.cpp:1290-1321): the representative rays
//           u = unique(pathMatch) with the group's power, mean delay and mean phase that myKernel2 left (aggregation.cu:88-93)
// results / delay / phase / pathMatch: for mode 1 the outputs of orc_aggregate_literal on the pulse's received rays.
void orc_cube(const PerRayData* results, unsigned receivedRays, const double* delay_arr, const double* phase_arr, const int* pathMatch, int mode,
              unsigned n_rx, unsigned n_pulses, unsigned n_bins, unsigned pulse, double t0, double dt, double cspeed, double carrier, double* cube)
{
    for (unsigned i = 0; i < receivedRays; i++) {
        const PerRayData& r = results[i];
        if (r.received < 0 || (unsigned)r.received >= n_rx) continue;
        double delay, phase;
        if (mode == 0) { delay = (r.rayLength)/cspeed; phase = -fmod(delay*2*M_PI*carrier, 2*M_PI); }
        else { if (pathMatch[i] != (int)i) continue; delay = delay_arr[i]; phase = phase_arr[i]; }
        const double b = floor((delay - t0) / dt);
        if (!(b >= 0.0) || !(b < (double)n_bins)) continue;
        const double amp = sqrt(r.power);
        double* cell = cube + 2 * (((size_t)r.received * n_pulses + pulse) * n_bins + (size_t)b);
        cell[0] += amp * orc_cos(phase); cell[1] += amp * orc_sin(phase);
    }
}

// ray_tracer.cpp:1290-1292  sort + unique of pathMatch; returns count, writes ascending unique values
unsigned orc_unique_paths(const int* pathMatch, unsigned receivedRays, int* out)
{
    std::vector<int> u(pathMatch, pathMatch + receivedRays);
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    for (size_t i = 0; i < u.size(); i++) out[i] = u[i];
    return (unsigned)u.size();
}

uint32_t orc_sizeof_prd() { return (uint32_t)sizeof(PerRayData); }

} // extern "C"
