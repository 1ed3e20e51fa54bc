// rts_mesh.cpp -- host scene helpers of the C-ABI: the mesh builders and rigid rotation of
// the reference's host driver (ray_tracer.cpp:85-504) and the receiver-sphere set-up
// (ray_tracer.cpp:894-918).  These define the INPUTS of the device path (vertex values,
// vertex order, triangle order => primitive ids), so they reproduce the reference's
// arithmetic and ordering exactly; the data structures are flat arrays instead of
// vector<vector<double>>, and the icosphere de-duplication is a sort instead of the
// reference's O(V^2) std::find over a std::set.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>
#include "../../include/rts_amd.h"

void rts_set_error(const char* fmt, ...);

// R_total = Rz * (Ry * Rx) with FLOAT trigonometry (yaw/pitch/roll are float arguments, so
// std::cos/std::sin resolve to the float overloads; ray_tracer.cpp:156-162), products
// accumulated from zero in k order (matrix_multiply, :120-137).
static void rotation_matrix(float yaw, float pitch, float roll, double R[3][3])
{
    const double Rx[3][3] = {{1, 0, 0}, {0, std::cos(roll), -std::sin(roll)}, {0, std::sin(roll), std::cos(roll)}};
    const double Ry[3][3] = {{std::cos(pitch), 0, std::sin(pitch)}, {0, 1, 0}, {-std::sin(pitch), 0, std::cos(pitch)}};
    const double Rz[3][3] = {{std::cos(yaw), -std::sin(yaw), 0}, {std::sin(yaw), std::cos(yaw), 0}, {0, 0, 1}};
    double T[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Ry[i][k] * Rx[k][j]; T[i][j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rz[i][k] * T[k][j]; R[i][j] = s; }
}

static void rotate_in_place(double* v, size_t n, const double R[3][3])
{
    for (size_t p = 0; p < n; p++) {
        const double x = v[3*p], y = v[3*p+1], z = v[3*p+2];
        for (int i = 0; i < 3; i++) { double s = 0; s += R[i][0] * x; s += R[i][1] * y; s += R[i][2] * z; v[3*p+i] = s; }
    }
}

extern "C" int rts_rotation_matrix(float yaw, float pitch, float roll, double* r9)
{
    if (!r9) { rts_set_error("rts_rotation_matrix: null output"); return RTS_ERR_INVALID; }
    double R[3][3]; rotation_matrix(yaw, pitch, roll, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r9[3*i+j] = R[i][j];
    return RTS_OK;
}

extern "C" int rts_vertex_rotation(double* vertices, uint32_t n, float yaw, float pitch, float roll)
{
    if (n && !vertices) { rts_set_error("rts_vertex_rotation: null vertices"); return RTS_ERR_INVALID; }
    double R[3][3]; rotation_matrix(yaw, pitch, roll, R);
    rotate_in_place(vertices, n, R);
    return RTS_OK;
}

// "rect": 8 vertices, 12 triangles, 12 unit FACE normals returned in the normals slot
// (ray_tracer.cpp:226-297).  Extents are float products (w*+0.5f is a float multiply).
extern "C" int rts_rect_mesh(float w, float h, float d, float yaw, float pitch, float roll, double* V, uint32_t* T, double* N)
{
    if (!V || !T || !N) { rts_set_error("rts_rect_mesh: null output"); return RTS_ERR_INVALID; }
    static const float sx[8] = {+0.5f, +0.5f, +0.5f, +0.5f, -0.5f, -0.5f, -0.5f, -0.5f};
    static const float sy[8] = {-0.5f, +0.5f, -0.5f, +0.5f, -0.5f, +0.5f, -0.5f, +0.5f};
    static const float sz[8] = {-0.5f, -0.5f, +0.5f, +0.5f, -0.5f, -0.5f, +0.5f, +0.5f};
    for (int i = 0; i < 8; i++) { V[3*i] = w*sx[i]; V[3*i+1] = h*sy[i]; V[3*i+2] = d*sz[i]; }
    static const uint32_t F[12][3] = {{0,1,2},{1,3,2},{2,3,7},{2,7,6},{1,7,3},{1,5,7},{6,7,4},{7,5,4},{0,4,1},{1,4,5},{2,6,4},{0,2,4}};
    double R[3][3]; rotation_matrix(yaw, pitch, roll, R);
    rotate_in_place(V, 8, R);
    for (int i = 0; i < 12; i++) {
        const double* p0 = V + 3*F[i][0]; const double* p1 = V + 3*F[i][1]; const double* p2 = V + 3*F[i][2];
        const double a[3] = {p1[0]-p0[0], p1[1]-p0[1], p1[2]-p0[2]}, b[3] = {p2[0]-p0[0], p2[1]-p0[1], p2[2]-p0[2]};
        double f[3] = {(a[1]*b[2] - a[2]*b[1]), (a[2]*b[0] - a[0]*b[2]), (a[0]*b[1] - a[1]*b[0])};
        const double norm = std::sqrt(f[0]*f[0] + f[1]*f[1] + f[2]*f[2]);
        for (int k = 0; k < 3; k++) { N[3*i+k] = f[k]/norm; T[3*i+k] = F[i][k]; }
    }
    return RTS_OK;
}

// "sphere": icosahedron subdivided n times, midpoints pushed to the unit sphere
// (ray_tracer.cpp:85-101, 300-426).  20*4^n triangles, 10*4^n + 2 vertices.  Vertex order =
// lexicographic order of the exact (x, y, z) doubles BEFORE rotation (std::set ordering,
// :397-403); triangle order = lexicographic order of the remapped index triples (:417-418).
extern "C" int rts_sphere_mesh(uint32_t n, float radius, float yaw, float pitch, float roll, double* vertices, uint32_t* n_vertices,
                               uint32_t* triangles, uint32_t* n_triangles, double* normals)
{
    if (n > 9) { rts_set_error("rts_sphere_mesh: more than 9 subdivisions (5.2M triangles)"); return RTS_ERR_UNSUPPORTED; }
    uint64_t nt = 20; for (uint32_t g = 0; g < n; g++) nt *= 4;
    const uint64_t nv = nt / 2 + 2;
    if (n_vertices) *n_vertices = (uint32_t)nv;
    if (n_triangles) *n_triangles = (uint32_t)nt;
    if (!vertices && !triangles && !normals) return RTS_OK;
    if (!vertices || !triangles || !normals) { rts_set_error("rts_sphere_mesh: pass all of vertices/triangles/normals or none"); return RTS_ERR_INVALID; }

    const double t = (1 + std::sqrt(5)) / 2;
    std::vector<double> v = {-1, t, 0,  1, t, 0,  -1, -t, 0,  1, -t, 0,  0, -1, t,  0, 1, t,  0, -1, -t,  0, 1, -t,  t, 0, -1,  t, 0, 1,  -t, 0, -1,  -t, 0, 1};
    for (int i = 0; i < 12; i++) {
        const double norm = std::sqrt(v[3*i]*v[3*i] + v[3*i+1]*v[3*i+1] + v[3*i+2]*v[3*i+2]);
        v[3*i] = v[3*i]/norm; v[3*i+1] = v[3*i+1]/norm; v[3*i+2] = v[3*i+2]/norm;
    }
    std::vector<uint32_t> f = {0,11,5, 0,5,1, 0,1,7, 0,7,10, 0,10,11, 1,5,9, 5,11,4, 11,10,2, 10,7,6, 7,1,8,
                               3,9,4, 3,4,2, 3,2,6, 3,6,8, 3,8,9, 4,9,5, 2,4,11, 6,2,10, 8,6,7, 9,8,1};
    auto midpoint = [&](uint32_t a, uint32_t b) -> uint32_t {
        double pm[3] = {(v[3*a] + v[3*b])/2, (v[3*a+1] + v[3*b+1])/2, (v[3*a+2] + v[3*b+2])/2};
        const double norm = std::sqrt(pm[0]*pm[0] + pm[1]*pm[1] + pm[2]*pm[2]);
        const uint32_t id = (uint32_t)(v.size()/3);
        v.push_back(pm[0]/norm); v.push_back(pm[1]/norm); v.push_back(pm[2]/norm);
        return id;
    };
    for (uint32_t gen = 0; gen < n; gen++) {
        std::vector<uint32_t> f2(f.size()*4);
        v.reserve(v.size() + 3*f.size());
        for (size_t i = 0; i < f.size()/3; i++) {
            const uint32_t t0 = f[3*i], t1 = f[3*i+1], t2 = f[3*i+2];
            const uint32_t a = midpoint(t0, t1), b = midpoint(t1, t2), c = midpoint(t2, t0);
            const uint32_t nf[12] = {t0, a, c,  t1, b, a,  t2, c, b,  a, b, c};
            memcpy(&f2[12*i], nf, sizeof(nf));
        }
        f.swap(f2);
    }
    // unique vertices in lexicographic order; equal keys keep the first inserted (std::set semantics)
    const size_t nall = v.size()/3;
    std::vector<uint32_t> order(nall); std::iota(order.begin(), order.end(), 0u);
    auto less = [&](uint32_t a, uint32_t b) {
        for (int k = 0; k < 3; k++) { if (v[3*a+k] < v[3*b+k]) return true; if (v[3*b+k] < v[3*a+k]) return false; }
        return false;
    };
    std::stable_sort(order.begin(), order.end(), less);
    std::vector<uint32_t> ix(nall); std::vector<double> uv; uv.reserve(3*nv);
    uint32_t nu = 0;
    for (size_t i = 0; i < nall; i++) {
        if (i == 0 || less(order[i-1], order[i])) { uv.push_back(v[3*order[i]]); uv.push_back(v[3*order[i]+1]); uv.push_back(v[3*order[i]+2]); nu++; }
        ix[order[i]] = nu - 1;
    }
    if (nu != nv) { rts_set_error("rts_sphere_mesh: internal vertex count mismatch (%u vs %llu)", nu, (unsigned long long)nv); return RTS_ERR_INVALID; }
    double R[3][3]; rotation_matrix(yaw, pitch, roll, R);
    rotate_in_place(uv.data(), nu, R);
    memcpy(normals, uv.data(), sizeof(double)*3*nu);                    // unit vertices double as the vertex normals (:409)
    // faces: remap, sort lexicographically, drop duplicates
    const size_t nf = f.size()/3;
    struct Tri { uint32_t a, b, c; };
    std::vector<Tri> tris(nf);
    for (size_t i = 0; i < nf; i++) tris[i] = Tri{ix[f[3*i]], ix[f[3*i+1]], ix[f[3*i+2]]};
    auto tless = [](const Tri& x, const Tri& y) { if (x.a != y.a) return x.a < y.a; if (x.b != y.b) return x.b < y.b; return x.c < y.c; };
    std::sort(tris.begin(), tris.end(), tless);
    tris.erase(std::unique(tris.begin(), tris.end(), [](const Tri& x, const Tri& y) { return x.a == y.a && x.b == y.b && x.c == y.c; }), tris.end());
    if (tris.size() != nt) { rts_set_error("rts_sphere_mesh: internal triangle count mismatch"); return RTS_ERR_INVALID; }
    for (size_t i = 0; i < tris.size(); i++) { triangles[3*i] = tris[i].a; triangles[3*i+1] = tris[i].b; triangles[3*i+2] = tris[i].c; }
    for (uint32_t i = 0; i < nu; i++) for (int k = 0; k < 3; k++) vertices[3*(size_t)i+k] = uv[3*(size_t)i+k] * radius;   // :421-425
    return RTS_OK;
}

// "file": one triangle per line, "x y z, x y z, x y z," for the vertices and the same layout
// for the per-vertex normals in a second file (ray_tracer.cpp:429-504).  Vertices are NOT
// shared: triangle i uses vertices 3i, 3i+1, 3i+2.  Call with NULL outputs to get the count.
extern "C" int rts_file_mesh(const char* v_file, const char* n_file, float yaw, float pitch, float roll, double* vertices,
                             uint32_t* triangles, double* normals, uint32_t* n_triangles)
{
    if (!v_file || !n_file || !n_triangles) { rts_set_error("rts_file_mesh: null argument"); return RTS_ERR_INVALID; }
    FILE* fp = fopen(v_file, "r");
    if (!fp) { rts_set_error("rts_file_mesh: cannot open vertex coordinates file %s", v_file); return RTS_ERR_IO; }   // the reference exit()s, :455-458
    uint32_t lines = 0; { char buf[65536]; size_t got; while ((got = fread(buf, 1, sizeof(buf), fp)) > 0) for (size_t i = 0; i < got; i++) if (buf[i] == '\n') lines++; }
    if (!vertices && !triangles && !normals) { fclose(fp); *n_triangles = lines; return RTS_OK; }
    if (!vertices || !triangles || !normals) { fclose(fp); rts_set_error("rts_file_mesh: pass all outputs or none"); return RTS_ERR_INVALID; }
    if (*n_triangles < lines) { fclose(fp); *n_triangles = lines; rts_set_error("rts_file_mesh: capacity too small"); return RTS_ERR_CAPACITY; }
    *n_triangles = lines;
    rewind(fp);
    // The reference's loop tests fscanf() == EOF only (:459-476): a line that yields fewer than its nine numbers -- a missing
    // comma, a word where a number belongs -- leaves the remaining coordinates at whatever the vector held (zeros) and the
    // stream wherever the match failed, i.e. a silently wrong mesh (NaN normals downstream).  A C-ABI that promises status codes
    // reports it: RTS_ERR_IO naming the file and the 1-based triangle (line) that did not parse.  Every file the reference reads
    // correctly -- nine numbers per line, with or without the final comma -- reads exactly as there.
    // (line by line: a line with too few numbers must not borrow the next line's -- fscanf's white space crosses line ends)
    auto read9 = [&](FILE* f, double* dst, uint32_t* bad_line, int* got) -> bool {
        char* line = nullptr; size_t cap = 0; bool ok = true;
        for (uint32_t i = 0; i < lines && ok; i++) {
            double* p = dst + 9*(size_t)i;
            int n = EOF, used = -1;
            if (getline(&line, &cap, f) >= 0) {
                n = sscanf(line, "%lf %lf %lf, %lf %lf %lf, %lf %lf %lf%n", p, p+1, p+2, p+3, p+4, p+5, p+6, p+7, p+8, &used);
                if (n == EOF) n = 0;                                   // (an empty line)
                // behind the ninth number: the closing comma -- or nothing: the reference's fscanf has matched its nine conversions by then and
                // returns 9 with or without it (ray_tracer.cpp:461), and goes on with the next line either way -- and white space only
                if (n == 9) { const char* q = line + used; while (isspace((unsigned char)*q)) q++; if (*q == ',') q++; for (; *q; q++) if (!isspace((unsigned char)*q)) { n = 10; break; } }
            }
            if (n != 9) { *bad_line = i + 1; *got = n; ok = false; }
        }
        free(line);
        return ok;
    };
    memset(vertices, 0, sizeof(double)*9*(size_t)lines); memset(normals, 0, sizeof(double)*9*(size_t)lines);
    uint32_t bad = 0; int got = 0;
    bool ok = read9(fp, vertices, &bad, &got); fclose(fp);
    if (!ok) { rts_set_error("rts_file_mesh: vertex file %s: triangle %u of %u %s (expected \"x y z, x y z, x y z,\": 9 numbers and nothing else, matched %d)", v_file, bad, lines, got == EOF ? "is missing (file ends early)" : "does not parse", got == EOF ? 0 : got); return RTS_ERR_IO; }
    fp = fopen(n_file, "r");
    if (!fp) { rts_set_error("rts_file_mesh: cannot open vertex normals file %s", n_file); return RTS_ERR_IO; }      // :480-483
    ok = read9(fp, normals, &bad, &got); fclose(fp);
    if (!ok) { rts_set_error("rts_file_mesh: normals file %s: triangle %u of %u %s (expected \"x y z, x y z, x y z,\": 9 numbers and nothing else, matched %d)", n_file, bad, lines, got == EOF ? "is missing (file ends early)" : "does not parse", got == EOF ? 0 : got); return RTS_ERR_IO; }
    double R[3][3]; rotation_matrix(yaw, pitch, roll, R);
    rotate_in_place(vertices, 3*(size_t)lines, R);
    rotate_in_place(normals, 3*(size_t)lines, R);
    for (uint32_t i = 0; i < lines; i++) { triangles[3*(size_t)i] = 3*i; triangles[3*(size_t)i+1] = 3*i + 1; triangles[3*(size_t)i+2] = 3*i + 2; }
    return RTS_OK;
}

// Receiver capture sphere from the receiver's position, boresight and sphere parameters
// (ray_tracer.cpp:894-918).  The reference evaluates this on the host with FLOAT trig
// (cosf/sinf/atan2f applied to doubles); so does this.
extern "C" int rts_rx_sphere(const double* repos, double azimuth, double elevation, double radius, double theta_span, double phi_span,
                             RtsReceiverSphere* out)
{
    if (!repos || !out) { rts_set_error("rts_rx_sphere: null argument"); return RTS_ERR_INVALID; }
    double h_Rx_azimuth = azimuth, h_Rx_elevation = elevation;
    const double cx = repos[0] + (radius * cosf(h_Rx_elevation) * cosf(h_Rx_azimuth));
    const double cy = repos[1] + (radius * cosf(h_Rx_elevation) * sinf(h_Rx_azimuth));
    const double cz = repos[2] + (radius * sinf(h_Rx_elevation));
    h_Rx_azimuth = atan2f((repos[1] - cy), (repos[0] - cx));
    h_Rx_elevation = atan2f((repos[2] - cz), std::sqrt((repos[0] - cx)*(repos[0] - cx) + (repos[1] - cy)*(repos[1] - cy)));
    out->centre[0] = cx; out->centre[1] = cy; out->centre[2] = cz; out->radius = radius;
    out->min_theta = h_Rx_azimuth - theta_span/2; out->max_theta = h_Rx_azimuth + theta_span/2;
    out->min_phi = h_Rx_elevation - phi_span/2; out->max_phi = h_Rx_elevation + phi_span/2;
    return RTS_OK;
}
