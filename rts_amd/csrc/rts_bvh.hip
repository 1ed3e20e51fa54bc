// rts_bvh.hip -- per-pulse placement of the scene on the device.
//
// The hierarchy itself is static (rts_sah.cpp: one target-space BVH4 per mesh, built when the scene is set); what a
// pulse changes is the placement  world = R * local + position  of every target (ray_tracer.cpp:993-1014).  Per pulse:
//   place  : world vertices/normals = R * local (+ position), the reference's arithmetic          (ray_tracer.cpp:120-137)
//   leaves : gather the three f64 world vertices of each primitive into the static leaf order
// The reference instead refills its vertex buffers and has OptiX rebuild the "Bvh" acceleration each pulse
// (ray_tracer.cpp:1126-1130); the per-target inverse placement and world bounds the traversal needs are launch
// constants computed on the host (rts_api.hip: fill_target_placement).
#include <cstring>
#include <hip/hip_runtime.h>
#include "rts_internal.h"

// --------------------------------------------------------------------------- placement
// verts_rot[v][i] = sum_k R[i][k] * vert[v][k], accumulated from 0 in k order
// (matrix_multiply, ray_tracer.cpp:120-137,166), then += position (:1010-1014).
__device__ __forceinline__ void place_one(const double* __restrict__ local, double* __restrict__ world, const uint32_t* __restrict__ targ_of,
                                          const RtsTargetMotion* __restrict__ motion, uint32_t i, bool add_position)
{
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
// vertices (rotated, then moved) and normals (rotated) of all targets in ONE launch: thread i < n_verts places vertex i, the rest normal i - n_verts
__global__ void k_place(const double* __restrict__ v_local, double* __restrict__ v_world, const uint32_t* __restrict__ v_targ, uint32_t n_verts,
                        const double* __restrict__ n_local, double* __restrict__ n_world, const uint32_t* __restrict__ n_targ, uint32_t n_normals,
                        const RtsTargetMotion* __restrict__ motion)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_verts) place_one(v_local, v_world, v_targ, motion, i, true);
    else if (i - n_verts < n_normals) place_one(n_local, n_world, n_targ, motion, i - n_verts, false);
}

// --------------------------------------------------------------------------- primary-ray mask (see RtsMaskFrame)
// One thread per LEAF, in leaf order: neighbouring leaves are neighbours in space, so the 256 triangles of a block cover a
// compact patch of the bitmap.  The block rasterises into an LDS tile over the bounding rectangle of its triangles' rectangles
// and then ORs the tile's non-zero words into the mask -- a few dozen global atomics per block instead of a few per triangle
// (100 k triangles marking the same ~5 k words: same-line atomics serialise in L2, the per-triangle version took 80-130 us).
// A block whose patch does not fit the tile falls back to per-triangle atomics.
#define RTS_MASK_TILE_WORDS 2048
struct RtsMaskLds { uint32_t tile[RTS_MASK_TILE_WORDS]; int box[4]; };          // box: min iu0, min iv0, max iu1, max iv1 of the block
// (called by every thread of the block: `have` = the thread holds a triangle, p[k] = its world-space vertices)
__device__ __forceinline__ void rts_mask_mark(RtsMaskLds& S, const bool have, const double (&p)[9], const double ox, const double oy, const double oz, const RtsMaskFrame& f, uint32_t* __restrict__ mask)
{
    uint32_t* flag = mask + (size_t)f.n * f.n / 32u;
    if (threadIdx.x == 0) { S.box[0] = 0x7fffffff; S.box[1] = 0x7fffffff; S.box[2] = -1; S.box[3] = -1; }
    __syncthreads();
    int iu0 = 0, iu1 = -1, iv0 = 0, iv1 = -1;                       // empty rectangle: nothing to mark
    if (have) {
        double u[3], v[3]; bool finite = true, front = true;
        for (int k = 0; k < 3; k++) {
            const double px = p[3*k] - ox, py = p[3*k + 1] - oy, pz = p[3*k + 2] - oz;
            finite = finite && isfinite(px) && isfinite(py) && isfinite(pz);
            const double w = px * (double)f.bx + py * (double)f.by + pz * (double)f.bz;
            front = front && (w > 0.0) && (w * w > 1.0e-6 * (px*px + py*py + pz*pz));   // well in front of the transmitter (within ~89.94 degrees of the boresight)
            u[k] = (px * (double)f.ux + py * (double)f.uy + pz * (double)f.uz) / w;
            v[k] = (px * (double)f.vx + py * (double)f.vy + pz * (double)f.vz) / w;
        }
        if (finite && !front) atomicExch(flag, 1u);                 // its projection is not a triangle: no mask for this pulse
        if (finite && front) {
            const double fu0 = (fmin(fmin(u[0], u[1]), u[2]) - (double)f.u0) * (double)f.inv_du, fu1 = (fmax(fmax(u[0], u[1]), u[2]) - (double)f.u0) * (double)f.inv_du;
            const double fv0 = (fmin(fmin(v[0], v[1]), v[2]) - (double)f.v0) * (double)f.inv_dv, fv1 = (fmax(fmax(v[0], v[1]), v[2]) - (double)f.v0) * (double)f.inv_dv;
            const double nn = (double)f.n;
            if (!(fu1 < -1.0 || fv1 < -1.0 || fu0 > nn || fv0 > nn)) {                    // else: outside the beam
                iu0 = (int)fmax(floor(fu0) - 1.0, 0.0); iu1 = (int)fmin(floor(fu1) + 1.0, nn - 1.0);
                iv0 = (int)fmax(floor(fv0) - 1.0, 0.0); iv1 = (int)fmin(floor(fv1) + 1.0, nn - 1.0);
                if ((long long)(iu1 - iu0 + 1) * (long long)(iv1 - iv0 + 1) > RTS_MASK_MAX_CELLS) { atomicExch(flag, 1u); iu1 = iu0 - 1; }
            }
        }
    }
    const bool any = iu1 >= iu0 && iv1 >= iv0;
    if (any) { atomicMin(&S.box[0], iu0); atomicMin(&S.box[1], iv0); atomicMax(&S.box[2], iu1); atomicMax(&S.box[3], iv1); }
    __syncthreads();
    if (S.box[2] < 0) return;                                       // (uniform) nothing to mark in this block
    const int W0 = S.box[0] >> 5, V0 = S.box[1], TW = (S.box[2] >> 5) - W0 + 1, TH = S.box[3] - V0 + 1;
    const bool tiled = (long long)TW * TH <= RTS_MASK_TILE_WORDS;   // (uniform)
    if (tiled) { for (int w = threadIdx.x; w < TW * TH; w += blockDim.x) S.tile[w] = 0u; __syncthreads(); }
    if (any) {
        for (int iv = iv0; iv <= iv1; iv++) {
            for (int w0 = iu0 >> 5; w0 <= (iu1 >> 5); w0++) {       // one atomic per touched word of the row
                const int lo = max(iu0, w0 << 5) & 31, hi = min(iu1, (w0 << 5) + 31) & 31;
                const uint32_t bits = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                if (tiled) atomicOr(&S.tile[(iv - V0) * TW + (w0 - W0)], bits);
                else atomicOr(&mask[((size_t)iv * f.n >> 5) + (size_t)w0], bits);
            }
        }
    }
    if (!tiled) return;
    __syncthreads();
    for (int w = threadIdx.x; w < TW * TH; w += blockDim.x) {
        const uint32_t bits = S.tile[w];
        if (bits) atomicOr(&mask[((size_t)(V0 + w / TW) * f.n >> 5) + (size_t)(W0 + w % TW)], bits);
    }
}

// Leaf record i = primitive leaf_prim[i] with its world-space vertices pre-gathered (the reference gathers through
// dbuf_triangles -> dbuf_triVertices per test, triangle_mesh.cu:147-154).  LEAVES && MASK: one pass over the placed triangles
// for both (a pulse that moves a target needs both; the vertices are gathered once).  MASK alone: the targets stand still,
// the beam moved.
// PLACE (r04): the kernel places what it gathers -- the three vertices of its primitive from the LOCAL arrays through the pulse's
// placement (place_one's arithmetic on registers: the bits of k_place's world array, which it then does not need) -- and its
// blocks behind the primitives' place the normals: ONE launch for the whole per-pulse scene update instead of k_place + k_leaves.
__device__ __forceinline__ void place_vertex(const double* __restrict__ local, const RtsTargetMotion& m, uint32_t i, double* __restrict__ out3)
{
    double x = local[3*(size_t)i], y = local[3*(size_t)i+1], z = local[3*(size_t)i+2];
    if (m.has_rotation) {
        double rx = 0.0, ry = 0.0, rz = 0.0;
        rx += m.rotation[0] * x; rx += m.rotation[1] * y; rx += m.rotation[2] * z;
        ry += m.rotation[3] * x; ry += m.rotation[4] * y; ry += m.rotation[5] * z;
        rz += m.rotation[6] * x; rz += m.rotation[7] * y; rz += m.rotation[8] * z;
        x = rx; y = ry; z = rz;
    }
    x += m.position[0]; y += m.position[1]; z += m.position[2];
    out3[0] = x; out3[1] = y; out3[2] = z;
}
template <bool LEAVES, bool MASK, bool PLACE = false>
__global__ void __launch_bounds__(256) k_leaves(const uint32_t* __restrict__ leaf_prim, const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts,
                                                const uint32_t* __restrict__ prim_targ, RtsLeafTri* __restrict__ leaves, uint32_t n,
                                                double ox, double oy, double oz, RtsMaskFrame f, uint32_t* __restrict__ mask,
                                                const RtsTargetMotion* __restrict__ motion = nullptr, uint32_t prim_blocks = 0, const double* __restrict__ n_local = nullptr,
                                                double* __restrict__ n_world = nullptr, const uint32_t* __restrict__ n_targ = nullptr, uint32_t n_normals = 0)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_raw[MASK ? sizeof(RtsMaskLds) / 4 : 1];
    if (PLACE && blockIdx.x >= prim_blocks) {                            // (uniform per block) the blocks behind the primitives': normals
        const uint32_t q = (blockIdx.x - prim_blocks) * blockDim.x + threadIdx.x;
        if (q < n_normals) place_one(n_local, n_world, n_targ, motion, q, false);
        return;
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double p[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i < n) {
        const uint32_t g = leaf_prim ? leaf_prim[i] : i;                 // (one record per primitive: leaf_prim == nullptr)
        const uint32_t a = tri_vidx[3*(size_t)g], b = tri_vidx[3*(size_t)g+1], c = tri_vidx[3*(size_t)g+2];
        if (PLACE) {                                                     // verts = the LOCAL vertices
            const RtsTargetMotion m = motion[prim_targ[g]];
            place_vertex(verts, m, a, p); place_vertex(verts, m, b, p + 3); place_vertex(verts, m, c, p + 6);
        } else {
        p[0] = verts[3*(size_t)a]; p[1] = verts[3*(size_t)a+1]; p[2] = verts[3*(size_t)a+2];
        p[3] = verts[3*(size_t)b]; p[4] = verts[3*(size_t)b+1]; p[5] = verts[3*(size_t)b+2];
        p[6] = verts[3*(size_t)c]; p[7] = verts[3*(size_t)c+1]; p[8] = verts[3*(size_t)c+2];
        }
        if (LEAVES) {
            RtsLeafTri L;
            L.p0x = p[0]; L.p0y = p[1]; L.p0z = p[2]; L.p1x = p[3]; L.p1y = p[4]; L.p1z = p[5]; L.p2x = p[6]; L.p2y = p[7]; L.p2z = p[8];
            L.prim = g; L.targ = prim_targ[g];
            leaves[i] = L;
        }
    }
    if (MASK) rts_mask_mark(*reinterpret_cast<RtsMaskLds*>(s_raw), i < n, p, ox, oy, oz, f, mask);
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

// The placement of a pulse (when a target moved: `place`) and its primary-ray mask (when the pulse has one: lc.mask.n), with
// one pass over the leaves for both where both are due.
int rts_scene_place(RtsContext* c, const RtsLaunchConsts& lc, bool place, uint32_t* pmask)
{
    hipStream_t st = c->stream;
    const RtsMaskFrame& f = lc.mask;
    const bool mask = f.n != 0;
    const RtsScene* sc = c->scene;
    // (the mask buffer -- behind the handle's zero block -- has been cleared with it: rts_trace_pulse_begin)
    // ONE launch for a pulse that moves a target (RtsContext::place_fused, RTS_PLACE_FUSED=0: k_place + k_leaves): the leaf kernel
    // places its primitive's vertices itself, its trailing blocks the normals.  (A later pulse that only re-marks the mask -- the
    // targets stand still, the beam moved -- gathers from the world vertices: kept up to date by k_place in that mode only.)
    if (place && c->place_fused && sc->n_prims) {
        const unsigned gp = blocks_for(sc->n_prims, 256), gn = blocks_for(sc->n_normals, 256);
        if (mask) k_leaves<true, true, true><<<gp + gn, 256, 0, st>>>(nullptr, sc->d_tri_vidx.p, sc->d_verts_local.p, sc->d_prim_targ.p, c->d_leaves.p, sc->n_prims, lc.ox, lc.oy, lc.oz, f, pmask,
                                                                     c->p_motion, gp, sc->d_normals_local.p, c->d_normals_world.p, sc->d_norm_targ.p, sc->n_normals);
        else k_leaves<true, false, true><<<gp + gn, 256, 0, st>>>(nullptr, sc->d_tri_vidx.p, sc->d_verts_local.p, sc->d_prim_targ.p, c->d_leaves.p, sc->n_prims, lc.ox, lc.oy, lc.oz, f, nullptr,
                                                                  c->p_motion, gp, sc->d_normals_local.p, c->d_normals_world.p, sc->d_norm_targ.p, sc->n_normals);
        c->verts_world_valid = false;
        RTS_HIP(hipGetLastError());
        return RTS_OK;
    }
    if (!place && mask && !c->verts_world_valid && sc->n_verts) {        // the world vertices the mask-only pass gathers from: placed now, once
        k_place<<<blocks_for((size_t)sc->n_verts, 256), 256, 0, st>>>(sc->d_verts_local.p, c->d_verts_world.p, sc->d_vert_targ.p, sc->n_verts, sc->d_normals_local.p, c->d_normals_world.p, sc->d_norm_targ.p, 0u, c->p_motion);
        c->verts_world_valid = true;
    }
    if (place) c->verts_world_valid = true;
    if (place) {
        if (sc->n_verts + sc->n_normals) k_place<<<blocks_for((size_t)sc->n_verts + sc->n_normals, 256), 256, 0, st>>>(sc->d_verts_local.p, c->d_verts_world.p, sc->d_vert_targ.p, sc->n_verts,
                                                                                                                         sc->d_normals_local.p, c->d_normals_world.p, sc->d_norm_targ.p, sc->n_normals, c->p_motion);
    }
    if (sc->n_prims) {                                               // one leaf record per primitive (rts_api.hip: k_children_to_prims)
        const unsigned g = blocks_for(sc->n_prims, 256);
        if (place && mask) k_leaves<true, true><<<g, 256, 0, st>>>(nullptr, sc->d_tri_vidx.p, c->d_verts_world.p, sc->d_prim_targ.p, c->d_leaves.p, sc->n_prims, lc.ox, lc.oy, lc.oz, f, pmask);
        else if (place) k_leaves<true, false><<<g, 256, 0, st>>>(nullptr, sc->d_tri_vidx.p, c->d_verts_world.p, sc->d_prim_targ.p, c->d_leaves.p, sc->n_prims, lc.ox, lc.oy, lc.oz, f, nullptr);
        else if (mask) k_leaves<false, true><<<g, 256, 0, st>>>(nullptr, sc->d_tri_vidx.p, c->d_verts_world.p, sc->d_prim_targ.p, nullptr, sc->n_prims, lc.ox, lc.oy, lc.oz, f, pmask);
    }
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}
