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

// Leaf record i = primitive leaf_prim[i] with its world-space vertices pre-gathered (the reference gathers through
// dbuf_triangles -> dbuf_triVertices per test, triangle_mesh.cu:147-154).
__global__ void k_leaves(const uint32_t* __restrict__ leaf_prim, const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts,
                         const uint32_t* __restrict__ prim_targ, RtsLeafTri* __restrict__ leaves, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t g = leaf_prim[i];
    uint32_t a = tri_vidx[3*g], b = tri_vidx[3*g+1], c = tri_vidx[3*g+2];
    RtsLeafTri L;
    L.p0x = verts[3*a]; L.p0y = verts[3*a+1]; L.p0z = verts[3*a+2];
    L.p1x = verts[3*b]; L.p1y = verts[3*b+1]; L.p1z = verts[3*b+2];
    L.p2x = verts[3*c]; L.p2y = verts[3*c+1]; L.p2z = verts[3*c+2];
    L.prim = g; L.targ = prim_targ[g];
    leaves[i] = L;
}

// --------------------------------------------------------------------------- primary-ray mask (see RtsMaskFrame)
// One thread per LEAF, in leaf order: neighbouring leaves are neighbours in space, so the 256 triangles of a block cover a
// compact patch of the bitmap.  The block rasterises into an LDS tile over the bounding rectangle of its triangles' rectangles
// and then ORs the tile's non-zero words into the mask -- a few dozen global atomics per block instead of a few per triangle
// (100 k triangles marking the same ~5 k words: same-line atomics serialise in L2, the per-triangle version took 80-130 us).
// A block whose patch does not fit the tile falls back to per-triangle atomics.
#define RTS_MASK_TILE_WORDS 2048
__global__ void __launch_bounds__(256) k_primary_mask(const uint32_t* __restrict__ leaf_prim, const uint32_t* __restrict__ tri_vidx, const double* __restrict__ verts, uint32_t n_leaves,
                                                      double ox, double oy, double oz, RtsMaskFrame f, uint32_t* __restrict__ mask)
{
    __shared__ uint32_t s_tile[RTS_MASK_TILE_WORDS];
    __shared__ int s_box[4];                                        // min iu0, min iv0, max iu1, max iv1 of the block
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t* flag = mask + (size_t)f.n * f.n / 32u;
    if (threadIdx.x == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -1; s_box[3] = -1; }
    __syncthreads();
    int iu0 = 0, iu1 = -1, iv0 = 0, iv1 = -1;                       // empty rectangle: nothing to mark
    if (i < n_leaves) {
        const uint32_t prim = leaf_prim[i];
        double u[3], v[3]; bool finite = true, front = true;
        for (int k = 0; k < 3; k++) {
            const uint32_t a = tri_vidx[3*(size_t)prim + k];
            const double px = verts[3*(size_t)a] - ox, py = verts[3*(size_t)a + 1] - oy, pz = verts[3*(size_t)a + 2] - oz;
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
    if (any) { atomicMin(&s_box[0], iu0); atomicMin(&s_box[1], iv0); atomicMax(&s_box[2], iu1); atomicMax(&s_box[3], iv1); }
    __syncthreads();
    if (s_box[2] < 0) return;                                       // (uniform) nothing to mark in this block
    const int W0 = s_box[0] >> 5, V0 = s_box[1], TW = (s_box[2] >> 5) - W0 + 1, TH = s_box[3] - V0 + 1;
    const bool tiled = (long long)TW * TH <= RTS_MASK_TILE_WORDS;   // (uniform)
    if (tiled) { for (int w = threadIdx.x; w < TW * TH; w += blockDim.x) s_tile[w] = 0u; __syncthreads(); }
    if (any) {
        for (int iv = iv0; iv <= iv1; iv++) {
            for (int w0 = iu0 >> 5; w0 <= (iu1 >> 5); w0++) {       // one atomic per touched word of the row
                const int lo = max(iu0, w0 << 5) & 31, hi = min(iu1, (w0 << 5) + 31) & 31;
                const uint32_t bits = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                if (tiled) atomicOr(&s_tile[(iv - V0) * TW + (w0 - W0)], bits);
                else atomicOr(&mask[((size_t)iv * f.n >> 5) + (size_t)w0], bits);
            }
        }
    }
    if (!tiled) return;
    __syncthreads();
    for (int w = threadIdx.x; w < TW * TH; w += blockDim.x) {
        const uint32_t bits = s_tile[w];
        if (bits) atomicOr(&mask[((size_t)(V0 + w / TW) * f.n >> 5) + (size_t)(W0 + w % TW)], bits);
    }
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

int rts_primary_mask_build(RtsContext* c, const RtsLaunchConsts& lc)
{
    const RtsMaskFrame& f = lc.mask;
    if (f.n == 0) return RTS_OK;
    hipStream_t st = c->stream;
    const size_t words = (size_t)f.n * f.n / 32u + 1u;
    RTS_HIP(c->d_pmask.reserve(words));
    RTS_HIP(hipMemsetAsync(c->d_pmask.p, 0, sizeof(uint32_t) * words, st));
    if (c->scene->n_leaves) k_primary_mask<<<blocks_for(c->scene->n_leaves, 256), 256, 0, st>>>(c->scene->d_leaf_prim.p, c->scene->d_tri_vidx.p, c->d_verts_world.p, c->scene->n_leaves, lc.ox, lc.oy, lc.oz, f, c->d_pmask.p);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

int rts_scene_place(RtsContext* c)
{
    hipStream_t st = c->stream;
    if (c->scene->n_verts) k_place<<<blocks_for(c->scene->n_verts, 256), 256, 0, st>>>(c->scene->d_verts_local.p, c->d_verts_world.p, c->scene->d_vert_targ.p, c->p_motion, c->scene->n_verts, 1);
    if (c->scene->n_normals) k_place<<<blocks_for(c->scene->n_normals, 256), 256, 0, st>>>(c->scene->d_normals_local.p, c->d_normals_world.p, c->scene->d_norm_targ.p, c->p_motion, c->scene->n_normals, 0);
    if (c->scene->n_leaves) k_leaves<<<blocks_for(c->scene->n_leaves, 256), 256, 0, st>>>(c->scene->d_leaf_prim.p, c->scene->d_tri_vidx.p, c->d_verts_world.p, c->scene->d_prim_targ.p, c->d_leaves.p, c->scene->n_leaves);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}
