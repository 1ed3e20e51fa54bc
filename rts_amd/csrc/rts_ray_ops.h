// rts_ray_ops.h -- the f64 triangle test of the trace kernel: a literal restatement of the cited reference lines
// (operand order matters: results are compared bit-for-bit with the CPU oracle).
#pragma once
#include "rts_internal.h"

struct TriHit { double t, beta, gamma; dvec3 n; bool ok; };

// intersect_triangle_doubles, triangle_mesh.cu:121-137 (tmin/tmax are the f32 ray constants)
__device__ __forceinline__ TriHit tri_test(const RtsLeafTri& L, dvec3 o, dvec3 d, float tmin, float tmax)
{
    const dvec3 p0 = mk3(L.p0x, L.p0y, L.p0z), p1 = mk3(L.p1x, L.p1y, L.p1z), p2 = mk3(L.p2x, L.p2y, L.p2z);
    const dvec3 e0 = sub3(p1, p0);
    const dvec3 e1 = sub3(p0, p2);
    TriHit h;
    h.n = cross3(e1, e0);
    const dvec3 e2 = scale3(1 / dot3(h.n, d), sub3(p0, o));
    const dvec3 i = cross3(d, e2);
    h.beta = dot3(i, e1);
    h.gamma = dot3(i, e0);
    h.t = dot3(h.n, e2);
    h.ok = (h.t < (double)tmax) & (h.t > (double)tmin) & (h.beta >= 0.0) & (h.gamma >= 0.0) & (h.beta + h.gamma <= 1);
    return h;
}
