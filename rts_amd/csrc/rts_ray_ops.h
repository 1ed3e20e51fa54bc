// rts_ray_ops.h -- per-ray operations shared by the trace kernels: the f64 triangle test, the miss program and
// the reflection shading step, each a literal restatement of the cited reference lines (operand order matters:
// results are compared bit-for-bit with the CPU oracle).
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


// payload of one ray chain (the fields of PerRayData that the device path carries, ray_tracer.h:13-28)
struct RtsRay {
    dvec3 dir, prev, first;
    double rayLength, power, doppler;
    uint32_t reflDepth;
    int received;
    bool end;
    uint64_t path_lo, path_hi;
};

// miss program, ray_tracer.cu:260-478 (maxRefr == 0 form: "direct" <=> reflDepth == 0)
__device__ __forceinline__ void rts_miss_program(RtsRay& r, const RtsRxDev* __restrict__ rxs, uint32_t n_rx, dvec3 origin)
{
    const dvec3 dir = r.dir, prev = r.prev;
    if (r.end == false) {
        for (uint32_t Rx_i = 0; Rx_i < n_rx; Rx_i++) {
            const RtsRxDev rx = rxs[Rx_i];
            double t[2] = {0, 0};
            const double A = (dir.x)*(dir.x) + (dir.y)*(dir.y) + (dir.z)*(dir.z);
            const double B = 2*(((prev.x - rx.cx)*dir.x) + ((prev.y - rx.cy)*dir.y) + ((prev.z - rx.cz)*dir.z));
            const double C = prev.x*prev.x + prev.y*prev.y + prev.z*prev.z + (rx.cx*rx.cx) + (rx.cy*rx.cy) + (rx.cz*rx.cz) -
                             2*((rx.cx*prev.x) + (rx.cy*prev.y) + (rx.cz*prev.z)) - rx.radius*rx.radius;
            double discriminant = B*B - 4*A*C;
            if (discriminant > 0.f) {
                discriminant = sqrt(discriminant);
                t[0] = (-B - discriminant)/(2*A);
                t[1] = (-B + discriminant)/(2*A);
                unsigned int received_root = 2;
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    if ((t[i] >= 0) && ((r.rayLength + t[i]) > SCENE_EPS) && ((r.rayLength + t[i]) > SCENE_EPS_R)) {
                        const dvec3 ep = mk3(prev.x + t[i]*dir.x, prev.y + t[i]*dir.y, prev.z + t[i]*dir.z);
                        double theta = rts_atan2f((float)(ep.y - rx.cy), (float)(ep.x - rx.cx));      // atan2f(float, float) :326-329
                        double phi = rts_atan2f((float)(ep.z - rx.cz), (float)sqrt(((ep.y - rx.cy) * (ep.y - rx.cy)) + ((ep.x - rx.cx) * (ep.x - rx.cx))));
                        if ((phi < -RTS_PI/2)) { theta += RTS_PI; phi = -RTS_PI - phi; }
                        if ((phi > RTS_PI/2)) { theta += RTS_PI; phi = RTS_PI - phi; }
                        double maxTheta1 = rx.maxTheta, minTheta1 = rx.minTheta, maxTheta2 = maxTheta1, minTheta2 = minTheta1;
                        double maxPhi1 = rx.maxPhi, minPhi1 = rx.minPhi, maxPhi2 = maxPhi1, minPhi2 = minPhi1;
                        if ((minPhi1 < -RTS_PI/2)) { maxTheta2 += RTS_PI; minTheta2 += RTS_PI; maxPhi2 = -RTS_PI - minPhi1; minPhi2 = -RTS_PI/2; minPhi1 = -RTS_PI/2; }
                        if ((maxPhi1 > RTS_PI/2)) { maxTheta2 += RTS_PI; minTheta2 += RTS_PI; minPhi2 = RTS_PI - maxPhi1; maxPhi2 = RTS_PI/2; maxPhi1 = RTS_PI/2; }
                        if (((rts_angle_in_range(theta, minTheta1, maxTheta1)) && (rts_angle_in_range(phi, minPhi1, maxPhi1))) ||
                            ((rts_angle_in_range(theta, minTheta2, maxTheta2)) && (rts_angle_in_range(phi, minPhi2, maxPhi2)))) {
                            if (received_root == 2) received_root = i;
                            else if (t[received_root] > t[i]) received_root = i;
                        }
                    }
                }
                if (received_root < 2) {
                    r.end = true;                                                              // :396
                    const double tr = received_root == 0 ? t[0] : t[1];
                    const dvec3 ep = mk3(prev.x + tr*dir.x, prev.y + tr*dir.y, prev.z + tr*dir.z);
                    if (r.reflDepth == 0) {                                                    // direct transmission :410-417
                        const dvec3 RxRange = sub3(ep, origin);
                        if (len3(RxRange) >= SCENE_EPS) {
                            r.power = 1/(4*RTS_PI*4*RTS_PI*(magsq3(RxRange)));
                            r.doppler = 0;
                            r.rayLength += tr;
                            r.received = (int)Rx_i;
                        }
                    } else {                                                                   // :419-425
                        const dvec3 RxRange = sub3(ep, prev);
                        if (len3(RxRange) >= SCENE_EPS_R) {
                            r.power *= 1/((magsq3(RxRange))*4*RTS_PI*4*RTS_PI);
                            r.rayLength += tr;
                            r.received = (int)Rx_i;
                        }
                    }
                }
            }
        }
    }
    if (r.end == false && r.rayLength > 0) {                                                   // Earth sphere :438-476
        const double d_earthRadius = 6378136;
        const double A = (dir.x)*(dir.x) + (dir.y)*(dir.y) + (dir.z)*(dir.z);
        const double B = 2*(prev.x*dir.x + prev.y*dir.y + prev.z*dir.z);
        const double C = prev.x*prev.x + prev.y*prev.y + prev.z*prev.z - d_earthRadius*d_earthRadius;
        double discriminant = B*B - 4*A*C;
        if (discriminant > 0.f) {
            discriminant = sqrt(discriminant);
            const double t0 = (-B - discriminant)/(2*A), t1 = (-B + discriminant)/(2*A);
            if ((t0 >= 0) && (r.rayLength > 0)) { r.end = true; r.rayLength += t0; }
            if ((t1 >= 0) && (r.rayLength > 0)) { r.end = true; r.rayLength += t1; }
        }
    }
}

// closest_hit without refraction, normal_shader.cu:128-340 with d_maxRefrDepth == 0.  The caller has checked the gate
// (:134).  `primary` = this is the first segment of the launch index (f32 direction rule, ray_tracer.cu:208).
// Returns the reflected f32 direction (also stored as r.dir widened, :303).
__device__ __forceinline__ fvec3 rts_shade_reflect(RtsRay& r, const RtsTraceArgs& a, const RtsLeafTri& L, const RtsTargetDev& T, float hit_t,
                                                   float tmin, bool primary, dvec3 origin)
{
    {   // path column = reflDepth < D always holds here (:140-146)
        const uint64_t code = (uint64_t)(L.targ + 1);
        if (r.reflDepth < 8) r.path_lo |= code << (8 * r.reflDepth); else r.path_hi |= code << (8 * (r.reflDepth - 8));
    }
    const dvec3 dir = r.dir, prev = r.prev;
    const dvec3 hitPoint = mk3(prev.x + (double)hit_t*dir.x, prev.y + (double)hit_t*dir.y, prev.z + (double)hit_t*dir.z);   // :149-152
    r.rayLength += hit_t;                                              // :153
    if (r.reflDepth == 0) {                                            // :159-166
        r.first = hitPoint;
        const dvec3 TxRange = sub3(r.first, origin);
        if (len3(TxRange) >= SCENE_EPS) r.power = 1/((magsq3(TxRange))*4*RTS_PI);
        else r.end = true;
    } else {                                                           // :167-173
        const dvec3 TargRange = sub3(hitPoint, prev);
        if (len3(TargRange) >= SCENE_EPS_R) r.power *= 1/((magsq3(TargRange))*4*RTS_PI);
        else r.end = true;
    }
    // attribute normal (triangle_mesh.cu:169-194): recompute the accepted test, same bits
    const TriHit h = tri_test(L, prev, dir, tmin, RTS_DEFAULT_TMAX);
    r.prev = hitPoint;                                                 // :176
    dvec3 normal;
    if (a.smooth) {
        const uint32_t* ni = a.tri_nidx + 3*(size_t)L.prim;
        if (T.perface_normals) {
            const double* n = a.normals + 3*(size_t)ni[0];
            normal = mk3(n[0], n[1], n[2]);
        } else {
            const double* n0 = a.normals + 3*(size_t)ni[0]; const double* n1 = a.normals + 3*(size_t)ni[1]; const double* n2 = a.normals + 3*(size_t)ni[2];
            const double w = 1.0f - h.beta - h.gamma;
            normal = mk3(n1[0]*h.beta + n2[0]*h.gamma + n0[0]*w, n1[1]*h.beta + n2[1]*h.gamma + n0[1]*w, n1[2]*h.beta + n2[2]*h.gamma + n0[2]*w);
        }
        normal = unit3(normal);
    } else {
        normal = unit3(h.n);
    }
    const fvec3 dirf = primary ? unit3_to_f32(dir) : mk3f((float)dir.x, (float)dir.y, (float)dir.z);
    r.reflDepth++;                                                     // :286
    const fvec3 nd = reflect3f(dirf, unit3_to_f32(normal));            // :296
    r.power *= T.reflCoeff;                                            // :298
    const dvec3 k0 = unit3(dir);                                       // :302
    r.dir = widen3(nd);                                                // :303
    const dvec3 k1 = unit3(r.dir);                                     // :304
    r.doppler += dot3(mk3(T.vx, T.vy, T.vz), sub3(k1, k0));            // :314
    return nd;
}
