// rts_device_math.h -- arithmetic building blocks of the device path.
//
// Every function here is a fixed tree of IEEE-754 basic operations (+ - * / sqrt, compares)
// and is compiled with -ffp-contract=off, so the bits it produces are a function of its
// inputs only: the same on gfx950 and on any IEEE host.  That is what lets the results be
// compared bit-for-bit with a CPU restatement of the reference.  Expression ORDER follows
// the reference source (cited per function); do not "simplify".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTS_HD __host__ __device__ __forceinline__
#define RTS_D __device__ __forceinline__

#define RTS_PI 3.14159265358979323846   // M_PI
#define RTS_DEFAULT_TMAX 1e27f          // RT_DEFAULT_MAX of the OptiX SDK

struct dvec3 { double x, y, z; };
struct fvec3 { float x, y, z; };

RTS_HD dvec3 mk3(double x, double y, double z) { dvec3 o; o.x = x; o.y = y; o.z = z; return o; }
RTS_HD fvec3 mk3f(float x, float y, float z) { fvec3 o; o.x = x; o.y = y; o.z = z; return o; }
RTS_HD dvec3 add3(dvec3 a, dvec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTS_HD dvec3 sub3(dvec3 a, dvec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTS_HD dvec3 scale3(double a, dvec3 b) { return mk3(a * b.x, a * b.y, a * b.z); }
// cross / dot / length exactly as triangle_mesh.cu:72-87, normal_shader.cu:86-115
RTS_HD dvec3 cross3(dvec3 a, dvec3 b) { return mk3(a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x); }
RTS_HD double dot3(dvec3 a, dvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTS_HD double magsq3(dvec3 a) { return (a.x*a.x + a.y*a.y + a.z*a.z); }
RTS_HD double len3(dvec3 a) { return sqrt(a.x*a.x + a.y*a.y + a.z*a.z); }
RTS_HD dvec3 unit3(dvec3 a) { double n = len3(a); return mk3(a.x/n, a.y/n, a.z/n); }
// normalise_float3 (ray_tracer.cu:125-129): f64 normalise, THEN narrow each component
RTS_HD fvec3 unit3_to_f32(dvec3 a) { double n = len3(a); return mk3f((float)(a.x/n), (float)(a.y/n), (float)(a.z/n)); }
RTS_HD dvec3 widen3(fvec3 a) { return mk3((double)a.x, (double)a.y, (double)a.z); }
RTS_HD float dot3f(fvec3 a, fvec3 b) { return a.x*b.x + a.y*b.y + a.z*b.z; }

// OptiX SDK optixu reflect(i, n) = i - 2*n*dot(n, i), f32 (call site normal_shader.cu:296)
RTS_HD fvec3 reflect3f(fvec3 i, fvec3 n) {
    float d = dot3f(n, i);
    return mk3f(i.x - (2.0f*n.x)*d, i.y - (2.0f*n.y)*d, i.z - (2.0f*n.z)*d);
}

// OptiX SDK optixu refract(r, i, n, ior) (call site normal_shader.cu:212), f32:
//   c = dot(i, n); if c > 0 { eta = ior; n = -n; c = -c } else eta = 1/ior;
//   k = 1 - eta^2 (1 - c^2); k < 0 -> r = 0, false (total internal reflection);
//   else r = normalize(eta*i - (eta*c + sqrtf(k))*n), normalize(v) = v * (1 / sqrtf(dot(v, v)))
RTS_HD bool refract3f(fvec3& r, fvec3 i, fvec3 n, float ior) {
    fvec3 nn = n;
    float negNdotV = dot3f(i, nn);
    float eta;
    if (negNdotV > 0.0f) { eta = ior; nn = mk3f(-n.x, -n.y, -n.z); negNdotV = -negNdotV; }
    else { eta = 1.f / ior; }
    const float k = 1.f - eta*eta * (1.f - negNdotV * negNdotV);
    if (k < 0.0f) { r = mk3f(0.f, 0.f, 0.f); return false; }
    const float sc = eta*negNdotV + sqrtf(k);
    const fvec3 v = mk3f(eta*i.x - sc*nn.x, eta*i.y - sc*nn.y, eta*i.z - sc*nn.z);
    const float inv = 1.0f / sqrtf(dot3f(v, v));
    r = mk3f(v.x*inv, v.y*inv, v.z*inv);
    return true;
}

// f64 -> f32 rounded toward -inf / +inf (__double2float_rd/ru, triangle_mesh.cu:228-229)
RTS_HD float f32_down(double v) {
    float f = (float)v;
    if ((double)f > v) {
        uint32_t u = __builtin_bit_cast(uint32_t, f);
        if (f == 0.0f) u = 0x80000001u; else if (f > 0.0f) u -= 1u; else u += 1u;
        f = __builtin_bit_cast(float, u);
    }
    return f;
}
RTS_HD float f32_up(double v) {
    float f = (float)v;
    if ((double)f < v) {
        uint32_t u = __builtin_bit_cast(uint32_t, f);
        if (f == 0.0f) u = 0x00000001u; else if (f > 0.0f) u += 1u; else u -= 1u;
        f = __builtin_bit_cast(float, u);
    }
    return f;
}
// next f32 above a positive finite value
RTS_HD float f32_next_up_pos(float f) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) + 1u); }

// ---------------------------------------------------------------------------------------
// atan2f for the capture-window test (ray_tracer.cu:326-329).  The reference calls CUDA's
// atan2f (<= 2 ulp, exact bits unknowable).  Here atan2 is evaluated in f64 from basic
// operations and rounded ONCE to f32, which (a) is within the reference's error bound and
// (b) is reproducible bit-for-bit off-device.  atan on [0,1]: reduce about k/8 with
// tabulated atan(k/8), then an odd series to r^17 (|r| <= 1/16 -> error < 1e-17).
RTS_HD double rts_atan_unit(double x) {
    int k = (int)(x * 8.0 + 0.5);
    double c = (double)k * 0.125;
    double r = (x - c) / (1.0 + x * c);
    double r2 = r * r;
    double s = 1.0/17.0;
    s = 1.0/15.0 - r2 * s;
    s = 1.0/13.0 - r2 * s;
    s = 1.0/11.0 - r2 * s;
    s = 1.0/9.0 - r2 * s;
    s = 1.0/7.0 - r2 * s;
    s = 1.0/5.0 - r2 * s;
    s = 1.0/3.0 - r2 * s;
    s = 1.0 - r2 * s;
    // atan(k/8), k = 0..8 (a table, not a switch: the switch became a tree of branches with its constants in scratch)
    static const double atan_k8[9] = { 0.0, 0.12435499454676143503, 0.24497866312686415417, 0.35877067027057222040, 0.46364760900080611621,
                                       0.55859931534356243597, 0.64350110879328438680, 0.71882999962162450542, 0.78539816339744830962 };
    const double base = atan_k8[k < 0 ? 0 : (k > 8 ? 8 : k)];
    return base + r * s;
}
RTS_HD double rts_atan2_f64(double y, double x) {
    const double PI = 3.14159265358979323846, PI_2 = 1.57079632679489661923;
    if (x != x || y != y) return x + y;
    double ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax) a = rts_atan_unit(ay / ax);
    else a = PI_2 - rts_atan_unit(ax / ay);
    if (__builtin_signbit(x)) a = PI - a;
    return __builtin_signbit(y) ? -a : a;
}
RTS_HD float rts_atan2f(float y, float x) { return (float)rts_atan2_f64((double)y, (double)x); }

// normalise_angle / angle_in_range, ray_tracer.cu:53-69
RTS_HD void rts_normalise_angle(double& angle) {
    while (angle < -RTS_PI) angle += 2*RTS_PI;
    while (angle > RTS_PI) angle -= 2*RTS_PI;
}
RTS_HD bool rts_angle_in_range(double testAngle, double a, double b) {
    a -= testAngle; b -= testAngle;
    rts_normalise_angle(a); rts_normalise_angle(b);
    if (a * b >= 0) return false;
    return fabs(a - b) < RTS_PI;
}
