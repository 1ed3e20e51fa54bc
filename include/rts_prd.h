/* rts_prd.h -- the per-ray record and scene epsilons shared by the host simulator and the
 * device path.  This is the boundary type of the reference (ray_tracer.h:9-28): SOARS code
 * that includes "ray_tracer.h" can include this header instead and sees the same names,
 * the same field order and the same 144-byte / 16-aligned layout, so buffers of
 * PerRayData cross the C-ABI unchanged (rs::kernel_wrapper, rts_get_received).
 *
 * Plain C/C++: no CUDA or HIP vector headers are needed on the host side.
 */
#ifndef RTS_PRD_H
#define RTS_PRD_H

#include <stddef.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifndef SCENE_EPS
#define SCENE_EPS 0.005f   /* minimum incident / refracted ray length   (ray_tracer.h:9)  */
#endif
#ifndef SCENE_EPS_R
#define SCENE_EPS_R 0.005f /* minimum reflected ray length               (ray_tracer.h:10) */
#endif

#if defined(__HIPCC__) || defined(__CUDACC__)
/* device compilers already provide double2/double3 with this layout */
#include <hip/hip_runtime.h>
typedef double2 rts_double2;
typedef double3 rts_double3;
#else
typedef struct rts_double2_s { double x, y; } __attribute__((aligned(16))) rts_double2;
typedef struct rts_double3_s { double x, y, z; } rts_double3;
#ifdef RTS_DEFINE_VECTOR_ALIASES   /* opt-in: CUDA/HIP spellings for host code that has no vector header */
typedef rts_double2 double2;
typedef rts_double3 double3;
#endif
#endif

struct PerRayData {
    double rayLength;          /* total path length                                   @0   */
    rts_double2 refrIndex;     /* previous / current refractive index                 @16  */
    unsigned int reflDepth;    /* number of reflections                               @32  */
    unsigned int refrDepth;    /* number of refractions                               @36  */
    unsigned int maxRayIndex;  /* refraction row offset                               @40  */
    rts_double3 rayDirection;  /* (left zero in returned records, ray_tracer.cu:254)  @48  */
    rts_double3 firstHitPoint; /* first hit point                                     @72  */
    rts_double3 prevHitPoint;  /* last hit point                                      @96  */
    double power;              /* partial received power                              @120 */
    double doppler;            /* sum of V.(k1-k0); Doppler [Hz] after finalisation   @128 */
    int received;              /* -1, or receiver index                               @136 */
    bool end;                  /* terminated flag                                     @140 */
};

#ifdef __cplusplus
static_assert(sizeof(PerRayData) == 144, "PerRayData must be 144 bytes (ray_tracer.h:13-28)");
static_assert(alignof(PerRayData) == 16, "PerRayData must be 16-byte aligned");
static_assert(offsetof(PerRayData, refrIndex) == 16 && offsetof(PerRayData, reflDepth) == 32 &&
              offsetof(PerRayData, rayDirection) == 48 && offsetof(PerRayData, firstHitPoint) == 72 &&
              offsetof(PerRayData, prevHitPoint) == 96 && offsetof(PerRayData, power) == 120 &&
              offsetof(PerRayData, doppler) == 128 && offsetof(PerRayData, received) == 136 &&
              offsetof(PerRayData, end) == 140, "PerRayData field offsets");
#endif

#endif /* RTS_PRD_H */
