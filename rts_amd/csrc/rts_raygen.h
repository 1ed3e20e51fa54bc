// rts_raygen.h -- primary ray direction of one launch index (ray_generation, ray_tracer.cu:144-205).
// All trigonometry of the reference's ray generation depends only on launch constants
// (d_txSpan, d_txDir); it is hoisted to the host (rts_api: fill_launch_constants) and the
// device evaluates the remaining per-ray expression tree in the reference's order.
#pragma once
#include "rts_internal.h"

// global launch index of local slot `slot` (contiguous range, or interleaved tiles of the range)
__device__ __forceinline__ uint64_t rts_global_index(const RtsLaunchConsts& a, uint32_t slot)
{
    if (a.il_parts <= 1) return a.ray_first + slot;
    const uint32_t j = slot / a.il_tile, r = slot - j * a.il_tile;
    const uint64_t t = a.il_list ? (uint64_t)a.il_list[j] : (uint64_t)j * a.il_parts + a.il_part;      // (uniform branch; the list entry is one broadcast load per wave tile)
    return a.ray_first + t * a.il_tile + r;
}

// direction of lattice point (lx, ly, lz); host + device: the host uses it for the extent of the primary-ray mask
RTS_HD dvec3 rts_lattice_dir(const RtsLaunchConsts& a, uint32_t lx, uint32_t ly, uint32_t lz)
{
    if (a.W == 1) return mk3(a.w1x, a.w1y, a.w1z);                       // ray_tracer.cu:160-161
    dvec3 v = mk3(a.bsx + a.stx * (double)lx, a.bsy + a.sty * (double)ly, a.bsz + a.stz * (double)lz);   // :167-169
    v = unit3(v);                                                         // :170
    dvec3 r;                                                              // rotated = 0; rotated += Rot*v  :178-182
    r.x = 0.0 + (a.rot[0]*v.x + a.rot[1]*v.y + a.rot[2]*v.z);
    r.y = 0.0 + (a.rot[3]*v.x + a.rot[4]*v.y + a.rot[5]*v.z);
    r.z = 0.0 + (a.rot[6]*v.x + a.rot[7]*v.y + a.rot[8]*v.z);
    v = unit3(r);
    dvec3 dir;                                                            // :199-203 (left un-normalised)
    dir.x = 0.0 + (a.rot1[0]*v.x + a.rot1[1]*v.y + a.rot1[2]*v.z);
    dir.y = 0.0 + (a.rot1[3]*v.x + a.rot1[4]*v.y + a.rot1[5]*v.z);
    dir.z = 0.0 + (a.rot1[6]*v.x + a.rot1[7]*v.y + a.rot1[8]*v.z);
    return dir;
}

// launch index -> lattice coordinates.  rayIndex = z*W*W + y*W + x (:151), an unsigned int in the reference and < 2^32 here
// (rts_create): two 32-bit divisions.
__device__ __forceinline__ void rts_lattice_coords(const RtsLaunchConsts& a, uint32_t slot, uint32_t& lx, uint32_t& ly, uint32_t& lz)
{
    const uint32_t g = (uint32_t)rts_global_index(a, slot);
    // two divisions by the launch constant W: multiply-shift with host-made constants (exact for every 32-bit g; a hardware
    // u32 division is ~35 VALU instructions, and a launch index that hits nothing has little else to do)
    const uint32_t m = a.w_magic, s = a.w_more;
    const uint32_t t0 = __umulhi(m, g), q = (((g - t0) >> 1) + t0) >> s;
    const uint32_t t1 = __umulhi(m, q); lz = (((q - t1) >> 1) + t1) >> s;
    lx = g - q * a.W; ly = q - lz * a.W;
}

__device__ __forceinline__ dvec3 rts_primary_dir(const RtsLaunchConsts& a, uint32_t slot)
{
    if (a.W == 1) return mk3(a.w1x, a.w1y, a.w1z);
    uint32_t lx, ly, lz; rts_lattice_coords(a, slot, lx, ly, lz);
    return rts_lattice_dir(a, lx, ly, lz);
}
