// rts_post.hip -- everything after the trace kernel, on the device:
//   * ordering of the received rays by launch index and expansion into the reference's
//     output records (PerRayData + path row + RCS-angle row)        ray_tracer.cpp:1180-1257
//   * the uniform-gain finalisation                                  ray_tracer.cpp:1219-1253
//   * the aggregation (myKernel1/myKernel2 + unique paths) as a sort / group-by instead of
//     the reference's O(R^2 D) all-pairs scan                        aggregation.cu:32-97
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "rts_internal.h"
#include "rts_raygen.h"

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }
static inline uint32_t rts_small_cap_recv(const RtsContext* c) { return c->last_args.max_refr == 0 ? RTS_SMALL_CAP32 : RTS_SMALL_CAP64; }
// received sets up to this size: single-block ordering / finishing kernels (see k_agg_order_small) -- 4 096 rays when the sort
// keys fit 32 bits (no refraction chains in the row key; a (receiver, path) key of <= 31 bits), 2 048 with 64-bit keys: the
// block's sort storage has to stay below the 40 KB of a free block slot
#define RTS_SMALL_THREADS 256
static_assert(RTS_SMALL_CAP32 == 16 * RTS_SMALL_THREADS && RTS_SMALL_CAP64 == 8 * RTS_SMALL_THREADS, "items per thread of the one-block sorts");
template <typename K, int ITEMS> __global__ void k_recv_order_small(const RtsEndRecord* __restrict__ rec, uint32_t n, uint32_t n_rays, int with_chain, uint32_t bits, uint32_t* __restrict__ perm, const unsigned long long* __restrict__ R_dev);

// --------------------------------------------------------------------------- record expansion
// received rays are ordered as the reference's host scan meets them (ray_tracer.cpp:1190): ascending buffer
// row = chain * n + launch slot
__global__ void k_recv_keys(const RtsEndRecord* __restrict__ rec, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = rec[i].slot; vals[i] = i; }
}
__global__ void k_recv_keys64(const RtsEndRecord* __restrict__ rec, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n, uint32_t n_rays)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = (uint64_t)(rec[i].pad & 3u) * n_rays + rec[i].slot; vals[i] = i; }
}

// cart_to_sph, normal_shader.cu:118-124
__device__ __forceinline__ void cart_to_sph(dvec3 v, double& az, double& el) { az = atan2(v.y, v.x); el = atan2(v.z, sqrt(v.x*v.x + v.y*v.y)); }

__device__ __forceinline__ dvec3 hist_dir(const RtsTraceArgs& a, size_t plane, uint32_t slot) {
    const float* dh = a.dir_hist + plane * 3 * a.n_rays;
    return unit3(mk3((double)dh[slot], (double)dh[(size_t)a.n_rays + slot], (double)dh[2*(size_t)a.n_rays + slot]));
}
__device__ __forceinline__ void put_angle(double* angles, size_t row, uint32_t D, uint32_t col, dvec3 k0, dvec3 k1) {
    double a0, e0, a1, e1;                     // tAngle = sph(k0) + sph(-k1)   (normal_shader.cu:259-265, 320-326)
    cart_to_sph(k0, a0, e0);
    cart_to_sph(mk3(-k1.x, -k1.y, -k1.z), a1, e1);
    angles[(row*D + col)*2] = a0 + a1; angles[(row*D + col)*2 + 1] = e0 + e1;
}

// One thread per output row j.  perm != nullptr: row j is received record perm[j].  perm == nullptr: the full
// buffers of the reference (keep-all): row j = chain * n + slot for chain < rows, rec = all_records.
__device__ __forceinline__ void expand_row(const RtsTraceArgs& a, const RtsEndRecord* __restrict__ rec, const uint32_t* __restrict__ perm, const uint32_t j, const uint32_t D,
                                           PerRayData* __restrict__ rays, int32_t* __restrict__ paths, double* __restrict__ angles, uint64_t* __restrict__ slots)
{
    RtsEndRecord r; bool valid = true; uint32_t chain, slot, prefill_code = 0;
    if (perm) { r = rec[perm[j]]; chain = r.pad & 3u; slot = r.slot; }
    else {
        chain = j / a.n_rays; slot = j - chain * a.n_rays;
        const RtsEndRecord r0 = rec[slot];
        prefill_code = (r0.pad >> 8) & 0xffu;
        if (chain == 0) r = r0;
        else if (chain == 1) { valid = (r0.pad >> 16) & 1u; if (valid) r = rec[(size_t)a.n_rays + slot]; }
        else if (chain == 2) { valid = (r0.pad >> 16) & 1u; if (valid) { const RtsEndRecord r1 = rec[(size_t)a.n_rays + slot]; valid = (r1.pad >> 17) & 1u; if (valid) r = rec[2*(size_t)a.n_rays + slot]; } }
        else valid = false;
    }
    // dbuf_results[row] as left by ray_generation: initialised fields (ray_tracer.cu:227-240) overwritten by the
    // eight written-back ones (:246-253, normal_shader.cu:272-279)
    PerRayData o;
    __builtin_memset(&o, 0, sizeof(o));
    o.refrIndex.x = 1; o.refrIndex.y = 1; o.received = -1; o.end = false;
    if (valid) {
        o.rayLength = r.rayLength; o.reflDepth = r.reflDepth; o.refrDepth = (r.pad >> 2) & 3u;
        o.firstHitPoint.x = r.firstx; o.firstHitPoint.y = r.firsty; o.firstHitPoint.z = r.firstz;
        o.prevHitPoint.x = r.prevx; o.prevHitPoint.y = r.prevy; o.prevHitPoint.z = r.prevz;
        o.power = r.power; o.doppler = r.doppler; o.received = r.received;
    }
    rays[j] = o;
    if (slots) slots[j] = rts_global_index(*a.lc, slot) + (uint64_t)chain * ((uint64_t)a.W * a.W * a.W);
    // path row: dbuf_targ_intersect[row][col], -1 default (ray_tracer.cpp:854-857, normal_shader.cu:140-146, 221-239)
    for (uint32_t col = 0; col < D; col++) {
        int code = 0;
        if (valid) { if (col < 8) code = (int)((r.path_lo >> (8*col)) & 0xff); else if (col < 16) code = (int)((r.path_hi >> (8*(col-8))) & 0xff); }
        else if (chain >= 1 && prefill_code && (chain == 1 || col < chain)) code = (int)prefill_code;
        paths[(size_t)j*D + col] = code - 1;
    }
    for (uint32_t col = 0; col < D; col++) { angles[((size_t)j*D + col)*2] = -1000000; angles[((size_t)j*D + col)*2 + 1] = -1000000; }   // ray_tracer.cpp:861-867
    if (!valid) return;
    // RCS angle rows, rebuilt from the f32 direction history
    const dvec3 kprim = unit3(rts_primary_dir(*a.lc, slot));
    if (a.max_refr == 0) {
        // ... and, in the product builds, the Doppler sum the trace kernel left out (rts_trace.hip rts_shade; normal_shader.cu:302-314): per reflection
        // V(target of the path column) . (unit(new direction) - unit(old direction)), added in bounce order from 0 -- the unit vectors are the ones of the angles.
        dvec3 kin = kprim; double doppler = 0;
        for (uint32_t c = 0; c < r.reflDepth; c++) {
            const dvec3 k1 = hist_dir(a, c, slot);
            if (c < D) put_angle(angles, j, D, c, kin, k1);
            if (!a.keep_all) {
                const uint32_t code = (uint32_t)(((c < 8 ? r.path_lo : r.path_hi) >> (8*(c & 7u))) & 0xff);
                const RtsTargetDev T = a.targets[code ? code - 1u : 0u];          // (every reflection wrote its column: reflDepth <= max_refl <= D)
                doppler += dot3(mk3(T.vx, T.vy, T.vz), sub3(k1, kin));
            }
            kin = k1;
        }
        if (!a.keep_all) rays[j].doppler = doppler;
    } else {
        const size_t P = a.max_refl + 1;
        // with refraction enabled the depth gate (normal_shader.cu:134) also shades a hit at reflDepth == maxRefl,
        // which then does NOT reflect (:293): reflections performed = min(reflDepth, maxRefl)
        const uint32_t nrefl = r.reflDepth < a.max_refl ? r.reflDepth : a.max_refl;
        if (chain == 0) {
            dvec3 kin = kprim;
            for (uint32_t c = 0; c < nrefl && c < D; c++) { const dvec3 k1 = hist_dir(a, c + 1, slot); put_angle(angles, j, D, c, kin, k1); kin = k1; }
        } else if (chain == 1) {
            dvec3 kin = hist_dir(a, P, slot);
            put_angle(angles, j, D, 0, kprim, kin);
            for (uint32_t q = 1; q <= nrefl && q < D; q++) { const dvec3 k1 = hist_dir(a, P + q, slot); put_angle(angles, j, D, q, kin, k1); kin = k1; }
        } else {
            const dvec3 kpar = hist_dir(a, P, slot);
            dvec3 kin = hist_dir(a, 2*P, slot);
            put_angle(angles, j, D, 1, kpar, kin);
            for (uint32_t q = 1; q <= nrefl && q + 1 < D; q++) { const dvec3 k1 = hist_dir(a, 2*P + q, slot); put_angle(angles, j, D, q + 1, kin, k1); kin = k1; }
        }
    }
}
__global__ void k_expand(const RtsTraceArgs a, const RtsEndRecord* __restrict__ rec, const uint32_t* __restrict__ perm, uint32_t n_out, uint32_t D,
                         PerRayData* __restrict__ rays, int32_t* __restrict__ paths, double* __restrict__ angles, uint64_t* __restrict__ slots, const unsigned long long* __restrict__ R_dev)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    { uint32_t R = n_out; if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; } n_out = R; }
    if (j >= n_out) return;
    expand_row(a, rec, perm, j, D, rays, paths, angles, slots);
}

__global__ void k_fill_i32(int32_t* p, int32_t v, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
int rts_fill_i32(hipStream_t st, int32_t* p, int32_t v, size_t n) { if (n) k_fill_i32<<<blocks_for(n, 256), 256, 0, st>>>(p, v, n); return RTS_OK; }

// --------------------------------------------------------------------------- tile order of the next launch
// The handle keeps what every GLOBAL wave tile (64 consecutive launch indices of the W^3 lattice) cost the last time one of
// its launches traced it, so the history carries over between launch shapes (whole pulse, contiguous shard, interleaved
// part): a launch's local tile j is global tile (ray_first + local index of its first ray) / 64.
struct RtsTileShape { uint64_t first; uint32_t il_tile, il_parts, il_part, n_tiles; const uint32_t* il_list; };

__device__ __forceinline__ uint32_t tile_global(const RtsTileShape& s, uint32_t j)
{
    const uint64_t slot = (uint64_t)j * RTS_WTILE;
    if (s.il_parts <= 1) return (uint32_t)((s.first + slot) / RTS_WTILE);
    const uint64_t t = slot / s.il_tile, r = slot - t * s.il_tile;
    const uint64_t g = s.il_list ? (uint64_t)s.il_list[t] : t * s.il_parts + s.il_part;      // a dealt list of tiles (rts_set_tile_list) or interleaved parts
    return (uint32_t)((s.first + g * s.il_tile + r) / RTS_WTILE);
}

// Head of the cost order (cooperative units, rts_trace.hip): the tiles flagged LONG WALKS (bit 31 of the cost record) whose
// estimated cost exceeds `frac` of the launch's balanced time -- sum of the tile costs / resident waves -- and a floor.  The
// sum is formed while the previous launch's costs are merged into the history, the count while the sort keys are written:
// no kernel, no readback of its own (a single-block reduction + a 4-byte device-to-host copy per pulse on the handle's
// stream cost the three-pulse pipeline 7 %: 0.68 -> 0.73 ms per pulse).  head[0..1] = sum (u64), head[2] = count; zeroed with
// the draw counters they share a buffer with.
struct RtsHeadRule { double frac, big, mid; uint32_t floor_cost, resident_waves; };
// A tile goes to the head of the order -- to the cooperative kernel -- if its cost record says
//   LONG WALKS (bit 31: >= coop_walk_steps walk iterations per bounce round, 1 000) and it cost more than frac x the launch's balanced time; or
//   LONGISH WALKS (bit 30: >= coop_walk_steps_lo, 400) and it cost more than mid x the balanced time (1.5 since round 5; 3 before: alone it would triple the
//   launch -- a launch that consists of its tail, e.g. one GPU's eighth of a BASELINE configs[3] pulse, where the ~5 x more work of
//   64 units with short walks is free because the chip is idle; in a launch that is busy throughout the same tile stays where it is:
//   BASELINE configs[4] lost 9 % when such tiles were made cooperative, profiles/r04_coop_steps_scan.log); or
//   the experiment RtsContext::coop_big.
__device__ __forceinline__ bool rts_head_rule(const uint32_t est, const double cost, const double balanced, const double thr, const double thr_big, const RtsHeadRule& rule)
{
    const double thr_mid = fmax(rule.mid * balanced, (double)rule.floor_cost);
    return ((est >> 31) && cost > thr) || (rule.mid > 0.0 && ((est >> 30) & 1u) && balanced > 0.0 && cost > thr_mid) || (rule.big > 0.0 && balanced > 0.0 && cost > thr_big);
}

// A record left by the COOPERATIVE kernel (bit 31 without bit 30: the ordinary kernel never writes that -- its LONG WALKS imply LONGISH) is the sum
// of the tile's 64 units' wave time, and that is NOT "at least what the tile costs one wave" (round 3's assumption): a tile with one straggling
// ray of 6 ms is walked by 64 lanes in 0.1 ms, its units sum to less than the head rule's threshold, the next launch hands it back to the
// ordinary kernel -- 6 ms again, one wave, the launch's tail -- and the launch after that makes it cooperative again (BASELINE configs[3] with
// longish tiles admitted: lone launches 5.0 / 6.1 / 5.0 / 6.1 ms, the ordinary kernel 4.0 / 5.9, profiles/r05c_c4_tail_scan.log).  Such a record keeps
// the larger of its own cost and 0.95 x what the history held: a tile stays at the head while it deserves to (its last ordinary cost decays over
// ~20 launches) and one that has become cheap -- the target moved on -- leaves it.  The LAUNCH's cost sum takes what was measured.
__device__ __forceinline__ uint32_t rts_record_to_keep(const uint32_t v, const uint32_t old)
{
    if ((v >> 30) != 2u) return v;
    const uint32_t keep = (uint32_t)(0.95f * (float)(old & 0x3fffffffu));
    return 0x80000000u | max(v & 0x3fffffffu, keep);
}

// fold the costs measured by the previous launch into the history
__global__ void k_tile_merge(uint32_t* __restrict__ cost, RtsTileShape prev, uint32_t* __restrict__ hist, uint32_t n_hist, unsigned long long* __restrict__ head_sum, uint32_t* __restrict__ coarse)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v64 = 0;
    if (j < prev.n_tiles) {
        const uint32_t v = cost[j], g = tile_global(prev, j);
        if (v && g < n_hist) hist[g] = rts_record_to_keep(v, hist[g]);
        v64 = v & 0x3fffffffu;
        cost[j] = 0u;                                                          // (ready for the coming launch: no fill of its own)
    }
    if (coarse) {                                                              // (uniform) XCD-affine sub-orders: the launch's cost by 1/1024 of its tile range, in units of 16
        __shared__ uint32_t s_cc[257];                                         // a block's 256 tiles touch at most 257 cells (one or two when the launch is large)
        for (uint32_t q = threadIdx.x; q < 257u; q += blockDim.x) s_cc[q] = 0u;
        __syncthreads();
        const uint32_t cell0 = (uint32_t)(((unsigned long long)(blockIdx.x * blockDim.x) * RTS_COARSE_CELLS) / prev.n_tiles);
        if (j < prev.n_tiles && v64) atomicAdd(&s_cc[(uint32_t)(((unsigned long long)j * RTS_COARSE_CELLS) / prev.n_tiles) - cell0], (uint32_t)((v64 + 15u) >> 4));
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < 257u; q += blockDim.x) if (s_cc[q] && cell0 + q < RTS_COARSE_CELLS) atomicAdd(&coarse[cell0 + q], s_cc[q]);
        __syncthreads();
    }
    // one atomic per BLOCK (per wave they were 2 400 on one address for a C3 launch: ~25 us of serialised L2 atomics on the
    // critical chain of every pulse)
    __shared__ unsigned long long s_part[4];
    for (int o = 32; o > 0; o >>= 1) v64 += __shfl_down(v64, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v64;
    __syncthreads();
    if (threadIdx.x == 0 && head_sum) { const unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3]; if (t) atomicAdd(head_sum, t); }
}

// sort key of local tile j of the coming launch: ~(estimated cost record); a tile never traced yet takes the largest record
// known within 16 global tiles of it (expensive regions are contiguous in launch-index space).  Flagged records sort first
// (their bit 31), by descending cost: the flagged tiles above the threshold are a prefix of the order.
__global__ void k_tile_keys(const uint32_t* __restrict__ hist, uint32_t n_hist, RtsTileShape cur, uint32_t* __restrict__ key, uint32_t* __restrict__ id,
                            const unsigned long long* __restrict__ head_sum, uint32_t* __restrict__ head_count, RtsHeadRule rule, uint32_t* __restrict__ bucket_hist,
                            int affine, const uint32_t* __restrict__ bnd)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t is_head = 0, bucket = 0;
    if (j < cur.n_tiles) {
        const uint32_t g = tile_global(cur, j);
        uint32_t est = g < n_hist ? hist[g] : 0u;
        if (est == 0) {
            for (uint32_t d = 1; d <= 16; d++) {
                if (g >= d && g - d < n_hist) est = max(est, hist[g - d]);
                if (g + d < n_hist) est = max(est, hist[g + d]);
            }
        }
        // head of the order = the tiles the cooperative kernel traces: LONG WALKS tiles above frac x the balanced time (and, as an
        // experiment that is OFF by default -- RtsContext::coop_big -- any tile above big x the balanced time).  The key's top bit
        // is this DECISION (not the record's flag): the head is a prefix of the sorted order.
        const uint32_t cost = est & 0x3fffffffu;
        if (head_count && rule.frac > 0.0) {
            const double balanced = (double)head_sum[0] / (double)(rule.resident_waves ? rule.resident_waves : 1u);
            double thr = rule.frac * balanced; if (thr < (double)rule.floor_cost) thr = (double)rule.floor_cost;
            double thr_big = rule.big * balanced; if (thr_big < (double)rule.floor_cost) thr_big = (double)rule.floor_cost;
            is_head = rts_head_rule(est, (double)cost, balanced, thr, thr_big, rule) ? 1u : 0u;
        }
        key[j] = ~((is_head << 31) | cost); id[j] = j;
        // bucket of the counting order (bucket_hist != nullptr): the head of the order in the first half of the bins, the rest in
        // the second, each by descending cost in steps of 1/16 octave -- finer than a tile's cost repeats from pulse to pulse
        // (the head is ordered too: its longest cooperative unit has to start first)
        bucket = (is_head ? 0u : RTS_TILE_BUCKETS / 2u) + (RTS_TILE_BUCKETS / 2u - 1u) - (cost <= 1u ? 16u * cost : min((uint32_t)(__log2f((float)cost + 1.0f) * 16.0f), RTS_TILE_BUCKETS / 2u - 1u));      // (cost 1, a dead tile: RTS_DEAD_BIN, said exactly)
        if (affine) {
            // XCD-affine sub-orders: [64 head bins, half octaves | RTS_XCD bands x 120 bins, quarter octaves]; band = the contiguous range
            // of local tile indices (a slab of the lattice) that held an eighth of the cost of the launch before last (bnd: written by
            // k_tile_bucket_scan of the previous build; none yet: equal counts)
            uint32_t band = 0;
            if (bnd) { for (uint32_t r = 1; r < RTS_XCD; r++) if (j >= bnd[r]) band = r; }
            else band = min((uint32_t)(((unsigned long long)j * RTS_XCD) / cur.n_tiles), (uint32_t)RTS_XCD - 1u);
            const float lg = __log2f((float)cost + 1.0f);
            bucket = is_head ? 63u - min((uint32_t)(lg * 2.0f), 63u) : 64u + band * 120u + 119u - min((uint32_t)(lg * 4.0f), 119u);
        }
        if (bucket_hist) key[j] = bucket;
    }
    if (head_count) {                                                          // (uniform)
        const unsigned long long m = __ballot(is_head != 0);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(head_count, (uint32_t)__popcll(m));      // (heads are rare: a handful of waves at most)
    }
    if (bucket_hist) {                                                         // (uniform) block histogram in LDS, one global atomic per non-empty bin
        __shared__ uint32_t s_cnt[RTS_TILE_BUCKETS];
        for (uint32_t b = threadIdx.x; b < RTS_TILE_BUCKETS; b += blockDim.x) s_cnt[b] = 0u;
        __syncthreads();
        if (j < cur.n_tiles) atomicAdd(&s_cnt[bucket], 1u);
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < RTS_TILE_BUCKETS; b += blockDim.x) if (s_cnt[b]) atomicAdd(&bucket_hist[b], s_cnt[b]);
    }
}

// Counting order of the tiles (instead of a device-wide radix sort of 157 k keys -- eight launches on the chain in front of
// every trace launch): histogram in k_tile_keys, exclusive scan of the 1 024 bins by one block, and a scatter in which every
// block reserves its share of each bin with ONE atomic and ranks its tiles inside it in LDS.  Tiles of a bin come out in no
// particular order (only the schedule depends on it); the head of the order is the first half of the bins, a prefix as before.
// xcd != null (XCD-affine sub-orders): xcd[0 .. RTS_XCD] <- first order position of each band's segment (the last entry: the number of
// tiles), and xcd[16 .. 16 + RTS_XCD] <- the bands of the NEXT build: local tile indices at which the cost cells of the launch just
// merged (coarse) reach 1/8, 2/8, ... of their sum
// The bin of a DEAD tile (cost record 1: every launch index cleared by the pre-filter, k_trace): everything in front of it in the order cost more.
// Its first position + 1 goes to `live` (the trace kernel draws the order behind it 64 tiles at a time, one lane per tile).
#define RTS_DEAD_BIN (RTS_TILE_BUCKETS / 2u + (RTS_TILE_BUCKETS / 2u - 1u) - 16u)       // (is_head ? 0 : 512) + 511 - min(log2(1 + 1) * 16, 511)
__global__ void __launch_bounds__(RTS_TILE_BUCKETS) k_tile_bucket_scan(uint32_t* __restrict__ hist, uint32_t* __restrict__ xcd, const uint32_t* __restrict__ coarse, uint32_t n_tiles, uint32_t* __restrict__ live)
{
    __shared__ uint32_t s[2][RTS_TILE_BUCKETS];
    const uint32_t t = threadIdx.x, v = hist[t];
    s[0][t] = v; __syncthreads();
    int cur = 0;
    for (uint32_t off = 1; off < RTS_TILE_BUCKETS; off <<= 1) { uint32_t x = s[cur][t]; if (t >= off) x += s[cur][t - off]; s[cur ^ 1][t] = x; cur ^= 1; __syncthreads(); }
    hist[t] = s[cur][t] - v;                                                   // exclusive: first position of the bin
    if (live && t == RTS_DEAD_BIN) live[0] = s[cur][t] - v + 1u;
    if (!xcd) return;                                                          // (uniform)
    if (t >= 64u && (t - 64u) % 120u == 0u && (t - 64u) / 120u < RTS_XCD) xcd[(t - 64u) / 120u] = s[cur][t] - v;
    if (t == 0) xcd[RTS_XCD] = s[cur][RTS_TILE_BUCKETS - 1];
    __syncthreads();
    static_assert(RTS_COARSE_CELLS == RTS_TILE_BUCKETS, "one thread per coarse cell");
    __shared__ unsigned long long p[2][RTS_COARSE_CELLS];
    p[0][t] = coarse[t]; __syncthreads();
    cur = 0;
    for (uint32_t off = 1; off < RTS_COARSE_CELLS; off <<= 1) { unsigned long long x = p[cur][t]; if (t >= off) x += p[cur][t - off]; p[cur ^ 1][t] = x; cur ^= 1; __syncthreads(); }
    const unsigned long long total = p[cur][RTS_COARSE_CELLS - 1], mine = p[cur][t], before = t ? p[cur][t - 1] : 0ULL;
    if (t == 0) { xcd[16] = 0u; xcd[16 + RTS_XCD] = n_tiles; }
    for (uint32_t r = 1; r < RTS_XCD; r++) {
        if (total == 0ULL) { if (t == 0) xcd[16 + r] = (uint32_t)(((unsigned long long)n_tiles * r) / RTS_XCD); continue; }
        const unsigned long long thr = (total * r + RTS_XCD - 1u) / RTS_XCD;      // band r starts behind the cell in which the running sum reaches r/8 of the total
        if (before < thr && mine >= thr) xcd[16 + r] = (uint32_t)(((unsigned long long)(t + 1u) * n_tiles) / RTS_COARSE_CELLS);
    }
}
__global__ void __launch_bounds__(256) k_tile_bucket_scatter(const uint32_t* __restrict__ bucket_of, uint32_t n, uint32_t* __restrict__ next, uint32_t* __restrict__ order)
{
    __shared__ uint32_t s_cnt[RTS_TILE_BUCKETS], s_base[RTS_TILE_BUCKETS];
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t b = threadIdx.x; b < RTS_TILE_BUCKETS; b += blockDim.x) s_cnt[b] = 0u;
    __syncthreads();
    uint32_t b = 0, r = 0;
    if (j < n) { b = bucket_of[j]; r = atomicAdd(&s_cnt[b], 1u); }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < RTS_TILE_BUCKETS; q += blockDim.x) if (s_cnt[q]) s_base[q] = atomicAdd(&next[q], s_cnt[q]);
    __syncthreads();
    if (j < n) order[s_base[b] + r] = j;
}

// ---- the same order in TWO launches instead of four (r04), for the common case: the launch before had this launch's shape, so
// every tile has a fresh cost record and k_tile_merge + k_tile_keys are one pass over the tiles; and every block of the scatter
// scans the 1 024 bins for itself instead of waiting for a one-block scan kernel.  The head decision needs the launch's balanced
// time -- the SUM of the costs, a grid-wide quantity: the sum of the launch BEFORE (persist[0..1], written by the previous
// build's scatter) stands in for it; pulses of an interval look alike, and only the schedule depends on it.
// `per`: wave tiles per THREAD (1 up to 2^18 tiles).  Every block ends with one atomic on the launch's cost sum and one per non-empty bin;
// most tiles of a pulse are dead and share ONE bin, so those are same-address atomics, ~12 ns apiece at the L2: BASELINE configs[3]'s
// 1.57 M tiles in 6 136 blocks of 256 spent 145 us per kernel on them (configs[2]'s 616 blocks: 10 us).  Fatter blocks: never more than 1 024.
__global__ void k_tile_merge_keys(uint32_t* __restrict__ cost, RtsTileShape cur, uint32_t* __restrict__ hist, uint32_t n_hist, unsigned long long* __restrict__ head_sum,
                                  const unsigned long long* __restrict__ sum_before, uint32_t* __restrict__ head_count, RtsHeadRule rule, uint32_t* __restrict__ key, uint32_t* __restrict__ bucket_hist,
                                  uint32_t per)
{
    __shared__ unsigned long long s_part[4];
    __shared__ uint32_t s_cnt[RTS_TILE_BUCKETS];
    for (uint32_t b = threadIdx.x; b < RTS_TILE_BUCKETS; b += blockDim.x) s_cnt[b] = 0u;
    __syncthreads();
    unsigned long long v64 = 0; uint32_t n_head_wave = 0;
    for (uint32_t it = 0; it < per; it++) {
        const uint32_t j = (blockIdx.x * per + it) * blockDim.x + threadIdx.x;
        uint32_t is_head = 0;
        if (j < cur.n_tiles) {
            const uint32_t v = cost[j], g = tile_global(cur, j);
            const uint32_t vk = (v && g < n_hist) ? rts_record_to_keep(v, hist[g]) : v;
            if (v && g < n_hist) hist[g] = vk;
            cost[j] = 0u;
            uint32_t est = v ? vk : (g < n_hist ? hist[g] : 0u);
            v64 += v & 0x3fffffffu;
            const uint32_t c = est & 0x3fffffffu;
            if (head_count && rule.frac > 0.0) {
                const double balanced = (double)sum_before[0] / (double)(rule.resident_waves ? rule.resident_waves : 1u);
                double thr = rule.frac * balanced; if (thr < (double)rule.floor_cost) thr = (double)rule.floor_cost;
                double thr_big = rule.big * balanced; if (thr_big < (double)rule.floor_cost) thr_big = (double)rule.floor_cost;
                is_head = rts_head_rule(est, (double)c, balanced, thr, thr_big, rule) ? 1u : 0u;
            }
            const uint32_t bucket = (is_head ? 0u : RTS_TILE_BUCKETS / 2u) + (RTS_TILE_BUCKETS / 2u - 1u) - (c <= 1u ? 16u * c : min((uint32_t)(__log2f((float)c + 1.0f) * 16.0f), RTS_TILE_BUCKETS / 2u - 1u));      // (cost 1, a dead tile: RTS_DEAD_BIN, said exactly)
            key[j] = bucket;
            atomicAdd(&s_cnt[bucket], 1u);
        }
        if (head_count) n_head_wave += (uint32_t)__popcll(__ballot(is_head != 0));      // (uniform per wave)
    }
    for (int o = 32; o > 0; o >>= 1) v64 += __shfl_down(v64, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v64;
    if (head_count && (threadIdx.x & 63) == 0 && n_head_wave) atomicAdd(head_count, n_head_wave);
    __syncthreads();
    if (threadIdx.x == 0 && head_sum) { const unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3]; if (t) atomicAdd(head_sum, t); }
    for (uint32_t b = threadIdx.x; b < RTS_TILE_BUCKETS; b += blockDim.x) if (s_cnt[b]) atomicAdd(&bucket_hist[b], s_cnt[b]);
}
// scatter with the bins' scan inside every block: hist = the bins' COUNTS (left untouched), taken = zeroed reservation counters.  Two passes over the
// block's `per` x 256 tiles: count per bin, reserve each bin's share with ONE atomic, then rank inside it (the order inside a bin is free)
__global__ void __launch_bounds__(256) k_tile_scan_scatter(const uint32_t* __restrict__ bucket_of, uint32_t n, const uint32_t* __restrict__ hist, uint32_t* __restrict__ taken, uint32_t* __restrict__ order,
                                                           const unsigned long long* __restrict__ head_sum, unsigned long long* __restrict__ sum_persist, uint32_t per, uint32_t* __restrict__ live)
{
    __shared__ uint32_t s_cnt[RTS_TILE_BUCKETS], s_base[RTS_TILE_BUCKETS], s_rank[RTS_TILE_BUCKETS], s_wave[4];
    const uint32_t t = threadIdx.x;
    if (blockIdx.x == 0 && t == 0 && sum_persist) sum_persist[0] = head_sum[0];      // (complete: the keys kernel is over) the next build's "sum of the launch before"
    static_assert(RTS_TILE_BUCKETS == 4 * 256, "four bins per thread");
    const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
    s_cnt[4 * t] = 0u; s_cnt[4 * t + 1] = 0u; s_cnt[4 * t + 2] = 0u; s_cnt[4 * t + 3] = 0u;
    s_rank[4 * t] = 0u; s_rank[4 * t + 1] = 0u; s_rank[4 * t + 2] = 0u; s_rank[4 * t + 3] = 0u;
    uint32_t x = h0 + h1 + h2 + h3;                              // inclusive scan of the threads' sums: wave shuffle, then the four waves
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((t & 63u) >= (uint32_t)o) x += y; }
    if ((t & 63u) == 63u) s_wave[t >> 6] = x;
    __syncthreads();
    uint32_t before = x - (h0 + h1 + h2 + h3);
    for (uint32_t w = 0; w < (t >> 6); w++) before += s_wave[w];
    s_base[4 * t] = before; s_base[4 * t + 1] = before + h0; s_base[4 * t + 2] = before + h0 + h1; s_base[4 * t + 3] = before + h0 + h1 + h2;
    __syncthreads();
    if (live && blockIdx.x == 0 && t == 0) live[0] = s_base[RTS_DEAD_BIN] + 1u;      // first position of the dead tiles' bin (+ 1: 0 means unknown)
    for (uint32_t it = 0; it < per; it++) { const uint32_t j = (blockIdx.x * per + it) * blockDim.x + t; if (j < n) atomicAdd(&s_cnt[bucket_of[j]], 1u); }
    __syncthreads();
    for (uint32_t q = t; q < RTS_TILE_BUCKETS; q += blockDim.x) if (s_cnt[q]) s_base[q] += atomicAdd(&taken[q], s_cnt[q]);
    __syncthreads();
    for (uint32_t it = 0; it < per; it++) { const uint32_t j = (blockIdx.x * per + it) * blockDim.x + t; if (j < n) { const uint32_t b = bucket_of[j]; order[s_base[b] + atomicAdd(&s_rank[b], 1u)] = j; } }
}

// ---- the history as a table other workers can use (rts_tile_records_get / _set: ray sharding dealt by last-seen cost)
static RtsTileShape rts_shape_of(const RtsContext* c, const uint64_t* sig)
{
    RtsTileShape s; s.first = sig[1]; s.il_tile = (uint32_t)(sig[2] & 0xffffffffu); s.il_parts = (uint32_t)(sig[2] >> 32); s.il_part = (uint32_t)sig[3];
    s.n_tiles = (uint32_t)((sig[0] + RTS_WTILE - 1) / RTS_WTILE); s.il_list = s.il_parts == RTS_INTERLEAVE_LIST ? c->d_il_list.p : nullptr; return s;
}
__global__ void k_tile_records_masked(const uint32_t* __restrict__ hist, uint32_t n_hist, RtsTileShape last, uint32_t* __restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= last.n_tiles) return;
    const uint32_t g = tile_global(last, j);
    if (g < n_hist) out[g] = hist[g];
}
// the cost of the coming launch as the history knows it (the order build's "sum of the previous launch" when there was none of
// this handle's to merge: a history set from other workers' records, a change of launch shape)
__global__ void k_tile_est_sum(const uint32_t* __restrict__ hist, uint32_t n_hist, RtsTileShape cur, unsigned long long* __restrict__ head_sum)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v64 = 0;
    if (j < cur.n_tiles) { const uint32_t g = tile_global(cur, j); if (g < n_hist) v64 = hist[g] & 0x3fffffffu; }
    __shared__ unsigned long long s_part[4];
    for (int o = 32; o > 0; o >>= 1) v64 += __shfl_down(v64, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v64;
    __syncthreads();
    if (threadIdx.x == 0) { const unsigned long long t = s_part[0] + s_part[1] + s_part[2] + s_part[3]; if (t) atomicAdd(head_sum, t); }
}
int rts_tile_costs_flush(RtsContext* c)
{
    if (!c->tile_cost_pending) return RTS_OK;
    c->tile_cost_pending = false; c->order_sum_valid = false;
    if (c->hist->n == 0 || !c->d_tile_cost.p || !c->hist->d.p) return RTS_OK;
    const RtsTileShape p = rts_shape_of(c, c->tile_cost_sig);
    if (p.n_tiles == 0) return RTS_OK;
    k_tile_merge<<<blocks_for(p.n_tiles, 256), 256, 0, c->stream>>>(c->d_tile_cost.p, p, c->hist->d.p, c->hist->n, nullptr, nullptr);
    RTS_HIP(hipGetLastError());
    c->hist->any = true;
    return RTS_OK;
}
int rts_tile_records_masked(RtsContext* c, uint32_t* d_out, uint32_t n)
{
    RTS_HIP(hipMemsetAsync(d_out, 0, sizeof(uint32_t) * n, c->stream));
    if (!c->tile_last_valid || c->hist->n == 0 || !c->hist->d.p) return RTS_OK;
    const RtsTileShape p = rts_shape_of(c, c->tile_last_sig);
    if (p.n_tiles == 0) return RTS_OK;
    k_tile_records_masked<<<blocks_for(p.n_tiles, 256), 256, 0, c->stream>>>(c->hist->d.p, std::min(c->hist->n, n), p, d_out);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// prev_valid: d_tile_cost holds the costs of a launch of shape prev_shape that have not been merged yet
int rts_tile_order_build(RtsContext* c, const uint64_t* prev_sig, bool prev_valid, const uint64_t* cur_sig, uint32_t n_tiles_cur, uint32_t resident_waves)
{
    hipStream_t st = c->stream;
    const uint32_t n_hist = c->hist->n;
    // (a launch over a dealt tile list has il_parts = RTS_INTERLEAVE_LIST and the list's generation in il_part; the list on the device is
    // the one BOTH shapes mean: rts_set_tile_list merges pending cost records before it replaces the list)
    auto shape = [c](const uint64_t* sig) { RtsTileShape s; s.first = sig[1]; s.il_tile = (uint32_t)(sig[2] & 0xffffffffu); s.il_parts = (uint32_t)(sig[2] >> 32); s.il_part = (uint32_t)sig[3];
                                           s.n_tiles = (uint32_t)((sig[0] + RTS_WTILE - 1) / RTS_WTILE); s.il_list = s.il_parts == RTS_INTERLEAVE_LIST ? c->d_il_list.p : nullptr; return s; };
    uint32_t* head = c->coop_frac > 0.0 ? c->d_tile_ctr.p + RTS_OFF_HEAD : nullptr;      // [sum lo, sum hi, count, pad]: zeroed with the draw counters
    uint32_t* bins = c->tile_bucket_order ? c->d_tile_ctr.p + RTS_OFF_BINS : nullptr;      // zeroed with the draw counters
    // XCD-AFFINE sub-orders (RtsContext::xcd_affine: 0 never -- the default: measured slower, DESIGN.md section 5 --, 1 whenever there is a counting order, 2 for launches of
    // >= 2^18 wave tiles, i.e. BASELINE configs[3]'s 100 M launch indices, whose scene is a hundred times an XCD's L2)
    const bool affine = bins != nullptr && (c->xcd_affine == 1 || (c->xcd_affine == 2 && n_tiles_cur >= (1u << 18)));
    c->xcd_affine_now = affine;
    if (affine) RTS_HIP(c->d_xcd.reserve(64));
    uint32_t* coarse = affine ? c->d_tile_ctr.p + RTS_OFF_COARSE : nullptr;
    if (bins && !affine && c->order_fused && prev_valid && memcmp(prev_sig, cur_sig, 4 * sizeof(uint64_t)) == 0 && c->order_sum_valid) {
        const RtsTileShape cur2 = shape(cur_sig);
        const RtsHeadRule rule2 = {c->coop_frac, c->coop_big_now, c->coop_mid, c->coop_floor, resident_waves};
        RTS_HIP(c->d_tile_key.reserve(n_tiles_cur)); RTS_HIP(c->d_tile_order.reserve(n_tiles_cur)); RTS_HIP(c->d_xcd.reserve(64));
        unsigned long long* persist = reinterpret_cast<unsigned long long*>(c->d_xcd.p + 32);
        const uint32_t per = (n_tiles_cur + (256u << 10) - 1u) / (256u << 10), fat = blocks_for(n_tiles_cur, 256u * per);      // at most 1 024 blocks
        k_tile_merge_keys<<<fat, 256, 0, st>>>(c->d_tile_cost.p, cur2, c->hist->d.p, n_hist, reinterpret_cast<unsigned long long*>(head), persist, head ? head + 2 : nullptr, rule2, c->d_tile_key.p, bins, per);
        k_tile_scan_scatter<<<fat, 256, 0, st>>>(c->d_tile_key.p, n_tiles_cur, bins, c->d_tile_ctr.p + RTS_OFF_COARSE, c->d_tile_order.p, reinterpret_cast<const unsigned long long*>(head), head ? persist : nullptr, per, c->d_tile_ctr.p + RTS_OFF_LIVE);
        RTS_HIP(hipGetLastError());
        return RTS_OK;
    }
    bool merged = false;
    if (!prev_valid && head) { const RtsTileShape cur0 = shape(cur_sig); k_tile_est_sum<<<blocks_for(n_tiles_cur, 256), 256, 0, st>>>(c->hist->d.p, n_hist, cur0, reinterpret_cast<unsigned long long*>(head)); }
    if (prev_valid) { const RtsTileShape p = shape(prev_sig); if (p.n_tiles) { k_tile_merge<<<blocks_for(p.n_tiles, 256), 256, 0, st>>>(c->d_tile_cost.p, p, c->hist->d.p, n_hist, reinterpret_cast<unsigned long long*>(head), coarse); merged = p.n_tiles == n_tiles_cur; } }
    RTS_HIP(c->d_tile_key.reserve(n_tiles_cur)); RTS_HIP(c->d_tile_key_sorted.reserve(n_tiles_cur)); RTS_HIP(c->d_tile_id.reserve(n_tiles_cur)); RTS_HIP(c->d_tile_order.reserve(n_tiles_cur));
    const RtsTileShape cur = shape(cur_sig);
    const RtsHeadRule rule = {c->coop_frac, c->coop_big_now, c->coop_mid, c->coop_floor, resident_waves};
    // (the bands in force were computed by the previous build's scan from the launch before last; a launch of another shape, or no
    // build yet: equal counts)
    const uint32_t* bnd = affine && c->xcd_bnd_tiles == n_tiles_cur ? c->d_xcd.p + 16 : nullptr;
    k_tile_keys<<<blocks_for(n_tiles_cur, 256), 256, 0, st>>>(c->hist->d.p, n_hist, cur, c->d_tile_key.p, c->d_tile_id.p, reinterpret_cast<const unsigned long long*>(head), head ? head + 2 : nullptr, rule, bins,
                                                              affine ? 1 : 0, bnd);
    if (bins) {
        k_tile_bucket_scan<<<1, RTS_TILE_BUCKETS, 0, st>>>(bins, affine ? c->d_xcd.p : nullptr, coarse, n_tiles_cur, affine ? nullptr : c->d_tile_ctr.p + RTS_OFF_LIVE);      // (the affine keys use other bins)
        if (affine) c->xcd_bnd_tiles = merged ? n_tiles_cur : 0u;               // (bands from a launch of another shape are not used)
        k_tile_bucket_scatter<<<blocks_for(n_tiles_cur, 256), 256, 0, st>>>(c->d_tile_key.p, n_tiles_cur, bins, c->d_tile_order.p);
        RTS_HIP(hipGetLastError());
        if (head && c->order_fused && prev_valid) {                     // the launch's cost sum, kept for the two-launch form of the next build
            RTS_HIP(c->d_xcd.reserve(64));
            RTS_HIP(hipMemcpyAsync(c->d_xcd.p + 32, head, sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
            c->order_sum_valid = true;
        } else if (!head && c->order_fused) { RTS_HIP(c->d_xcd.reserve(64)); RTS_HIP(hipMemsetAsync(c->d_xcd.p + 32, 0, sizeof(unsigned long long), st)); c->order_sum_valid = true; }
        return RTS_OK;
    }
    size_t tmp = 0;
    RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d_tile_key.p, c->d_tile_key_sorted.p, c->d_tile_id.p, c->d_tile_order.p, n_tiles_cur, 0, 32, st));
    RTS_HIP(c->d_sort_tmp.reserve(tmp));
    RTS_HIP(rocprim::radix_sort_pairs(c->d_sort_tmp.p, tmp, c->d_tile_key.p, c->d_tile_key_sorted.p, c->d_tile_id.p, c->d_tile_order.p, n_tiles_cur, 0, 32, st));
    return RTS_OK;
}

int rts_post_order_and_expand(RtsContext* c)
{
    const uint32_t R = (uint32_t)c->n_recv, D = c->depth;
    if (R == 0) return RTS_OK;
    hipStream_t st = c->stream;
    RTS_HIP(c->d_ri.reserve(R)); RTS_HIP(c->d_ri_sorted.reserve(R));
    RTS_HIP(c->d_rx_rays.reserve(R)); RTS_HIP(c->d_rx_paths.reserve((size_t)R*D + 1)); RTS_HIP(c->d_rx_angles.reserve((size_t)R*D*2 + 1)); RTS_HIP(c->d_rx_slots.reserve(R));
    size_t tmp = 0;
    if (c->post_small && R <= rts_small_cap_recv(c)) {
        // items per thread by the size of the set (a speculative chain -- count on the device -- is sized for the capacity); sort bits by the largest row
        const uint32_t cap = c->recv_dev ? rts_small_cap_recv(c) : R;
        const uint64_t rows = (uint64_t)c->n_rays * (c->last_args.max_refr != 0 ? 3u : 1u);
        uint32_t bits = 1; while (bits < 40 && ((uint64_t)1 << bits) <= rows) bits++;       // (the padding key, 2^bits - 1, stays above every row)
        if (c->last_args.max_refr == 0) {
            if (cap <= 4u * RTS_SMALL_THREADS) k_recv_order_small<uint32_t, 4><<<1, RTS_SMALL_THREADS, 0, st>>>(c->d_recv.p, R, c->n_rays, 0, bits, c->d_ri_sorted.p, c->recv_dev);
            else if (cap <= 8u * RTS_SMALL_THREADS) k_recv_order_small<uint32_t, 8><<<1, RTS_SMALL_THREADS, 0, st>>>(c->d_recv.p, R, c->n_rays, 0, bits, c->d_ri_sorted.p, c->recv_dev);
            else k_recv_order_small<uint32_t, 16><<<1, RTS_SMALL_THREADS, 0, st>>>(c->d_recv.p, R, c->n_rays, 0, bits, c->d_ri_sorted.p, c->recv_dev);
        } else {
            if (cap <= 4u * RTS_SMALL_THREADS) k_recv_order_small<uint64_t, 4><<<1, RTS_SMALL_THREADS, 0, st>>>(c->d_recv.p, R, c->n_rays, 1, bits, c->d_ri_sorted.p, c->recv_dev);
            else k_recv_order_small<uint64_t, 8><<<1, RTS_SMALL_THREADS, 0, st>>>(c->d_recv.p, R, c->n_rays, 1, bits, c->d_ri_sorted.p, c->recv_dev);
        }
        RTS_STAGE(c, "recv order (one block)");
    } else if (c->last_args.max_refr == 0) {
        RTS_HIP(c->d_rk.reserve(R)); RTS_HIP(c->d_rk_sorted.reserve(R));
        k_recv_keys<<<blocks_for(R, 256), 256, 0, st>>>(c->d_recv.p, c->d_rk.p, c->d_ri.p, R);
        RTS_STAGE(c, "k_recv_keys");
        RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d_rk.p, c->d_rk_sorted.p, c->d_ri.p, c->d_ri_sorted.p, R, 0, 32, st));
        RTS_HIP(c->d_sort_tmp.reserve(tmp));
        RTS_HIP(rocprim::radix_sort_pairs(c->d_sort_tmp.p, tmp, c->d_rk.p, c->d_rk_sorted.p, c->d_ri.p, c->d_ri_sorted.p, R, 0, 32, st));
    } else {
        RTS_HIP(c->d_rk64.reserve(R)); RTS_HIP(c->d_rk64_sorted.reserve(R));
        k_recv_keys64<<<blocks_for(R, 256), 256, 0, st>>>(c->d_recv.p, c->d_rk64.p, c->d_ri.p, R, c->n_rays);
        RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d_rk64.p, c->d_rk64_sorted.p, c->d_ri.p, c->d_ri_sorted.p, R, 0, 34, st));
        RTS_HIP(c->d_sort_tmp.reserve(tmp));
        RTS_HIP(rocprim::radix_sort_pairs(c->d_sort_tmp.p, tmp, c->d_rk64.p, c->d_rk64_sorted.p, c->d_ri.p, c->d_ri_sorted.p, R, 0, 34, st));
    }
    RTS_STAGE(c, "recv sort");
    k_expand<<<blocks_for(R, 256), 256, 0, st>>>(c->last_args, c->d_recv.p, c->d_ri_sorted.p, R, D, c->d_rx_rays.p, c->d_rx_paths.p, c->d_rx_angles.p, c->d_rx_slots.p, c->recv_dev);
    RTS_STAGE(c, "k_expand");
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// --------------------------------------------------------------------------- host mirror (rts_received_prefetch; rts_internal.h: RtsHostMirror)
// The received set of a BASELINE configs[2] pulse is ~2 000 rays x 272 bytes.  Fetched with copy calls it cost the C++ adapter
// three blocking hipMemcpy from pageable memory per pulse -- 0.12 ms, or 0.34 ms in a process whose runtime had not grown its
// staging pool (profiles/r04a_adapter_bench.json: the whole host-tree / device-tree difference of round 3) -- and the aggregation
// another eight uploads and four downloads.  Here kernels store the rows that exist straight into mapped pinned host memory,
// behind the kernels that produce them, and the host reads them after the ONE wait it has anyway.
struct RtsMirrorSegs { uint32_t* dst[4]; const uint32_t* src[4]; uint32_t row_words[4]; };
__global__ void __launch_bounds__(256) k_mirror_rows(const RtsMirrorSegs g, uint32_t R, const unsigned long long* __restrict__ R_dev)
{
    if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; }
    const uint32_t k = blockIdx.y;
    const size_t n = (size_t)R * g.row_words[k];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g.dst[k][i] = g.src[k][i];
}
__global__ void k_mirror_agg(const PerRayData* __restrict__ rays, const double* __restrict__ delay, const double* __restrict__ phase, const int32_t* __restrict__ pm, uint32_t R,
                             double* __restrict__ h_power, double* __restrict__ h_doppler, double* __restrict__ h_delay, double* __restrict__ h_phase, int32_t* __restrict__ h_pm,
                             const unsigned long long* __restrict__ R_dev)
{
    if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    h_power[i] = rays[i].power; h_doppler[i] = rays[i].doppler; h_delay[i] = delay[i]; h_phase[i] = phase[i]; h_pm[i] = pm[i];
}
// ray_tracer.cpp:1219-1253 with the simulator's own RCS / gain callbacks: the host has formed every received ray's power and
// Doppler shift; they replace the traced values before the aggregation
__global__ void k_set_values(PerRayData* __restrict__ rays, const double* __restrict__ power, const double* __restrict__ doppler, uint32_t R)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    rays[i].power = power[i]; rays[i].doppler = doppler[i];
}

int rts_mirror_reserve(RtsContext* c, uint32_t rows)
{
    RtsHostMirror& m = c->mirror;
    const uint32_t D = c->depth;
    if (m.host && m.cap >= rows && m.D == D) return RTS_OK;
    if (m.host) { RTS_HIP(hipStreamSynchronize(c->stream)); (void)hipHostFree(m.host); m.host = nullptr; m.dev = nullptr; m.cap = 0; }
    uint32_t cap = std::max<uint32_t>(rows, RTS_SMALL_CAP32); cap = (cap + 63u) & ~63u;
    size_t o = 0;
    auto take = [&](size_t bytes_per_row) { const size_t at = o; o += ((size_t)cap * bytes_per_row + 255) & ~(size_t)255; return at; };
    m.o_rays = take(sizeof(PerRayData)); m.o_paths = take(4 * (size_t)std::max(D, 1u)); m.o_angles = take(16 * (size_t)std::max(D, 1u)); m.o_slots = take(8);
    m.o_apower = take(8); m.o_adoppler = take(8); m.o_adelay = take(8); m.o_aphase = take(8); m.o_apm = take(4); m.o_vpower = take(8); m.o_vdoppler = take(8);
    RTS_HIP(hipHostMalloc((void**)&m.host, o, hipHostMallocDefault));
    void* dp = nullptr; RTS_HIP(hipHostGetDevicePointer(&dp, m.host, 0));
    m.dev = (char*)dp; m.bytes = o; m.cap = cap; m.D = D; m.recv_valid = false; m.agg_valid = false;
    return RTS_OK;
}

int rts_post_mirror_received(RtsContext* c)
{
    RtsHostMirror& m = c->mirror;
    const uint32_t R = (uint32_t)c->n_recv, D = c->depth;
    m.recv_valid = false; m.agg_valid = false;
    if (R == 0) { m.recv_valid = true; return RTS_OK; }
    if (R > m.cap) return RTS_OK;                                  // (a set beyond the mirror: the views fall back to copies)
    RtsMirrorSegs g;
    g.dst[0] = (uint32_t*)(m.dev + m.o_rays); g.src[0] = (const uint32_t*)c->d_rx_rays.p; g.row_words[0] = sizeof(PerRayData) / 4;
    g.dst[1] = (uint32_t*)(m.dev + m.o_paths); g.src[1] = (const uint32_t*)c->d_rx_paths.p; g.row_words[1] = D;
    g.dst[2] = (uint32_t*)(m.dev + m.o_angles); g.src[2] = (const uint32_t*)c->d_rx_angles.p; g.row_words[2] = 4 * D;
    g.dst[3] = (uint32_t*)(m.dev + m.o_slots); g.src[3] = (const uint32_t*)c->d_rx_slots.p; g.row_words[3] = 2;
    const uint32_t bx = std::min<uint32_t>(blocks_for((size_t)R * (sizeof(PerRayData) / 4), 256), 64u);
    k_mirror_rows<<<dim3(bx, 4), 256, 0, c->stream>>>(g, R, c->recv_dev);
    RTS_HIP(hipGetLastError());
    m.recv_valid = true;
    return RTS_OK;
}

int rts_post_mirror_aggregated(RtsContext* c)
{
    RtsHostMirror& m = c->mirror;
    const uint32_t R = (uint32_t)c->n_recv;
    m.agg_valid = false;
    if (R == 0) { m.agg_valid = true; return RTS_OK; }
    if (!m.host || R > m.cap) return RTS_OK;
    k_mirror_agg<<<blocks_for(R, 256), 256, 0, c->stream>>>(c->d_rx_rays.p, c->d_delay.p, c->d_phase.p, c->d_pathmatch.p, R, (double*)(m.dev + m.o_apower), (double*)(m.dev + m.o_adoppler),
                                                           (double*)(m.dev + m.o_adelay), (double*)(m.dev + m.o_aphase), (int32_t*)(m.dev + m.o_apm), c->recv_dev);
    RTS_HIP(hipGetLastError());
    m.agg_valid = true;
    return RTS_OK;
}

int rts_post_set_values(RtsContext* c, const double* power, const double* doppler)
{
    const uint32_t R = (uint32_t)c->n_recv;
    if (R == 0) return RTS_OK;
    k_set_values<<<blocks_for(R, 256), 256, 0, c->stream>>>(c->d_rx_rays.p, power, doppler, R);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

int rts_post_expand_all(RtsContext* c)
{
    const uint64_t n64 = (uint64_t)c->n_rays * c->last_args.rows; const uint32_t D = c->depth;
    if (n64 == 0) return RTS_OK;
    if (n64 > 0xffffffffULL) { rts_set_error("keep-all buffers: rows * rays exceeds 2^32"); return RTS_ERR_UNSUPPORTED; }
    const uint32_t n = (uint32_t)n64;
    RTS_HIP(c->d_all_rays.reserve(n)); RTS_HIP(c->d_all_paths.reserve((size_t)n*D + 1)); RTS_HIP(c->d_all_angles.reserve((size_t)n*D*2 + 1));
    k_expand<<<blocks_for(n, 256), 256, 0, c->stream>>>(c->last_args, c->d_all.p, nullptr, n, D, c->d_all_rays.p, c->d_all_paths.p, c->d_all_angles.p, nullptr, nullptr);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// --------------------------------------------------------------------------- uniform finalisation, ray_tracer.cpp:1219-1253
__device__ __forceinline__ void finalise_row(PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, const uint32_t i, const uint32_t D, const double* __restrict__ rcs, const uint32_t n_targets,
                                             const double wl, const double gt, const double gr, const double carrier, const double cspeed)
{
    double power = rays[i].power;
    for (uint32_t k = 0; k < D; k++) {
        int targ_k = paths[(size_t)i*D + k];
        if (targ_k >= 0) { double targRCS = ((uint32_t)targ_k < n_targets) ? rcs[targ_k] : 1.0; power *= targRCS; }   // :1225-1229
    }
    power *= (wl*wl*gt*gr);                                                  // :1247
    double Vr = rays[i].doppler/2;                                           // :1252
    rays[i].doppler = carrier*(((1 + Vr/cspeed)/(1 - Vr/cspeed)) - 1);       // :1253
    rays[i].power = power;
}
__global__ void k_finalise(PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, uint32_t R, uint32_t D,
                           const double* __restrict__ rcs, uint32_t n_targets, double wl, double gt, double gr, double carrier, double cspeed, const unsigned long long* __restrict__ R_dev)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; }      // (speculative post-processing: the count is the trace kernel's, on the device; more than the caller's capacity: nothing is done here)
    if (i >= R) return;
    finalise_row(rays, paths, i, D, rcs, n_targets, wl, gt, gr, carrier, cspeed);
}

int rts_post_finalise(RtsContext* c, const double* rcs_host, double wl, double gt, double gr, double carrier, double cspeed)
{
    const uint32_t R = (uint32_t)c->n_recv;
    if (R == 0) return RTS_OK;
    const uint32_t nt = (uint32_t)c->scene->meshes.size();
    RTS_HIP(c->d_rcsval.reserve(nt + 1));
    bool changed = !c->rcs_uploaded;
    for (uint32_t t = 0; t < nt && t < 256; t++) changed = changed || memcmp(&c->pin->rcs[t], &rcs_host[t], sizeof(double)) != 0;
    if (changed) {                                                            // (the same values pulse after pulse: uploaded once)
        RTS_HIP(hipStreamSynchronize(c->stream));                            // an upload of the previous values may still read the staging
        for (uint32_t t = 0; t < nt && t < 256; t++) c->pin->rcs[t] = rcs_host[t];
        if (nt) RTS_HIP(hipMemcpyAsync(c->d_rcsval.p, c->pin->rcs, sizeof(double)*nt, hipMemcpyHostToDevice, c->stream));
        c->rcs_uploaded = true;
    }
    k_finalise<<<blocks_for(R, 256), 256, 0, c->stream>>>(c->d_rx_rays.p, c->d_rx_paths.p, R, c->depth, c->d_rcsval.p, nt, wl, gt, gr, carrier, cspeed, c->recv_dev);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// --------------------------------------------------------------------------- complex return cube
// one lane per received ray, two f64 atomics (global_atomic_add_f64) into cube[rx][pulse][bin]
__device__ __forceinline__ void cube_row(const PerRayData* __restrict__ rays, const uint32_t i, double* __restrict__ cube, const uint32_t n_rx, const uint32_t n_pulses, const uint32_t n_bins,
                                         const uint32_t pulse, const double t0, const double dt, const double cspeed, const double carrier)
{
    const PerRayData r = rays[i];
    if (r.received < 0 || (uint32_t)r.received >= n_rx) return;
    const double delay = (r.rayLength)/cspeed;                               // aggregation.cu:59
    const double phase = -fmod(delay*2*RTS_PI*carrier, 2*RTS_PI);            // aggregation.cu:60
    const double b = floor((delay - t0) / dt);
    if (!(b >= 0.0) || !(b < (double)n_bins)) return;
    const double amp = sqrt(r.power);
    double sn, cs; sincos(phase, &sn, &cs);
    double* cell = cube + 2 * (((size_t)r.received * n_pulses + pulse) * n_bins + (size_t)b);
    atomicAdd(cell, amp * cs); atomicAdd(cell + 1, amp * sn);
}
__global__ void k_cube_accumulate(const PerRayData* __restrict__ rays, uint32_t R, double* __restrict__ cube, uint32_t n_rx, uint32_t n_pulses,
                                  uint32_t n_bins, uint32_t pulse, double t0, double dt, double cspeed, double carrier, const unsigned long long* __restrict__ R_dev)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; }      // (speculative post-processing: the count is the trace kernel's, on the device; more than the caller's capacity: nothing is done here)
    if (i >= R) return;
    cube_row(rays, i, cube, n_rx, n_pulses, n_bins, pulse, t0, dt, cspeed, carrier);
}

int rts_cube_accumulate_device(RtsContext* c, uint32_t pulse_index, double cspeed, double carrier)
{
    const uint32_t R = (uint32_t)c->n_recv;
    if (R == 0) return RTS_OK;
    const RtsCubeParams& q = c->cube_params;
    k_cube_accumulate<<<blocks_for(R, 256), 256, 0, c->stream>>>(c->d_rx_rays.p, R, c->cube, q.n_rx, q.n_pulses, q.n_bins, pulse_index, q.t0, q.dt, cspeed, carrier, c->recv_dev);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// The same product PER UNIQUE PATH: one contribution per response the reference would emit for the pulse
// (ray_tracer.cpp:1290-1321) -- the representative ray of every (receiver, path) group (pathMatch[i] == i) with the GROUP's
// power ((sum sqrt p / n)^2), mean delay and mean phase (aggregation.cu:88-93).  Needs the aggregation of the pulse.
__global__ void k_cube_accumulate_paths(const PerRayData* __restrict__ rays, const double* __restrict__ delay, const double* __restrict__ phase, const int32_t* __restrict__ pm,
                                        uint32_t R, int64_t base, double* __restrict__ cube, uint32_t n_rx, uint32_t n_pulses, uint32_t n_bins, uint32_t pulse, double t0, double dt)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    if ((int64_t)pm[i] != base + (int64_t)i) return;                          // not the representative of its group
    const PerRayData r = rays[i];
    if (r.received < 0 || (uint32_t)r.received >= n_rx) return;
    const double b = floor((delay[i] - t0) / dt);
    if (!(b >= 0.0) || !(b < (double)n_bins)) return;
    const double amp = sqrt(r.power);
    double sn, cs; sincos(phase[i], &sn, &cs);
    double* cell = cube + 2 * (((size_t)r.received * n_pulses + pulse) * n_bins + (size_t)b);
    atomicAdd(cell, amp * cs); atomicAdd(cell + 1, amp * sn);
}

int rts_cube_accumulate_paths_device(RtsContext* c, uint32_t pulse_index, int64_t base)
{
    const uint32_t R = (uint32_t)c->n_recv;
    if (R == 0) return RTS_OK;
    const RtsCubeParams& q = c->cube_params;
    k_cube_accumulate_paths<<<blocks_for(R, 256), 256, 0, c->stream>>>(c->d_rx_rays.p, c->d_delay.p, c->d_phase.p, c->d_pathmatch.p, R, base, c->cube, q.n_rx, q.n_pulses, q.n_bins,
                                                                        pulse_index, q.t0, q.dt);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// --------------------------------------------------------------------------- slow-time (Doppler) transform of the cube
// out[rx][k][bin] = sum_{p < n_pulses} cube[rx][p][bin] e^{-2 pi j k p / N}, N a power of two >= n_pulses (zero padded).
// One block per (receiver, tile of BT consecutive range bins): the tile's N x BT complex128 column set lives in LDS
// (bit-reversed on the way in), log2 N radix-2 stages with a twiddle table in LDS, every stage's N/2 x BT butterflies dealt to
// the block's threads; rows of BT x 16 bytes are contiguous in the cube, so loads and stores are BT-bin segments.  No MFMA: a
// 1024-point f64 transform per (receiver, bin) is 5 120 butterflies; the whole C3 interval (4 x 1024 columns) is ~20 Mflop.
__global__ void __launch_bounds__(256) k_cube_doppler(const double* __restrict__ cube, double* __restrict__ out, uint32_t n_pulses, uint32_t n_bins, uint32_t N, uint32_t logN, uint32_t BT)
{
    extern __shared__ __attribute__((aligned(16))) double s_fft[];               // [N][BT] complex, then [N/2] complex twiddles
    double* x = s_fft; double* tw = s_fft + 2 * (size_t)N * BT;
    const uint32_t rx = blockIdx.y, bin0 = blockIdx.x * BT, t = threadIdx.x;
    for (uint32_t k = t; k < N / 2; k += blockDim.x) { double sn, cs; sincospi(-2.0 * (double)k / (double)N, &sn, &cs); tw[2 * k] = cs; tw[2 * k + 1] = sn; }
    for (uint32_t idx = t; idx < N * BT; idx += blockDim.x) {
        const uint32_t p = idx / BT, b = idx - p * BT, bin = bin0 + b;
        double re = 0.0, im = 0.0;
        if (p < n_pulses && bin < n_bins) { const double* src = cube + 2 * (((size_t)rx * n_pulses + p) * n_bins + bin); re = src[0]; im = src[1]; }
        const uint32_t pr = __brev(p) >> (32u - logN);
        x[2 * ((size_t)pr * BT + b)] = re; x[2 * ((size_t)pr * BT + b) + 1] = im;
    }
    __syncthreads();
    for (uint32_t s = 1; s <= logN; s++) {
        const uint32_t half = 1u << (s - 1), tstep = N >> s;
        for (uint32_t idx = t; idx < (N / 2) * BT; idx += blockDim.x) {
            const uint32_t j = idx / BT, b = idx - j * BT, k = j & (half - 1u), i0 = ((j >> (s - 1)) << s) + k, i1 = i0 + half;
            const double wr = tw[2 * (k * tstep)], wi = tw[2 * (k * tstep) + 1];
            double* a0 = x + 2 * ((size_t)i0 * BT + b); double* a1 = x + 2 * ((size_t)i1 * BT + b);
            const double xr = a1[0], xi = a1[1], tr = wr * xr - wi * xi, ti = wr * xi + wi * xr, ur = a0[0], ui = a0[1];
            a0[0] = ur + tr; a0[1] = ui + ti; a1[0] = ur - tr; a1[1] = ui - ti;
        }
        __syncthreads();
    }
    for (uint32_t idx = t; idx < N * BT; idx += blockDim.x) {
        const uint32_t k = idx / BT, b = idx - k * BT, bin = bin0 + b;
        if (bin < n_bins) { double* dst = out + 2 * (((size_t)rx * N + k) * n_bins + bin); dst[0] = x[2 * ((size_t)k * BT + b)]; dst[1] = x[2 * ((size_t)k * BT + b) + 1]; }
    }
}

int rts_cube_doppler_device(RtsContext* c, uint32_t n_fft, double* out)
{
    const RtsCubeParams& q = c->cube_params;
    uint32_t logN = 0; while ((1u << logN) < n_fft) logN++;
    uint32_t BT = 8; while (BT > 1 && (size_t)n_fft * BT * 16 > 65536) BT >>= 1;
    const size_t lds = (size_t)n_fft * BT * 16 + (size_t)n_fft * 8;
    RTS_HIP(hipFuncSetAttribute((const void*)k_cube_doppler, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((q.n_bins + BT - 1) / BT, q.n_rx);
    k_cube_doppler<<<grid, 256, lds, c->stream>>>(c->cube, out, q.n_pulses, q.n_bins, n_fft, logN, BT);
    RTS_HIP(hipGetLastError());
    return RTS_OK;
}

// --------------------------------------------------------------------------- aggregation (group-by)
// key = rx << (D*B) | sum_k (path[k] + 1) << (k*B): equal keys <=> same receiver and identical
// path row, which is the row_equal test of myKernel1 (aggregation.cu:46-53).
__global__ void k_agg_keys(const PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, uint32_t R, uint32_t D, uint32_t B,
                           uint64_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    uint64_t k = 0;
    for (uint32_t c = 0; c < D; c++) k |= (uint64_t)(uint32_t)(paths[(size_t)i*D + c] + 1) << (c*B);
    k |= (uint64_t)(uint32_t)rays[i].received << (D*B);
    keys[i] = k; idx[i] = i;
}

// Wide keys (D*B + RXB > 64 bits, e.g. 16 bounces in a scene of 100 targets): the same key as a 128- or 192-bit integer,
// sorted by least-significant-word-first passes of the stable 64-bit radix sort.  Word w of the key of ray idx[j]:
__global__ void k_agg_keys_wide(const PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, const uint32_t* __restrict__ idx_in, uint32_t R,
                                uint32_t D, uint32_t B, uint32_t w, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx_out)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    const uint32_t i = idx_in ? idx_in[j] : j;
    uint64_t words[4] = {0, 0, 0, 0};
    auto put = [&](uint64_t v, uint32_t pos) {
        const uint32_t wd = pos >> 6, off = pos & 63u;
        if (wd < 4) words[wd] |= v << off;
        if (off && wd + 1 < 4) words[wd + 1] |= v >> (64u - off);
    };
    for (uint32_t c = 0; c < D; c++) put((uint64_t)(uint32_t)(paths[(size_t)i*D + c] + 1), c * B);
    put((uint64_t)(uint32_t)rays[i].received, D * B);
    keys[j] = words[w]; idx_out[j] = i;
}

// head flags of the sorted order from the rays themselves (same receiver and identical path row: aggregation.cu:46-53), and
// a 64-bit surrogate key that keeps what the later stages read from a key: the receiver, in the top 32 bits
__global__ void k_agg_heads_wide(const PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, const uint32_t* __restrict__ idx_sorted, uint32_t R,
                                 uint32_t D, uint32_t* __restrict__ head, uint64_t* __restrict__ keys_sorted)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= R) return;
    const uint32_t i = idx_sorted[j];
    bool h = (j == 0);
    if (!h) {
        const uint32_t p = idx_sorted[j - 1];
        h = rays[i].received != rays[p].received;
        for (uint32_t c = 0; c < D && !h; c++) h = paths[(size_t)i*D + c] != paths[(size_t)p*D + c];
    }
    head[j] = h ? 1u : 0u;
    keys_sorted[j] = (uint64_t)(uint32_t)rays[i].received << 32;
}

// path rows of the groups' first rays (wide keys: the host cannot decode the path from a 64-bit key)
__global__ void k_agg_gather_paths(const int32_t* __restrict__ paths, const uint32_t* __restrict__ gmin, const uint32_t* __restrict__ g_count, uint32_t D,
                                   int32_t* __restrict__ gpath)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= *g_count) return;
    for (uint32_t c = 0; c < D; c++) gpath[(size_t)g*D + c] = paths[(size_t)gmin[g]*D + c];
}

__global__ void k_agg_heads(const uint64_t* __restrict__ keys, uint32_t* __restrict__ head, uint32_t R)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) head[i] = (i == 0 || keys[i] != keys[i-1]) ? 1u : 0u;
}

// group start positions: gstart[gid] = i for head elements; gstart[G] = R
__global__ void k_agg_starts(const uint32_t* __restrict__ head, const uint32_t* __restrict__ gid_incl, uint32_t* __restrict__ gstart, uint32_t R)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R && head[i]) gstart[gid_incl[i] - 1] = i;
    if (i == R - 1) gstart[gid_incl[i]] = R;
}

#define AGG_TILE 256
// Per-ray contributions (aggregation.cu:59-65) summed per group with a FIXED reduction shape
// (segmented Hillis-Steele scan inside 256-element tiles, then tile partials in tile order), so
// the f64 sums are reproducible run to run.  vals: 5 doubles {n, sqrt(power), delay, phase, doppler}.
#define RTS_AGG_TILE_LDS (2 * 5 * AGG_TILE * 8 + AGG_TILE * 4)      // bytes of LDS one tile's reduction takes
// (raw: RTS_AGG_TILE_LDS bytes of LDS, 8-byte aligned; called by all AGG_TILE threads of the block; ends with the block in step only
// if the caller synchronises before `raw` is reused)
__device__ __forceinline__ void agg_tile_block(unsigned char* __restrict__ raw, const uint32_t tile, const PerRayData* __restrict__ rays, const uint32_t* __restrict__ idx_sorted,
        const uint32_t* __restrict__ gid_incl, const uint32_t* __restrict__ gstart, const uint32_t R, const double cspeed, const double carrier,
        double* __restrict__ gsum, double* __restrict__ tile_first, double* __restrict__ tile_last)
{
    double (*s_v)[5][AGG_TILE] = reinterpret_cast<double (*)[5][AGG_TILE]>(raw);
    uint32_t* s_g = reinterpret_cast<uint32_t*>(raw + 2 * 5 * AGG_TILE * 8);
    const uint32_t t = threadIdx.x, i = tile * AGG_TILE + t;
    const bool valid = i < R;
    uint32_t g = 0xffffffffu;
    double v[5] = {0, 0, 0, 0, 0};
    if (valid) {
        const PerRayData r = rays[idx_sorted[i]];
        g = gid_incl[i] - 1;
        const double delay = (r.rayLength)/cspeed;                            // aggregation.cu:59
        const double phase = -fmod(delay*2*RTS_PI*carrier, 2*RTS_PI);         // :60
        v[0] = 1; v[1] = sqrt(r.power); v[2] = delay; v[3] = phase; v[4] = r.doppler;
    }
    s_g[t] = g;
    for (int k = 0; k < 5; k++) s_v[0][k][t] = v[k];
    __syncthreads();
    int cur = 0;
    for (uint32_t off = 1; off < AGG_TILE; off <<= 1) {
        const bool take = (t >= off) && (s_g[t - off] == g);
        for (int k = 0; k < 5; k++) {
            double x = s_v[cur][k][t];
            if (take) x = s_v[cur][k][t - off] + x;
            s_v[cur ^ 1][k][t] = x;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (!valid) return;
    const bool run_end = (t == AGG_TILE - 1) || (i == R - 1) || (s_g[t + 1] != g);
    if (!run_end) return;
    const uint32_t gs = gstart[g], ge = gstart[g + 1];
    const uint32_t tile_lo = tile * AGG_TILE, tile_hi = tile_lo + AGG_TILE;
    double* dst;
    if (gs >= tile_lo && ge <= tile_hi) dst = gsum + 5*(size_t)g;            // group lies inside this tile: final
    else if (gs < tile_lo) dst = tile_first + 5*(size_t)tile;                // run continues from the previous tile
    else dst = tile_last + 5*(size_t)tile;                                   // run continues into the next tile
    for (int k = 0; k < 5; k++) dst[k] = s_v[cur][k][t];
}
__global__ void __launch_bounds__(AGG_TILE) k_agg_tiles(const PerRayData* __restrict__ rays, const uint32_t* __restrict__ idx_sorted,
        const uint32_t* __restrict__ gid_incl, const uint32_t* __restrict__ gstart, uint32_t R, double cspeed, double carrier,
        double* __restrict__ gsum, double* __restrict__ tile_first, double* __restrict__ tile_last, const unsigned long long* __restrict__ R_dev)
{
    if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; }      // (speculative post-processing: the count is the trace kernel's, on the device; more than the caller's capacity: nothing is done here)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RTS_AGG_TILE_LDS];
    agg_tile_block(s_raw, blockIdx.x, rays, idx_sorted, gid_incl, gstart, R, cspeed, carrier, gsum, tile_first, tile_last);
}

// groups spanning several tiles: partials added in tile order by one wave, fixed shape.  One wave per
// TILE (the tile in which such a group starts does the work), so the launch needs no group count.
__device__ __forceinline__ void agg_span_body(const uint32_t T0, const uint32_t lane, const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gid_incl, uint32_t R,
                                              const double* __restrict__ tile_first, const double* __restrict__ tile_last, double* __restrict__ gsum)
{
    const uint32_t tile_lo = T0 * AGG_TILE;
    if (tile_lo >= R) return;
    const uint32_t tile_hi = min(tile_lo + AGG_TILE, R);
    const uint32_t g = gid_incl[tile_hi - 1] - 1;                 // the run that touches the end of this tile
    const uint32_t gs = gstart[g], ge = gstart[g + 1];
    if (gs < tile_lo || ge <= tile_lo + AGG_TILE) return;         // started earlier, or ends inside this tile
    const uint32_t T1 = (ge - 1) / AGG_TILE;
    double acc[5] = {0, 0, 0, 0, 0};
    // pieces in order: piece 0 = the run that starts in T0 (stored as that tile's "last" partial),
    // pieces 1.. = tiles T0+1 .. T1 (each that tile's "first" partial)
    const uint32_t npieces = T1 - T0 + 1;
    for (uint32_t p = lane; p < npieces; p += 64) {
        const double* src = (p == 0) ? tile_last + 5*(size_t)T0 : tile_first + 5*(size_t)(T0 + p);
        for (int k = 0; k < 5; k++) acc[k] += src[k];
    }
    for (int off = 32; off > 0; off >>= 1) for (int k = 0; k < 5; k++) acc[k] += __shfl_down(acc[k], off);
    if (lane == 0) for (int k = 0; k < 5; k++) gsum[5*(size_t)g + k] = acc[k];
}
__global__ void __launch_bounds__(64) k_agg_span(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gid_incl, uint32_t R,
                                                  const double* __restrict__ tile_first, const double* __restrict__ tile_last, double* __restrict__ gsum)
{
    agg_span_body(blockIdx.x, threadIdx.x, gstart, gid_incl, R, tile_first, tile_last, gsum);
}

__device__ __forceinline__ void agg_groupinfo_body(const uint32_t g, const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ idx_sorted, const uint64_t* __restrict__ keys_sorted,
                                                   const uint32_t* __restrict__ gid_incl, uint32_t R, uint32_t* __restrict__ gmin, uint64_t* __restrict__ gkey, uint32_t* __restrict__ g_out,
                                                   const uint64_t* __restrict__ rows, uint64_t* __restrict__ grow)
{
    const uint32_t G = gid_incl[R - 1];
    if (g == 0) *g_out = G;
    if (g >= G) return;
    gmin[g] = idx_sorted[gstart[g]];          // stable sort: first element of the run has the smallest ray index
    gkey[g] = keys_sorted[gstart[g]];
    if (rows) grow[g] = rows[gmin[g]];        // its global buffer row (order-isomorphic to the received index, comparable across ranks)
}
__global__ void k_agg_groupinfo(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ idx_sorted, const uint64_t* __restrict__ keys_sorted,
                                const uint32_t* __restrict__ gid_incl, uint32_t R, uint32_t* __restrict__ gmin, uint64_t* __restrict__ gkey, uint32_t* __restrict__ g_out,
                                const uint64_t* __restrict__ rows, uint64_t* __restrict__ grow)
{
    agg_groupinfo_body(blockIdx.x * blockDim.x + threadIdx.x, gstart, idx_sorted, keys_sorted, gid_incl, R, gmin, gkey, g_out, rows, grow);
}

// Per-receiver totals for the direct-ray rule (aggregation.cu:56): groups are sorted by key with the
// receiver in the top bits, so a receiver's groups are one contiguous run; one wave per receiver adds
// them with a fixed shape.  rxtot[rx][5], rxmin[rx].
__device__ __forceinline__ void agg_rxtot_body(const uint32_t rx, const uint32_t lane, const uint64_t* __restrict__ gkey, const double* __restrict__ gsum, const uint32_t* __restrict__ gmin,
                                               const uint32_t* __restrict__ g_count, uint32_t shift, double* __restrict__ rxtot, uint32_t* __restrict__ rxmin)
{
    const uint32_t G = *g_count;
    auto lower = [&](uint64_t rxv) {            // first group whose receiver field is >= rxv
        uint32_t lo = 0, hi = G;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; uint64_t r = (shift >= 64) ? 0 : (gkey[mid] >> shift); if (r < rxv) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const uint32_t glo = lower(rx), ghi = lower((uint64_t)rx + 1);
    double acc[5] = {0, 0, 0, 0, 0}; uint32_t mn = 0xffffffffu;
    for (uint32_t g = glo + lane; g < ghi; g += 64) { for (int k = 0; k < 5; k++) acc[k] += gsum[5*(size_t)g + k]; mn = min(mn, gmin[g]); }
    for (int off = 32; off > 0; off >>= 1) { for (int k = 0; k < 5; k++) acc[k] += __shfl_down(acc[k], off); mn = min(mn, (uint32_t)__shfl_down((int)mn, off)); }
    if (lane == 0) { for (int k = 0; k < 5; k++) rxtot[5*(size_t)rx + k] = acc[k]; rxmin[rx] = mn; }
}
__global__ void __launch_bounds__(64) k_agg_rxtot(const uint64_t* __restrict__ gkey, const double* __restrict__ gsum, const uint32_t* __restrict__ gmin,
                                                   const uint32_t* __restrict__ g_count, uint32_t shift, double* __restrict__ rxtot, uint32_t* __restrict__ rxmin)
{
    agg_rxtot_body(blockIdx.x, threadIdx.x, gkey, gsum, gmin, g_count, shift, rxtot, rxmin);
}

// myKernel1's per-ray totals + myKernel2 (aggregation.cu:56-69, 88-93), scattered back per ray.
// rxtot: per receiver {n, sum sqrt p, sum delay, sum phase, sum doppler}, rxmin: smallest ray index.
__device__ __forceinline__ void agg_scatter_body(const uint32_t i, PerRayData* __restrict__ rays, const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ gid_incl,
                              const double* __restrict__ gsum, const uint32_t* __restrict__ gmin, const double* __restrict__ rxtot,
                              const uint32_t* __restrict__ rxmin, uint32_t n_rx_tab, uint32_t R, int64_t base,
                              const double* __restrict__ npath0, const double* __restrict__ power0, const double* __restrict__ doppler0,
                              double* __restrict__ delay, double* __restrict__ phase, int32_t* __restrict__ pm, int32_t pm_init_const, int use_pm_in, int dly_in)
{
    if (i >= R) return;
    const uint32_t r = idx_sorted[i], g = gid_incl[i] - 1;
    PerRayData ray = rays[r];
    const bool direct = (ray.reflDepth == 0) && (ray.refrDepth == 0);
    const double* s; uint32_t mn;
    const uint32_t rx = (uint32_t)ray.received;
    if (direct && rx < n_rx_tab) { s = rxtot + 5*(size_t)rx; mn = rxmin[rx]; } else { s = gsum + 5*(size_t)g; mn = gmin[g]; }
    const double npath = (npath0 ? npath0[r] : 0.0) + s[0];
    const double psum = (power0 ? power0[r] : 0.0) + s[1];
    const double dsum = (dly_in ? delay[r] : 0.0) + s[2];                     // (dly_in = 0: the caller's delay / phase sums start at zero -- no fill of the two arrays)
    const double phsum = (dly_in ? phase[r] : 0.0) + s[3];
    const double dopsum = (doppler0 ? doppler0[r] : 0.0) + s[4];
    double dly = dsum, ph = phsum;
    if (npath > 0) {                                                          // myKernel2
        const double v = psum/npath;
        ray.power = v*v;                                                      // pow(x, 2)
        dly = dsum/npath; ph = phsum/npath;
        ray.doppler = dopsum/npath;
        rays[r].power = ray.power; rays[r].doppler = ray.doppler;
    }
    delay[r] = dly; phase[r] = ph;
    const int64_t m = base + (int64_t)mn;
    const int32_t prev = use_pm_in ? pm[r] : pm_init_const;
    pm[r] = (m < (int64_t)prev) ? (int32_t)m : prev;                          // if (r < d_pathMatch[i]) d_pathMatch[i] = r
}
__global__ void k_agg_scatter(PerRayData* __restrict__ rays, const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ gid_incl,
                              const double* __restrict__ gsum, const uint32_t* __restrict__ gmin, const double* __restrict__ rxtot,
                              const uint32_t* __restrict__ rxmin, uint32_t n_rx_tab, uint32_t R, int64_t base,
                              const double* __restrict__ npath0, const double* __restrict__ power0, const double* __restrict__ doppler0,
                              double* __restrict__ delay, double* __restrict__ phase, int32_t* __restrict__ pm, int32_t pm_init_const, int use_pm_in, int dly_in)
{
    agg_scatter_body(blockIdx.x * blockDim.x + threadIdx.x, rays, idx_sorted, gid_incl, gsum, gmin, rxtot, rxmin, n_rx_tab, R, base, npath0, power0, doppler0, delay, phase, pm, pm_init_const, use_pm_in, dly_in);
}

// the group count and the first `spec` groups of the table, into the handle's pinned host block (device addresses of its members)
__device__ __forceinline__ void agg_export_body(const uint32_t i, const uint32_t* __restrict__ d_G, const double* __restrict__ gsum, const uint32_t* __restrict__ gmin, const uint64_t* __restrict__ gkey, const uint64_t* __restrict__ grow,
                                                uint32_t spec, uint32_t* __restrict__ h_G, double* __restrict__ h_gsum, uint32_t* __restrict__ h_gmin, uint64_t* __restrict__ h_gkey, uint64_t* __restrict__ h_grow)
{
    const uint32_t G = *d_G;
    if (i == 0) *h_G = G;
    if (i >= spec || i >= G) return;
    for (int k = 0; k < 5; k++) h_gsum[5 * (size_t)i + k] = gsum[5 * (size_t)i + k];
    h_gmin[i] = gmin[i]; h_gkey[i] = gkey[i];
    if (grow) h_grow[i] = grow[i];
}
__global__ void k_agg_export(const uint32_t* __restrict__ d_G, const double* __restrict__ gsum, const uint32_t* __restrict__ gmin, const uint64_t* __restrict__ gkey, const uint64_t* __restrict__ grow,
                             uint32_t spec, uint32_t* __restrict__ h_G, double* __restrict__ h_gsum, uint32_t* __restrict__ h_gmin, uint64_t* __restrict__ h_gkey, uint64_t* __restrict__ h_grow)
{
    agg_export_body(blockIdx.x * blockDim.x + threadIdx.x, d_G, gsum, gmin, gkey, grow, spec, h_G, h_gsum, h_gmin, h_gkey, h_grow);
}

// ---------------------------------------------------------------------------- small received sets: one block instead of a chain
// A BASELINE configs[2] pulse receives ~2 000 rays.  Sorted, scanned and reduced by library calls sized for millions of
// elements that is ~27 launches of a few microseconds each -- which, among the blocks of the neighbouring pulses' trace
// kernels, each wait their turn: 0.35 ms of dependent launches per pulse, in the submitting thread's loop.  Up to
// RTS_SMALL_CAP32 / RTS_SMALL_CAP64 rays one block does the ordering (keys, a block-wide radix sort of (key, index) pairs -- stable, as the
// device-wide one --, head flags, scan, group starts) and one block the tail of the aggregation
// (spans, group table, receiver totals, the scatter back to the rays, the export): the SAME statements as the kernels above
// (their bodies are shared), the tile sums in between unchanged, so every sum is the same bits whichever path ran.
// (one block of 256 threads and < 40 KB of LDS: it has to fit the block slot a trace launch leaves free on a CU -- a first
// version with 1 024 threads and 56 KB waited for a CU to drain and made the pulse slower, 0.72 against 0.63 ms)

// received rays in ascending buffer row (k_recv_keys / k_recv_keys64 + the sort): perm[j] = record of output row j
// (raw: LDS for the block sort's storage, 16-byte aligned, at least sizeof(block_radix_sort<K, 256, ITEMS, uint32_t>::storage_type))
template <typename K, int ITEMS>
__device__ __forceinline__ void recv_order_block(unsigned char* __restrict__ raw, const RtsEndRecord* __restrict__ rec, const uint32_t n, const uint32_t n_rays, const int with_chain, const uint32_t bits, uint32_t* __restrict__ perm)
{
    typedef rocprim::block_radix_sort<K, RTS_SMALL_THREADS, ITEMS, uint32_t> Sort;
    typename Sort::storage_type& s_sort = *reinterpret_cast<typename Sort::storage_type*>(raw);
    K k[ITEMS]; uint32_t v[ITEMS];
    for (uint32_t j = 0; j < ITEMS; j++) {
        const uint32_t i = threadIdx.x * ITEMS + j;
        const unsigned long long key = i < n ? (with_chain ? (unsigned long long)(rec[i].pad & 3u) * n_rays + rec[i].slot : (unsigned long long)rec[i].slot) : (1ULL << bits) - 1ULL;      // (padding sorts last: no key is that large)
        k[j] = (K)key; v[j] = i;
    }
    Sort().sort(k, v, s_sort, 0, bits);
    for (uint32_t j = 0; j < ITEMS; j++) { const uint32_t i = threadIdx.x * ITEMS + j; if (i < n) perm[i] = v[j]; }
}
template <typename K, int ITEMS>
__global__ void __launch_bounds__(RTS_SMALL_THREADS) k_recv_order_small(const RtsEndRecord* __restrict__ rec, uint32_t n, uint32_t n_rays, int with_chain, uint32_t bits, uint32_t* __restrict__ perm, const unsigned long long* __restrict__ R_dev)
{
    typedef rocprim::block_radix_sort<K, RTS_SMALL_THREADS, ITEMS, uint32_t> Sort;
    { uint32_t R = n; bool skip = false; if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) skip = true; else R = (uint32_t)v_; } if (skip) return; n = R; }
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[sizeof(typename Sort::storage_type)];
    recv_order_block<K, ITEMS>(s_raw, rec, n, n_rays, with_chain, bits, perm);
}

// k_agg_keys + sort + k_agg_heads + inclusive scan + k_agg_starts
template <typename K, int ITEMS> struct RtsAggOrderLds {
    typedef rocprim::block_radix_sort<K, RTS_SMALL_THREADS, ITEMS, uint32_t> Sort;
    typename Sort::storage_type sort; K last[RTS_SMALL_THREADS]; uint32_t sum[2][RTS_SMALL_THREADS];
};
template <typename K, int ITEMS>
__device__ __forceinline__ void agg_order_block(unsigned char* __restrict__ raw, const PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, const uint32_t R, const uint32_t D, const uint32_t B, const uint32_t key_bits,
                                                uint64_t* __restrict__ keys_sorted, uint32_t* __restrict__ idx_sorted, uint32_t* __restrict__ head, uint32_t* __restrict__ gid_incl, uint32_t* __restrict__ gstart)
{
    typedef rocprim::block_radix_sort<K, RTS_SMALL_THREADS, ITEMS, uint32_t> Sort;
    RtsAggOrderLds<K, ITEMS>& L = *reinterpret_cast<RtsAggOrderLds<K, ITEMS>*>(raw);
    typename Sort::storage_type& s_sort = L.sort; K* s_last = L.last; uint32_t (*s_sum)[RTS_SMALL_THREADS] = L.sum;
    K k[ITEMS]; uint32_t v[ITEMS];
    const uint32_t i0 = threadIdx.x * ITEMS;
    const uint32_t sort_bits = key_bits < 8u * (uint32_t)sizeof(K) ? key_bits + 1u : 8u * (uint32_t)sizeof(K);
    for (uint32_t j = 0; j < ITEMS; j++) {
        const uint32_t i = i0 + j;
        unsigned long long key = key_bits < 64u ? 1ULL << key_bits : ~0ULL;       // (padding: one bit above every real key -- or, with keys that fill the word, equal to the largest at worst: the sort is stable and padding has the larger indices)
        if (i < R) {
            key = 0;
            for (uint32_t c = 0; c < D; c++) key |= (unsigned long long)(uint32_t)(paths[(size_t)i*D + c] + 1) << (c*B);
            key |= (unsigned long long)(uint32_t)rays[i].received << (D*B);
        }
        k[j] = (K)key; v[j] = i;
    }
    // (stable: equal keys keep their index order, as after the radix sort of the general path)
    Sort().sort(k, v, s_sort, 0, sort_bits);
    // head flags and their inclusive scan: thread t owns the sorted elements [t c, (t + 1) c)
    s_last[threadIdx.x] = k[ITEMS - 1];
    __syncthreads();
    K prev = threadIdx.x ? s_last[threadIdx.x - 1] : (K)0;
    uint32_t h[ITEMS], local = 0;
    for (uint32_t j = 0; j < ITEMS; j++) {
        const uint32_t i = i0 + j;
        h[j] = (i < R && (i == 0 || k[j] != prev)) ? 1u : 0u; local += h[j]; prev = k[j];
    }
    s_sum[0][threadIdx.x] = local;
    __syncthreads();
    int cur = 0;
    for (uint32_t off = 1; off < RTS_SMALL_THREADS; off <<= 1) {
        uint32_t x = s_sum[cur][threadIdx.x];
        if (threadIdx.x >= off) x += s_sum[cur][threadIdx.x - off];
        s_sum[cur ^ 1][threadIdx.x] = x; cur ^= 1;
        __syncthreads();
    }
    uint32_t run = s_sum[cur][threadIdx.x] - local;                 // groups that start before this thread's elements
    for (uint32_t j = 0; j < ITEMS; j++) {
        const uint32_t i = i0 + j;
        if (i >= R) break;
        run += h[j];
        keys_sorted[i] = (uint64_t)k[j]; idx_sorted[i] = v[j]; head[i] = h[j]; gid_incl[i] = run;
        if (h[j]) gstart[run - 1] = i;
        if (i == R - 1) gstart[run] = R;
    }
}
template <typename K, int ITEMS>
__global__ void __launch_bounds__(RTS_SMALL_THREADS) k_agg_order_small(const PerRayData* __restrict__ rays, const int32_t* __restrict__ paths, uint32_t R, uint32_t D, uint32_t B, uint32_t key_bits,
                                                                       uint64_t* __restrict__ keys_sorted, uint32_t* __restrict__ idx_sorted, uint32_t* __restrict__ head,
                                                                       uint32_t* __restrict__ gid_incl, uint32_t* __restrict__ gstart, const unsigned long long* __restrict__ R_dev)
{
    if (R_dev) { const unsigned long long v_ = *R_dev; if (v_ > (unsigned long long)R) return; R = (uint32_t)v_; }      // (speculative post-processing: the count is the trace kernel's, on the device; more than the caller's capacity: nothing is done here)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[sizeof(RtsAggOrderLds<K, ITEMS>)];
    agg_order_block<K, ITEMS>(s_raw, rays, paths, R, D, B, key_bits, keys_sorted, idx_sorted, head, gid_incl, gstart);
}

// k_agg_span + k_agg_groupinfo + k_agg_rxtot + k_agg_scatter + k_agg_export (after k_agg_tiles)
struct RtsAggFinish {
    PerRayData* rays; const uint32_t* idx_sorted; const uint64_t* keys_sorted; const uint32_t* gid_incl; const uint32_t* gstart; const double* tile_first; const double* tile_last;
    double* gsum; uint32_t* gmin; uint64_t* gkey; uint32_t* d_G; const uint64_t* rows; uint64_t* grow; uint32_t shift, n_rx_tab; double* rxtot; uint32_t* rxmin; int64_t base;
    const double* npath0; const double* power0; const double* doppler0; double* delay; double* phase; int32_t* pm; int32_t pm_init_const; int use_pm_in, dly_in;
    uint32_t spec; uint32_t* h_G; double* h_gsum; uint32_t* h_gmin; uint64_t* h_gkey; uint64_t* h_grow;
};
// (R >= 1; spec <= R; called by all 256 threads of the block)
__device__ __forceinline__ void agg_finish_block(const RtsAggFinish& q, const uint32_t R, const uint32_t ntiles, const uint32_t spec)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (uint32_t T0 = wave; T0 < ntiles; T0 += n_waves) agg_span_body(T0, lane, q.gstart, q.gid_incl, R, q.tile_first, q.tile_last, q.gsum);
    __syncthreads();
    const uint32_t G = q.gid_incl[R - 1];
    for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) agg_groupinfo_body(g, q.gstart, q.idx_sorted, q.keys_sorted, q.gid_incl, R, q.gmin, q.gkey, q.d_G, q.rows, q.grow);      // (G >= 1: thread 0 publishes the count)
    __syncthreads();
    for (uint32_t rx = wave; rx < q.n_rx_tab; rx += n_waves) agg_rxtot_body(rx, lane, q.gkey, q.gsum, q.gmin, q.d_G, q.shift, q.rxtot, q.rxmin);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < R; i += blockDim.x)
        agg_scatter_body(i, q.rays, q.idx_sorted, q.gid_incl, q.gsum, q.gmin, q.rxtot, q.rxmin, q.n_rx_tab, R, q.base, q.npath0, q.power0, q.doppler0, q.delay, q.phase, q.pm, q.pm_init_const, q.use_pm_in, q.dly_in);
    if (q.h_G) { __syncthreads(); for (uint32_t i = threadIdx.x; i < spec; i += blockDim.x) agg_export_body(i, q.d_G, q.gsum, q.gmin, q.gkey, q.grow, spec, q.h_G, q.h_gsum, q.h_gmin, q.h_gkey, q.h_grow); }      // (spec >= 1)
}
__global__ void __launch_bounds__(256) k_agg_finish_small(const RtsAggFinish q, uint32_t R, uint32_t ntiles, const unsigned long long* __restrict__ R_dev)
{
    uint32_t spec = q.spec;
    if (R_dev) {
        const unsigned long long v_ = *R_dev;
        if (v_ > (unsigned long long)R) return;
        R = (uint32_t)v_; ntiles = (R + AGG_TILE - 1) / AGG_TILE; spec = min(spec, R);
        if (R == 0u) { if (q.h_G && threadIdx.x == 0) *q.h_G = 0u; return; }
    }
    agg_finish_block(q, R, ntiles, spec);
}

// ---------------------------------------------------------------------------- ONE kernel behind the trace (r04)
// A pulse's post-processing for a caller without host callbacks (rts_trace_pulse_end_uniform) and a received set the one-block
// kernels take: order -> expand -> finalise -> cube -> aggregation order -> tile sums -> finish in ONE block of ONE launch instead
// of seven launches of one or a few blocks each.  Every one of the seven waited its turn among the blocks of the neighbouring
// pulses' trace kernels -- 40-350 us apiece in profiles/r03c_pulse_timeline.log for 4-50 us of work -- and the chain was the
// longest dependent sequence behind a trace.  Same bodies as the separate kernels (shared functions above), same bits.  The
// received count is read from the device (the trace kernel's counter): the launch needs no host wait, and sorts as many items per
// thread as the count asks for (the separate kernels are sized on the host, by a hint).
struct RtsPostAll {
    RtsTraceArgs ta; const RtsEndRecord* rec; uint32_t cap, n_rays; int with_chain; uint32_t bits; uint32_t* perm;
    uint32_t D; PerRayData* rays; int32_t* paths; double* angles; uint64_t* slots;
    const double* rcs; uint32_t n_targets; double wl, gt, gr, carrier, cspeed;
    int cube_on; double* cube; uint32_t cube_rx, cube_pulses, cube_bins, cube_pulse; double cube_t0, cube_dt;
    uint32_t B, key_bits; uint32_t* ahead;
    RtsAggFinish fin;
    const unsigned long long* R_dev; uint32_t prio;
};
template <typename K> struct RtsSortMax { enum { ITEMS = sizeof(K) == 4 ? 16 : 8 }; };
#define RTS_POST_ALL_LDS 36864
template <typename KR, typename KA>
__global__ void __launch_bounds__(RTS_SMALL_THREADS) k_post_all(const RtsPostAll q)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RTS_POST_ALL_LDS];
    static_assert(sizeof(RtsAggOrderLds<KA, RtsSortMax<KA>::ITEMS>) <= RTS_POST_ALL_LDS && sizeof(typename rocprim::block_radix_sort<KR, RTS_SMALL_THREADS, RtsSortMax<KR>::ITEMS, uint32_t>::storage_type) <= RTS_POST_ALL_LDS &&
                  RTS_AGG_TILE_LDS <= RTS_POST_ALL_LDS, "LDS of the fused post-processing kernel");
    // (one block that shares its CU with three blocks of trace kernels, on the critical path of its pulse: its four waves take the
    // issue slots first -- RTS_POST_PRIO, default 3)
    if (q.prio) { if (q.prio == 1) __builtin_amdgcn_s_setprio(1); else if (q.prio == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
    const unsigned long long R64 = *q.R_dev;
    if (R64 > (unsigned long long)q.cap) return;                       // more than the buffers were sized for: nothing is done, the host runs the general chain
    const uint32_t R = (uint32_t)R64;
    if (R == 0u) { if (q.fin.h_G && threadIdx.x == 0) *q.fin.h_G = 0u; return; }
    // ---- received rays in ascending buffer row
    if (R <= 4u * RTS_SMALL_THREADS) recv_order_block<KR, 4>(s_raw, q.rec, R, q.n_rays, q.with_chain, q.bits, q.perm);
    else if (R <= 8u * RTS_SMALL_THREADS || RtsSortMax<KR>::ITEMS == 8) recv_order_block<KR, 8>(s_raw, q.rec, R, q.n_rays, q.with_chain, q.bits, q.perm);
    else recv_order_block<KR, RtsSortMax<KR>::ITEMS>(s_raw, q.rec, R, q.n_rays, q.with_chain, q.bits, q.perm);
    __syncthreads();
    // ---- the reference's output records, the uniform finalisation, the return cube: row by row
    for (uint32_t j = threadIdx.x; j < R; j += RTS_SMALL_THREADS) {
        expand_row(q.ta, q.rec, q.perm, j, q.D, q.rays, q.paths, q.angles, q.slots);
        finalise_row(q.rays, q.paths, j, q.D, q.rcs, q.n_targets, q.wl, q.gt, q.gr, q.carrier, q.cspeed);
        if (q.cube_on) cube_row(q.rays, j, q.cube, q.cube_rx, q.cube_pulses, q.cube_bins, q.cube_pulse, q.cube_t0, q.cube_dt, q.cspeed, q.carrier);
    }
    __syncthreads();
    // ---- aggregation: order, tile sums, finish
    if (R <= 4u * RTS_SMALL_THREADS) agg_order_block<KA, 4>(s_raw, q.rays, q.paths, R, q.D, q.B, q.key_bits, const_cast<uint64_t*>(q.fin.keys_sorted), const_cast<uint32_t*>(q.fin.idx_sorted), q.ahead, const_cast<uint32_t*>(q.fin.gid_incl), const_cast<uint32_t*>(q.fin.gstart));
    else if (R <= 8u * RTS_SMALL_THREADS || RtsSortMax<KA>::ITEMS == 8) agg_order_block<KA, 8>(s_raw, q.rays, q.paths, R, q.D, q.B, q.key_bits, const_cast<uint64_t*>(q.fin.keys_sorted), const_cast<uint32_t*>(q.fin.idx_sorted), q.ahead, const_cast<uint32_t*>(q.fin.gid_incl), const_cast<uint32_t*>(q.fin.gstart));
    else agg_order_block<KA, RtsSortMax<KA>::ITEMS>(s_raw, q.rays, q.paths, R, q.D, q.B, q.key_bits, const_cast<uint64_t*>(q.fin.keys_sorted), const_cast<uint32_t*>(q.fin.idx_sorted), q.ahead, const_cast<uint32_t*>(q.fin.gid_incl), const_cast<uint32_t*>(q.fin.gstart));
    __syncthreads();
    const uint32_t ntiles = (R + AGG_TILE - 1) / AGG_TILE;
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        agg_tile_block(s_raw, tile, q.rays, q.fin.idx_sorted, q.fin.gid_incl, q.fin.gstart, R, q.cspeed, q.carrier, q.fin.gsum, const_cast<double*>(q.fin.tile_first), const_cast<double*>(q.fin.tile_last));
        __syncthreads();
    }
    agg_finish_block(q.fin, R, ntiles, min(q.fin.spec, R));
}

// Enqueues k_post_all for the pulse whose trace is in flight or over: buffers sized for `cap` rays, the count from the device.
// The caller has checked: no KEEP_ALL, a (receiver, path) key of <= 64 bits, cap within the one-block sorts.
int rts_post_all_small(RtsContext* c, uint32_t cap, const RtsSpecParams& sp, bool want_groups)
{
    const uint32_t D = c->depth; hipStream_t st = c->stream;
    const uint32_t nt = (uint32_t)c->scene->meshes.size();
    // ---- what rts_post_order_and_expand, rts_post_finalise and rts_aggregate_device reserve, for cap rays
    RTS_HIP(c->d_ri.reserve(cap)); RTS_HIP(c->d_ri_sorted.reserve(cap));
    RTS_HIP(c->d_rx_rays.reserve(cap)); RTS_HIP(c->d_rx_paths.reserve((size_t)cap*D + 1)); RTS_HIP(c->d_rx_angles.reserve((size_t)cap*D*2 + 1)); RTS_HIP(c->d_rx_slots.reserve(cap));
    RTS_HIP(c->d_rcsval.reserve(nt + 1));
    bool changed = !c->rcs_uploaded;
    for (uint32_t t = 0; t < nt && t < 256; t++) changed = changed || memcmp(&c->pin->rcs[t], &sp.rcs[t], sizeof(double)) != 0;
    if (changed) {
        RTS_HIP(hipStreamSynchronize(st));
        for (uint32_t t = 0; t < nt && t < 256; t++) c->pin->rcs[t] = sp.rcs[t];
        if (nt) RTS_HIP(hipMemcpyAsync(c->d_rcsval.p, c->pin->rcs, sizeof(double)*nt, hipMemcpyHostToDevice, st));
        c->rcs_uploaded = true;
    }
    const int32_t max_path = (int32_t)nt - 1, max_rx = c->n_rx ? (int32_t)c->n_rx - 1 : 0;
    uint32_t B = 1; while (((uint64_t)1 << B) < (uint64_t)(max_path + 2)) B++;
    uint32_t RXB = 1; while (((uint64_t)1 << RXB) < (uint64_t)(max_rx + 1)) RXB++;
    if (D == 0) B = 0;
    const uint32_t key_bits = D * B + RXB, shift = D * B, n_rx_tab = (uint32_t)max_rx + 1, ntiles = blocks_for(cap, AGG_TILE);
    const size_t R = cap;
    RTS_HIP(c->d_delay.reserve(R)); RTS_HIP(c->d_phase.reserve(R)); RTS_HIP(c->d_pathmatch.reserve(R));
    RTS_HIP(c->d_akeys.reserve(R)); RTS_HIP(c->d_akeys_sorted.reserve(R)); RTS_HIP(c->d_aidx.reserve(R)); RTS_HIP(c->d_aidx_sorted.reserve(R));
    RTS_HIP(c->d_ghead.reserve(R)); RTS_HIP(c->d_gid.reserve(R));
    RTS_HIP(c->d_gcount.reserve(R + 4)); RTS_HIP(c->d_gsum.reserve(5*(R + 2*(size_t)ntiles) + 16));
    RTS_HIP(c->d_gmin.reserve(R)); RTS_HIP(c->d_gkey.reserve(R));
    const bool use_rows = sp.base == RTS_BASE_USE_ROWS;
    if (use_rows) RTS_HIP(c->d_grow.reserve(R));
    RTS_HIP(c->d_rcs.reserve(5*(size_t)n_rx_tab + n_rx_tab + 8));
    uint32_t* gstart = c->d_gcount.p; uint32_t* d_G = c->d_gcount.p + R + 2;
    double* gsum = c->d_gsum.p; double* tile_first = gsum + 5*R; double* tile_last = tile_first + 5*(size_t)ntiles;
    double* d_rxtot = c->d_rcs.p; uint32_t* d_rxmin = (uint32_t*)(c->d_rcs.p + 5*(size_t)n_rx_tab);
    RtsPinned* pd = c->pin_dev;
    const uint64_t base = use_rows ? 0 : sp.base;
    RtsPostAll q; memset(&q, 0, sizeof(q));
    q.ta = c->last_args; q.rec = c->d_recv.p; q.cap = cap; q.n_rays = c->n_rays; q.with_chain = c->last_args.max_refr != 0 ? 1 : 0;
    { const uint64_t rows = (uint64_t)c->n_rays * (c->last_args.max_refr != 0 ? 3u : 1u); uint32_t bits = 1; while (bits < 40 && ((uint64_t)1 << bits) <= rows) bits++; q.bits = bits; }
    q.perm = c->d_ri_sorted.p; q.D = D; q.rays = c->d_rx_rays.p; q.paths = c->d_rx_paths.p; q.angles = c->d_rx_angles.p; q.slots = c->d_rx_slots.p;
    q.rcs = c->d_rcsval.p; q.n_targets = nt; q.wl = sp.wl; q.gt = sp.gt; q.gr = sp.gr; q.carrier = sp.carrier; q.cspeed = sp.cspeed;
    q.cube_on = sp.cube_pulse >= 0 ? 1 : 0;
    if (q.cube_on) { const RtsCubeParams& cp = c->cube_params; q.cube = c->cube; q.cube_rx = cp.n_rx; q.cube_pulses = cp.n_pulses; q.cube_bins = cp.n_bins; q.cube_pulse = (uint32_t)sp.cube_pulse; q.cube_t0 = cp.t0; q.cube_dt = cp.dt; }
    q.B = B; q.key_bits = key_bits; q.ahead = c->d_ghead.p;
    const uint32_t spec_s = std::min<uint32_t>(cap, RTS_PIN_GROUPS);
    q.fin = RtsAggFinish{c->d_rx_rays.p, c->d_aidx_sorted.p, c->d_akeys_sorted.p, c->d_gid.p, gstart, tile_first, tile_last, gsum, c->d_gmin.p, c->d_gkey.p, d_G,
                         use_rows ? c->d_rx_slots.p : nullptr, use_rows ? c->d_grow.p : nullptr, shift, n_rx_tab, d_rxtot, d_rxmin, (int64_t)base, nullptr, nullptr, nullptr,
                         c->d_delay.p, c->d_phase.p, c->d_pathmatch.p, INT32_MAX, 0, 0,
                         spec_s, want_groups ? &pd->G : nullptr, pd->gsum, pd->gmin, pd->gkey, pd->grow};
    q.R_dev = c->p_counters; q.prio = c->post_prio;
    const bool kr64 = c->last_args.max_refr != 0, ka64 = key_bits >= 32u;
    if (!kr64 && !ka64) k_post_all<uint32_t, uint32_t><<<1, RTS_SMALL_THREADS, 0, st>>>(q);
    else if (!kr64) k_post_all<uint32_t, uint64_t><<<1, RTS_SMALL_THREADS, 0, st>>>(q);
    else if (!ka64) k_post_all<uint64_t, uint32_t><<<1, RTS_SMALL_THREADS, 0, st>>>(q);
    else k_post_all<uint64_t, uint64_t><<<1, RTS_SMALL_THREADS, 0, st>>>(q);
    RTS_HIP(hipGetLastError());
    c->recv_index_base = sp.base; c->agg_base_local = use_rows ? 0 : (int64_t)sp.base;
    RtsAggPending& ap = c->agg_pending;
    ap.valid = true; ap.R = cap; ap.D = D; ap.B = B; ap.shift = shift; ap.wide = false; ap.base = base; ap.rows = use_rows; ap.spec = spec_s; ap.gsum = gsum;
    return RTS_OK;
}

// Aggregates R device-resident rays.  d_delay/d_phase/d_pm are in-out (initial values as the
// caller's h_delay_arr/h_phase_arr/h_pathMatch); d_npath/d_power_sum/d_doppler_sum optional
// initial values (read only).  groups (optional) receives the host copy of the group table.
// max_path / max_rx: largest path entry and receiver index that can occur (decide the key packing).
// One host synchronisation: the group count is consumed on the device, the table is fetched at the end.
int rts_aggregate_device(RtsContext* c, int32_t max_path, int32_t max_rx, const int32_t* d_paths, uint64_t R64, uint32_t D,
                         double cspeed, double carrier, uint64_t base, PerRayData* d_rays, double* d_delay,
                         double* d_phase, int32_t* d_pm, std::vector<RtsGroup>* groups, double* d_npath,
                         double* d_power_sum, double* d_doppler_sum, int32_t pm_init, const uint64_t* d_rows)
{
    if (groups) groups->clear();
    if (R64 == 0) return RTS_OK;
    if (R64 > 0x7fffffffULL) { rts_set_error("aggregate: more than 2^31 received rays"); return RTS_ERR_UNSUPPORTED; }
    const uint32_t R = (uint32_t)R64;
    hipStream_t st = c->stream;
    uint32_t B = 1; while (((uint64_t)1 << B) < (uint64_t)(max_path + 2)) B++;
    uint32_t RXB = 1; while (((uint64_t)1 << RXB) < (uint64_t)(max_rx + 1)) RXB++;
    if (D == 0) B = 0;
    if (D > RTS_MAX_DEPTH || (uint64_t)D * B + RXB > 256) {
        rts_set_error("aggregate: depth %u > %d or a (receiver, path) key of %u x %u + %u bits > 256", D, RTS_MAX_DEPTH, D, B, RXB);
        return RTS_ERR_UNSUPPORTED;
    }
    const uint32_t key_bits = D * B + RXB;
    const bool wide = key_bits > 64;                  // multi-word key: least-significant-word-first passes of the stable sort
    const uint32_t shift = wide ? 32u : D * B;        // where the receiver sits in the 64-bit (surrogate) key of the sorted order
    const uint32_t n_rx_tab = (uint32_t)max_rx + 1;
    const uint32_t ntiles = blocks_for(R, AGG_TILE);
    RTS_HIP(c->d_akeys.reserve(R)); RTS_HIP(c->d_akeys_sorted.reserve(R)); RTS_HIP(c->d_aidx.reserve(R)); RTS_HIP(c->d_aidx_sorted.reserve(R));
    RTS_HIP(c->d_ghead.reserve(R)); RTS_HIP(c->d_gid.reserve(R));
    RTS_HIP(c->d_gcount.reserve((size_t)R + 4)); RTS_HIP(c->d_gsum.reserve(5*((size_t)R + 2*(size_t)ntiles) + 16));
    RTS_HIP(c->d_gmin.reserve(R)); RTS_HIP(c->d_gkey.reserve(R)); if (d_rows) RTS_HIP(c->d_grow.reserve(R));
    RTS_HIP(c->d_rcs.reserve(5*(size_t)n_rx_tab + n_rx_tab + 8));
    uint32_t* gstart = c->d_gcount.p;                 // [<= R + 1]
    uint32_t* d_G = c->d_gcount.p + (size_t)R + 2;    // group count, device resident
    double* gsum = c->d_gsum.p; double* tile_first = gsum + 5*(size_t)R; double* tile_last = tile_first + 5*(size_t)ntiles;
    double* d_rxtot = c->d_rcs.p; uint32_t* d_rxmin = (uint32_t*)(c->d_rcs.p + 5*(size_t)n_rx_tab);
    size_t tmp = 0;
    const bool small = c->post_small && !wide && R <= (key_bits < 32u ? RTS_SMALL_CAP32 : RTS_SMALL_CAP64);       // one block orders, one block finishes (see k_agg_order_small)
    if (small) {
        const uint32_t cap = c->recv_dev ? (key_bits < 32u ? RTS_SMALL_CAP32 : RTS_SMALL_CAP64) : R;
#define RTS_AGG_ORDER(K, I) k_agg_order_small<K, I><<<1, RTS_SMALL_THREADS, 0, st>>>(d_rays, d_paths, R, D, B, key_bits, c->d_akeys_sorted.p, c->d_aidx_sorted.p, c->d_ghead.p, c->d_gid.p, gstart, c->recv_dev)
        if (key_bits < 32u) { if (cap <= 4u * RTS_SMALL_THREADS) RTS_AGG_ORDER(uint32_t, 4); else if (cap <= 8u * RTS_SMALL_THREADS) RTS_AGG_ORDER(uint32_t, 8); else RTS_AGG_ORDER(uint32_t, 16); }
        else { if (cap <= 4u * RTS_SMALL_THREADS) RTS_AGG_ORDER(uint64_t, 4); else RTS_AGG_ORDER(uint64_t, 8); }
#undef RTS_AGG_ORDER
    } else if (!wide) {
        k_agg_keys<<<blocks_for(R, 256), 256, 0, st>>>(d_rays, d_paths, R, D, B, c->d_akeys.p, c->d_aidx.p);
        RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d_akeys.p, c->d_akeys_sorted.p, c->d_aidx.p, c->d_aidx_sorted.p, R, 0, key_bits, st));
        RTS_HIP(c->d_sort_tmp.reserve(tmp));
        RTS_HIP(rocprim::radix_sort_pairs(c->d_sort_tmp.p, tmp, c->d_akeys.p, c->d_akeys_sorted.p, c->d_aidx.p, c->d_aidx_sorted.p, R, 0, key_bits, st));
        k_agg_heads<<<blocks_for(R, 256), 256, 0, st>>>(c->d_akeys_sorted.p, c->d_ghead.p, R);
    } else {
        const uint32_t n_words = (key_bits + 63u) / 64u;
        RTS_HIP(rocprim::radix_sort_pairs(nullptr, tmp, c->d_akeys.p, c->d_akeys_sorted.p, c->d_aidx.p, c->d_aidx_sorted.p, R, 0, 64, st));
        RTS_HIP(c->d_sort_tmp.reserve(tmp));
        RTS_HIP(c->d_gid.reserve(R));                  // (free until the scan: the order handed from one pass to the next)
        for (uint32_t w = 0; w < n_words; w++) {
            // keys of word w in the order left by the passes so far (pass 0: ray order), then a stable sort by that word
            k_agg_keys_wide<<<blocks_for(R, 256), 256, 0, st>>>(d_rays, d_paths, w == 0 ? nullptr : c->d_gid.p, R, D, B, w, c->d_akeys.p, c->d_aidx.p);
            const uint32_t bits = std::min(64u, key_bits - 64u * w);
            size_t t2 = tmp;
            RTS_HIP(rocprim::radix_sort_pairs(c->d_sort_tmp.p, t2, c->d_akeys.p, c->d_akeys_sorted.p, c->d_aidx.p, c->d_aidx_sorted.p, R, 0, bits, st));
            if (w + 1 < n_words) RTS_HIP(hipMemcpyAsync(c->d_gid.p, c->d_aidx_sorted.p, sizeof(uint32_t) * R, hipMemcpyDeviceToDevice, st));
        }
        k_agg_heads_wide<<<blocks_for(R, 256), 256, 0, st>>>(d_rays, d_paths, c->d_aidx_sorted.p, R, D, c->d_ghead.p, c->d_akeys_sorted.p);
    }
    if (!small) {
        RTS_HIP(rocprim::inclusive_scan(nullptr, tmp, c->d_ghead.p, c->d_gid.p, R, rocprim::plus<uint32_t>(), st));
        RTS_HIP(c->d_sort_tmp.reserve(tmp));
        RTS_HIP(rocprim::inclusive_scan(c->d_sort_tmp.p, tmp, c->d_ghead.p, c->d_gid.p, R, rocprim::plus<uint32_t>(), st));
        k_agg_starts<<<blocks_for(R, 256), 256, 0, st>>>(c->d_ghead.p, c->d_gid.p, gstart, R);
    }
    k_agg_tiles<<<ntiles, AGG_TILE, 0, st>>>(d_rays, c->d_aidx_sorted.p, c->d_gid.p, gstart, R, cspeed, carrier, gsum, tile_first, tile_last, c->recv_dev);
    if (small) {
        RtsPinned* pd = c->pin_dev;
        const uint32_t spec_s = std::min<uint32_t>(R, RTS_PIN_GROUPS);
        const RtsAggFinish fq = {d_rays, c->d_aidx_sorted.p, c->d_akeys_sorted.p, c->d_gid.p, gstart, tile_first, tile_last, gsum, c->d_gmin.p, c->d_gkey.p, d_G,
                                 d_rows, d_rows ? c->d_grow.p : nullptr, shift, n_rx_tab, d_rxtot, d_rxmin, (int64_t)base, d_npath, d_power_sum, d_doppler_sum, d_delay, d_phase, d_pm,
                                 pm_init, pm_init == INT32_MIN ? 1 : 0, c->agg_delay_in ? 1 : 0,
                                 spec_s, groups ? &pd->G : nullptr, pd->gsum, pd->gmin, pd->gkey, pd->grow};
        k_agg_finish_small<<<1, 256, 0, st>>>(fq, R, ntiles, c->recv_dev);
        RTS_HIP(hipGetLastError());
        if (!groups) { RTS_HIP(hipStreamSynchronize(st)); return RTS_OK; }
        RtsAggPending& ap = c->agg_pending;
        ap.valid = true; ap.R = R; ap.D = D; ap.B = B; ap.shift = shift; ap.wide = false; ap.base = base; ap.rows = d_rows != nullptr; ap.spec = spec_s; ap.gsum = gsum;
        if (groups != &c->groups) return rts_aggregate_fetch(c, groups);
        return RTS_OK;
    }
    k_agg_span<<<ntiles, 64, 0, st>>>(gstart, c->d_gid.p, R, tile_first, tile_last, gsum);
    k_agg_groupinfo<<<blocks_for(R, 256), 256, 0, st>>>(gstart, c->d_aidx_sorted.p, c->d_akeys_sorted.p, c->d_gid.p, R, c->d_gmin.p, c->d_gkey.p, d_G, d_rows, d_rows ? c->d_grow.p : nullptr);
    k_agg_rxtot<<<n_rx_tab, 64, 0, st>>>(c->d_gkey.p, gsum, c->d_gmin.p, d_G, shift, d_rxtot, d_rxmin);
    k_agg_scatter<<<blocks_for(R, 256), 256, 0, st>>>(d_rays, c->d_aidx_sorted.p, c->d_gid.p, gsum, c->d_gmin.p, d_rxtot, d_rxmin, n_rx_tab, R,
                                                       (int64_t)base, d_npath, d_power_sum, d_doppler_sum, d_delay, d_phase, d_pm, pm_init, pm_init == INT32_MIN ? 1 : 0, c->agg_delay_in ? 1 : 0);
    RTS_HIP(hipGetLastError());
    if (!groups) { RTS_HIP(hipStreamSynchronize(st)); return RTS_OK; }
    if (wide) {                                        // the groups' path rows, for the host copy of the table
        RTS_HIP(c->d_gpath.reserve((size_t)R * D + 1));
        k_agg_gather_paths<<<blocks_for(R, 256), 256, 0, st>>>(d_paths, c->d_gmin.p, d_G, D, c->d_gpath.p);
        RTS_HIP(hipGetLastError());
    }
    // group table to the host: count + the first AGG_SPEC groups speculatively in one batch (pinned), rest on demand
    const uint32_t spec = std::min<uint32_t>(R, RTS_PIN_GROUPS);
    // (written by ONE kernel straight into the pinned block -- five small device-to-host copies per pulse before)
    k_agg_export<<<blocks_for(spec, 256), 256, 0, st>>>(d_G, gsum, c->d_gmin.p, c->d_gkey.p, d_rows ? c->d_grow.p : nullptr, spec, &c->pin_dev->G, c->pin_dev->gsum, c->pin_dev->gmin, c->pin_dev->gkey, c->pin_dev->grow);
    RTS_HIP(hipGetLastError());
    // The table is READ when somebody asks for it (rts_aggregate_fetch): a caller that keeps several pulses in flight enqueues the
    // next pulse while this one's ~16 small kernels wait their turn among the trace kernels' blocks -- the submitting thread used
    // to sit out that chain here, 0.4 of the 0.63 ms of a pipelined BASELINE configs[2] pulse.
    RtsAggPending& ap = c->agg_pending;
    ap.valid = true; ap.R = R; ap.D = D; ap.B = B; ap.shift = shift; ap.wide = wide; ap.base = base; ap.rows = d_rows != nullptr; ap.spec = spec; ap.gsum = gsum;
    if (groups != &c->groups) return rts_aggregate_fetch(c, groups);          // (a caller's own vector: now)
    return RTS_OK;
}

// Second half of rts_aggregate_device: wait for the handle's stream and turn the exported table into RtsGroup records.
int rts_aggregate_fetch(RtsContext* c, std::vector<RtsGroup>* groups)
{
    RtsAggPending& ap = c->agg_pending;
    if (!ap.valid) return RTS_OK;
    ap.valid = false;
    hipStream_t st = c->stream;
    RtsPinned* pin = c->pin;
    const uint32_t R = ap.R, D = ap.D, B = ap.B, shift = ap.shift, spec = ap.spec; const bool wide = ap.wide; const uint64_t base = ap.base;
    const bool d_rows = ap.rows; double* gsum = ap.gsum; (void)R;
    RTS_HIP(rts_stream_wait(c, st));
    const uint32_t G = pin->G;
    const double* h_gsum = pin->gsum; const uint32_t* h_gmin = pin->gmin; const uint64_t* h_gkey = pin->gkey;
    const uint64_t* h_grow = pin->grow; std::vector<uint64_t> v_grow;
    std::vector<double> v_gsum; std::vector<uint32_t> v_gmin; std::vector<uint64_t> v_gkey;
    if (G > spec) {
        if (d_rows) { v_grow.resize(G); RTS_HIP(hipMemcpy(v_grow.data(), c->d_grow.p, sizeof(uint64_t)*G, hipMemcpyDeviceToHost)); h_grow = v_grow.data(); }
        v_gsum.resize(5*(size_t)G); v_gmin.resize(G); v_gkey.resize(G);
        RTS_HIP(hipMemcpy(v_gsum.data(), gsum, sizeof(double)*5*G, hipMemcpyDeviceToHost));
        RTS_HIP(hipMemcpy(v_gmin.data(), c->d_gmin.p, sizeof(uint32_t)*G, hipMemcpyDeviceToHost));
        RTS_HIP(hipMemcpy(v_gkey.data(), c->d_gkey.p, sizeof(uint64_t)*G, hipMemcpyDeviceToHost));
        h_gsum = v_gsum.data(); h_gmin = v_gmin.data(); h_gkey = v_gkey.data();
    }
    const uint64_t pmask = (shift >= 64) ? ~0ULL : (((uint64_t)1 << shift) - 1);
    std::vector<int32_t> v_gpath;
    if (wide && G) { v_gpath.resize((size_t)G * D); RTS_HIP(hipMemcpy(v_gpath.data(), c->d_gpath.p, sizeof(int32_t) * (size_t)G * D, hipMemcpyDeviceToHost)); }
    groups->resize(G);
    for (uint32_t g = 0; g < G; g++) {
        RtsGroup& gr = (*groups)[g]; memset(&gr, 0, sizeof(gr));
        gr.rx = (int32_t)((shift >= 64) ? 0u : (uint32_t)(h_gkey[g] >> shift));
        bool all_neg = true;
        for (uint32_t k = 0; k < RTS_MAX_DEPTH; k++) {
            int v = -1;
            if (k < D) v = wide ? v_gpath[(size_t)g * D + k] : (int)(((h_gkey[g] & pmask) >> (k*B)) & (((uint64_t)1 << B) - 1)) - 1;
            gr.path[k] = v; if (v >= 0) all_neg = false;
        }
        gr.direct = all_neg ? 1u : 0u;
        gr.min_ray = d_rows ? h_grow[g] : base + h_gmin[g];
        gr.n = h_gsum[5*(size_t)g]; gr.sum_sqrt_power = h_gsum[5*(size_t)g + 1]; gr.sum_delay = h_gsum[5*(size_t)g + 2];
        gr.sum_phase = h_gsum[5*(size_t)g + 3]; gr.sum_doppler = h_gsum[5*(size_t)g + 4];
    }
    return RTS_OK;
}
