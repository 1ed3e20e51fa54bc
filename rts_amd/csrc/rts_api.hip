// rts_api.hip -- host side of librts_amd.so: the C-ABI of include/rts_amd.h.
//
// Mirrors the host driver rs::RTS of the reference (ray_tracer.cpp:507-1364) from the point
// where it owns device state: context set-up, per-pulse scene placement, launch, read-back,
// aggregation.  There is no CPU compute path: without a usable HIP device every compute
// entry point fails with RTS_ERR_NO_DEVICE.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include "rts_internal.h"
#include "rts_raygen.h"
#include <atomic>

static thread_local char g_err[1024] = "";
void rts_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }
extern "C" const char* rts_last_error(void) { return g_err; }
#include "rts_build_id.h"
extern "C" const char* rts_build_id(void) { return RTS_SOURCE_HASH; }

// One process (or one group of host threads) per GPU, ON THE GPU'S SOCKET: every kernel launch is a doorbell write and every
// completion a signal read across the fabric otherwise -- on the two-socket hosts of this pool a pipelined BASELINE configs[2]
// pulse took 0.68 ms instead of 0.63 ms when the scheduler happened to start the process on the other socket (the host side
// of rts_trace_pulse_begin alone 0.13 instead of 0.08 ms).  Restricts EVERY thread the process has at the time of the call
// (the runtime's helper threads included; threads created later inherit) to the CPUs of the device's NUMA node, within the
// set the process is allowed already.  *numa_node: the node, -1 when the platform does not say (nothing is changed then).
#include <dirent.h>
#include <sched.h>
extern "C" int rts_bind_host_to_device(int device, int* numa_node)
{
    if (numa_node) *numa_node = -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) { (void)hipGetLastError(); rts_set_error("rts_bind_host_to_device: no PCI address for device %d", device); return RTS_ERR_INVALID; }
    for (char* q = bdf; *q; q++) *q = (char)tolower((unsigned char)*q);
    char path[256]; snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return RTS_OK;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    cpu_set_t want; CPU_ZERO(&want); int n_want = 0;
    if (FILE* f = fopen(path, "r")) {                                  // "64-127,192-255"
        int a = 0, b = 0; char sep = 0;
        while (fscanf(f, "%d", &a) == 1) {
            b = a;
            int ch = fgetc(f);
            if (ch == '-') { if (fscanf(f, "%d", &b) != 1) b = a; ch = fgetc(f); }
            for (int k = a; k <= b && k < CPU_SETSIZE; k++) { CPU_SET(k, &want); n_want++; }
            sep = (char)ch; if (sep != ',') break;
        }
        fclose(f);
    }
    if (n_want == 0) return RTS_OK;
    cpu_set_t have; CPU_ZERO(&have);
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return RTS_OK;
    cpu_set_t both; CPU_AND(&both, &want, &have);
    if (CPU_COUNT(&both) == 0) return RTS_OK;                           // (the process is confined to the other socket: leave it)
    if (DIR* d = opendir("/proc/self/task")) {
        while (struct dirent* e = readdir(d)) { const int tid = atoi(e->d_name); if (tid > 0) (void)sched_setaffinity(tid, sizeof(both), &both); }
        closedir(d);
    } else (void)sched_setaffinity(0, sizeof(both), &both);
    if (numa_node) *numa_node = node;
    return RTS_OK;
}

extern int rts_fill_i32(hipStream_t st, int32_t* p, int32_t v, size_t n);

__global__ void k_children_to_prims(RtsNode4* __restrict__ nodes, uint32_t n, const uint32_t* __restrict__ leaf_prim)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 4; k++) { const int32_t ch = nodes[i].child[k]; nodes[i].pad[k] = ch; if (ch < 0) nodes[i].child[k] = ~(int32_t)leaf_prim[~ch]; }
}

// OCTANT VERSIONS of the node records (round 5; RtsScene::d_nodes4v, [node][octant]).  A ray's octant in a target's frame -- the
// signs of its direction there -- is fixed for a whole walk, and two things the node step used to work out per visit depend on
// nothing else: which plane of a child's slab the ray enters through (low when it runs towards +axis, high towards -axis: the
// role fetch formed six per-lane addresses for that) and a good front-to-back order of the four children (the step sorted the
// four entry distances: 5 compares + 20 selects).  Both are baked here, once per scene: version o of a node holds, per child
// slot, [entry planes x y z | exit planes x y z | child: ~primitive, or 8 x node + o = the child's record of the same octant], the slots in ascending order of the children's position along the
// octant's diagonal (key_mode 1, the default: the box's centre; 0: its entry corner), unused slots (degenerate box, never entered) last.
// The trace kernel reads the record of (node, octant) straight through and visits the open children in slot order.
// Eight times the node bytes (BASELINE configs[2]: 18 -> 145 MB, configs[3]: 185 MB -> 1.5 GB of 288 GB), of which a walk
// touches one version only.
__global__ void k_node_versions(const RtsNode4* __restrict__ nodes, uint32_t n, RtsNode4* __restrict__ out, int key_mode)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (node, octant)
    if (g >= n * 8u) return;
    const uint32_t i = g >> 3, o = g & 7u;
    const RtsNode4 nd = nodes[i];
    const bool mx = (o & 1u) != 0u, my = (o & 2u) != 0u, mz = (o & 4u) != 0u;      // the ray runs towards -x / -y / -z
    float key[4]; int ord[4];
    for (int k = 0; k < 4; k++) {
        const float ex = mx ? -nd.hix[k] : nd.lox[k], ey = my ? -nd.hiy[k] : nd.loy[k], ez = mz ? -nd.hiz[k] : nd.loz[k];      // entry corner, along the diagonal
        const float cx = 0.5f * (nd.lox[k] + nd.hix[k]), cy = 0.5f * (nd.loy[k] + nd.hiy[k]), cz = 0.5f * (nd.loz[k] + nd.hiz[k]);
        const float kc = key_mode == 1 ? ((mx ? -cx : cx) + (my ? -cy : cy) + (mz ? -cz : cz)) : (ex + ey + ez);
        key[k] = nd.child[k] == 0x7fffffff ? __builtin_inff() : kc;
        ord[k] = k;
    }
    for (int a = 1; a < 4; a++)                                     // stable insertion sort of four
        for (int b = a; b > 0 && key[ord[b]] < key[ord[b - 1]]; b--) { const int t = ord[b]; ord[b] = ord[b - 1]; ord[b - 1] = t; }
    RtsNode4 v;
    for (int j = 0; j < 4; j++) {
        const int k = ord[j];
        v.lox[j] = mx ? nd.hix[k] : nd.lox[k]; v.hix[j] = mx ? nd.lox[k] : nd.hix[k];      // lo* <- entry planes, hi* <- exit planes
        v.loy[j] = my ? nd.hiy[k] : nd.loy[k]; v.hiy[j] = my ? nd.loy[k] : nd.hiy[k];
        v.loz[j] = mz ? nd.hiz[k] : nd.loz[k]; v.hiz[j] = mz ? nd.loz[k] : nd.hiz[k];
        const int32_t ch = nd.child[k];
        v.child[j] = (ch >= 0 && ch != 0x7fffffff) ? (int32_t)(((uint32_t)ch << 3) | o) : ch;      // a node child: the index of ITS record of this octant (a walk never changes octant)
        v.pad[j] = nd.pad[k];
    }
    out[g] = v;
}

// The two waits of a pulse (its trace, its post-processing): a thread blocked in hipStreamSynchronize lets its core fall
// asleep, and every wake-up -- two per pulse, each in front of work the GPU is waiting for -- costs tens of microseconds on
// the hosts of this pool (the C++ adapter's loop ran 0.44 ms per pulse right after a second of host-side hierarchy build had
// kept the core awake, 0.72 ms otherwise).  RtsContext::spin_wait (default on; RTS_SPIN_WAIT=0): poll the stream instead.
hipError_t rts_stream_wait(RtsContext* c, hipStream_t st)
{
    if (!c->spin_wait) return hipStreamSynchronize(st);
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        __builtin_ia32_pause();
    }
}
static int rts_attach_scene(RtsContext* c);
void rts_comm_cache_forget(RtsContext* c);

int rts_debug_stage(RtsContext* c, const char* name)
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("RTS_DEBUG_SYNC"); on = (e && e[0] == '1') ? 1 : 0; }
    if (!on) return RTS_OK;
    fprintf(stderr, "[rts] stage %s ... ", name); fflush(stderr);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->tstream);
    if (e != hipSuccess) { fprintf(stderr, "FAILED: %s\n", hipGetErrorString(e)); rts_set_error("stage %s: %s", name, hipGetErrorString(e)); return RTS_ERR_HIP; }
    fprintf(stderr, "ok\n"); fflush(stderr);
    return RTS_OK;
}

extern "C" int rts_device_count(int* n)
{
    if (!n) { rts_set_error("rts_device_count: null output"); return RTS_ERR_INVALID; }
    int k = 0; hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) { *n = 0; rts_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return RTS_ERR_NO_DEVICE; }
    *n = k; return RTS_OK;
}

extern "C" int rts_create(const RtsParams* p, RtsHandle* out)
{
    if (!p || !out) { rts_set_error("rts_create: null argument"); return RTS_ERR_INVALID; }
    *out = nullptr;
    if (p->width == 0) { rts_set_error("rts_create: width must be >= 1"); return RTS_ERR_INVALID; }
    if ((uint64_t)p->width * p->width * p->width > 0xffffffffULL) {      // rayIndex is unsigned int, ray_tracer.cu:151
        rts_set_error("rts_create: W^3 must fit 32 bits (W <= 1625)"); return RTS_ERR_INVALID; }
    const uint32_t refr = p->max_refr > 0 ? 2u : 0u;                       // ">0 is forced to exactly 2", ray_tracer.cpp:604-605
    if (p->max_refl + refr > RTS_MAX_DEPTH) { rts_set_error("rts_create: max_refl + max_refr > %d", RTS_MAX_DEPTH); return RTS_ERR_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { rts_set_error("rts_create: no HIP device (there is no CPU fallback)"); return RTS_ERR_NO_DEVICE; }
    if (p->device < 0 || p->device >= ndev) { rts_set_error("rts_create: device %d out of range (%d devices)", p->device, ndev); return RTS_ERR_INVALID; }
    RTS_HIP(hipSetDevice(p->device));
    RtsContext* c = new RtsContext();
    c->params = *p; c->params.max_refr = refr; c->depth = refr + p->max_refl; c->device = p->device;
    c->scene = new RtsScene(); c->scene->device = p->device;
    memset(&c->stats, 0, sizeof(c->stats));
    // two streams of different priority (=> different hardware queues): the long trace kernel on the low one, the many
    // short build/ordering/aggregation kernels on the high one so that they slot in while a trace kernel is running
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high);
    if (e != hipSuccess) { delete c; rts_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return RTS_ERR_HIP; }
    c->gate = new RtsGate(); c->gate->refs = 1; c->gate->device = p->device;
    e = hipStreamCreateWithPriority(&c->gate->tstream, hipStreamNonBlocking, prio_low);
    if (e != hipSuccess) { delete c->gate; (void)hipStreamDestroy(c->stream); delete c; rts_set_error("hipStreamCreate: %s", hipGetErrorString(e)); return RTS_ERR_HIP; }
    c->tstream = c->gate->tstream; c->tstream_now = c->tstream;
    e = hipEventCreateWithFlags(&c->ev_spec, hipEventDisableTiming); if (e != hipSuccess) { delete c; rts_set_error("hipEventCreate: %s", hipGetErrorString(e)); return RTS_ERR_HIP; }
    for (int i = 0; i < 2; i++) { e = hipEventCreateWithFlags(&c->ev_coop[i], hipEventDisableTiming); if (e != hipSuccess) { delete c; rts_set_error("hipEventCreate: %s", hipGetErrorString(e)); return RTS_ERR_HIP; } }
    for (int i = 0; i < 9; i++) { e = hipEventCreate(&c->ev[i]); if (e != hipSuccess) { delete c; rts_set_error("hipEventCreate: %s", hipGetErrorString(e)); return RTS_ERR_HIP; } }
    e = hipHostMalloc((void**)&c->pin, sizeof(RtsPinned), hipHostMallocDefault);
    if (e != hipSuccess) { delete c; rts_set_error("hipHostMalloc: %s", hipGetErrorString(e)); return RTS_ERR_HIP; }
    memset(c->pin, 0, sizeof(RtsPinned));
    { void* dp = nullptr; e = hipHostGetDevicePointer(&dp, c->pin, 0); if (e != hipSuccess) { delete c; rts_set_error("hipHostGetDevicePointer: %s", hipGetErrorString(e)); return RTS_ERR_HIP; } c->pin_dev = (RtsPinned*)dp; }
    hipDeviceProp_t prop; e = hipGetDeviceProperties(&prop, p->device);
    if (e != hipSuccess) { delete c; rts_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return RTS_ERR_HIP; }
    c->n_cu = prop.multiProcessorCount;
    rts_trace_preload();
    { const char* e = getenv("RTS_PRIMARY_MASK"); if (e && e[0] == '0') c->use_pmask = false; }            // experiments / tests: no primary-ray pre-filter
    if (p->flags & RTS_FLAG_NO_PREFILTER) c->use_pmask = false;
    // experiment / test knobs, read ONCE PER HANDLE at creation (never per process: two handles of one process may differ)
    { const char* e = getenv("RTS_GRID_MULT"); if (e) c->grid_mult = std::max(1, atoi(e)); }
    { const char* e = getenv("RTS_GRID_SPARE"); if (e) { c->grid_spare = std::max(0, atoi(e)); c->grid_spare_forced = true; } }
    { const char* e = getenv("RTS_TILE_LPT"); if (e && e[0] == '0') c->tile_lpt = false; }
    { const char* e = getenv("RTS_COOP_BIG_PART"); if (e) { const double v = atof(e); if (v >= 0) c->coop_big_part = v; } }
    { const char* e = getenv("RTS_COOP_FRAC"); if (e) { const double v = atof(e); if (v >= 0) c->coop_frac = v; } }                  // 0: no cooperative units; tests: tiny values put every tile at the head
    { const char* e = getenv("RTS_COOP_FLOOR"); if (e) c->coop_floor = (uint32_t)std::max(0, atoi(e)); }
    c->debug_coop = getenv("RTS_DEBUG_COOP") != nullptr;
    c->hist = new RtsTileHist();
    { const char* e = getenv("RTS_SHARE_HISTORY"); if (e) c->share_history = e[0] != '0'; }
    { const char* e = getenv("RTS_RX_WINDOW_SCREEN"); if (e) c->rx_window_screen = e[0] != '0'; }
    { const char* e = getenv("RTS_TIMELINE_BLOCKS"); if (e) c->timeline_blocks = e[0] != '0'; }
    { const char* e = getenv("RTS_COOP_VERSIONS"); if (e) c->coop_versions = e[0] != '0'; }
    { const char* e = getenv("RTS_DEAD_BATCH"); if (e) c->batch_dead = strcmp(e, "all") == 0 ? 2 : (e[0] != '0' ? 1 : 0); }      // dead-tile batches of the trace kernel: 0 never, 1 the order's dead part (default), all: every position is screened tile-wise first (tests)
    { const char* e = getenv("RTS_WALK_VERSIONS"); if (e) c->node_versions = e[0] != '0'; }      // (per handle: RTS_NODE_VERSIONS decides whether the scene HAS versions, this whether the handle walks them)
    { const char* e = getenv("RTS_SUM_IN_KERNEL"); if (e) c->sum_in_kernel = atoi(e) != 0; }
    { const char* e = getenv("RTS_SPIN_WAIT"); if (e) c->spin_wait = atoi(e) != 0; }
    { const char* e = getenv("RTS_ORDER_FUSED"); if (e) c->order_fused = atoi(e) != 0; }
    { const char* e = getenv("RTS_PLACE_FUSED"); if (e) c->place_fused = atoi(e) != 0; }
    { const char* e = getenv("RTS_TILE_SORT"); if (e) c->tile_bucket_order = strcmp(e, "radix") != 0; }
    { const char* e = getenv("RTS_XCD_AFFINE"); if (e) c->xcd_affine = e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2); }
    { const char* e = getenv("RTS_TRACE_OWN_STREAM"); if (e) c->trace_own_stream = atoi(e) != 0; }
    { const char* e = getenv("RTS_SPEC_STREAM"); if (e) c->spec_on_trace_stream = strcmp(e, "trace") == 0; }
    { const char* e = getenv("RTS_SPECULATE"); if (e) c->spec_enabled = atoi(e) != 0; }
    { const char* e = getenv("RTS_POST_SMALL"); if (e) c->post_small = atoi(e) != 0; }
    { const char* e = getenv("RTS_POST_ONE"); if (e) c->post_one = atoi(e) != 0; }
    { const char* e = getenv("RTS_POST_ONE_MAX"); if (e) c->post_one_max = (uint64_t)strtoull(e, nullptr, 10); }
    { const char* e = getenv("RTS_POST_PRIO"); if (e) c->post_prio = (uint32_t)std::min(3, std::max(0, atoi(e))); }
    { const char* e = getenv("RTS_ASYNC_IDLE0"); if (e) c->async_idle0 = (uint32_t)std::min(64, std::max(0, atoi(e))); }
    { const char* e = getenv("RTS_ASYNC_IDLE1"); if (e) c->async_idle1 = (uint32_t)std::min(64, std::max(1, atoi(e))); }
    { const char* e = getenv("RTS_ASYNC_AGE"); if (e) c->async_age = (uint32_t)std::max(0, atoi(e)); }
    { const char* e = getenv("RTS_COOP_STEPS"); if (e) c->coop_walk_steps = (uint32_t)std::max(0, atoi(e)); }
    { const char* e = getenv("RTS_COOP_STEPS_LO"); if (e) c->coop_walk_steps_lo = (uint32_t)std::max(0, atoi(e)); }
    { const char* e = getenv("RTS_COOP_MID"); if (e) c->coop_mid = std::max(0.0, atof(e)); }
    { const char* e = getenv("RTS_COOP_SEG"); if (e && atoi(e) == 0) c->coop_walk_steps = 0; }      // (the tests' old switch: every tile above the floor is flagged)
    { const char* e = getenv("RTS_COOP_BIG"); if (e) c->coop_big = std::max(0.0, atof(e)); }
    { const char* e = getenv("RTS_COOP_SPREAD"); if (e) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) c->coop_spread = (uint32_t)v; } }
    { const char* e = getenv("RTS_COOP_GRID"); if (e) c->coop_grid_max = (uint32_t)std::min(4096, std::max(1, atoi(e))); }
    { const char* e = getenv("RTS_EW_REL"); if (e) { const double v = atof(e); if (v > 0) c->ew_rel = v; } }
    { const char* e = getenv("RTS_STACK_LDS_DEBUG"); if (e) { int v = atoi(e); if (v >= 1 && v <= RTS_STACK_LDS) c->stack_lds = (uint32_t)v; } }   // tests: force the spill path
    *out = c;
    return RTS_OK;
}

// RTS_LAP=1: where the host side of rts_trace_pulse_begin spends its time, per section, summed per handle and printed by
// rts_destroy (stderr) -- the tool behind DESIGN.md's "what a slow run is"
static int rts_lap_on() { static int on = -1; if (on < 0) { const char* e = getenv("RTS_LAP"); on = (e && e[0] == '1') ? 1 : 0; } return on; }
struct RtsLapTimer {
    RtsContext* c; std::chrono::steady_clock::time_point t0; bool on;
    explicit RtsLapTimer(RtsContext* c_) : c(c_), on(rts_lap_on() != 0) { if (on) t0 = std::chrono::steady_clock::now(); }
    void lap(int k) { if (!on) return; const auto t1 = std::chrono::steady_clock::now(); c->lap_s[k] += std::chrono::duration<double>(t1 - t0).count(); c->lap_n[k]++; t0 = t1; }
};

// pulses begun and not yet ended, per device: a trace launch that will share the GPU with another pulse's kernels leaves block
// slots free for them (RtsContext::grid_spare), a lone pulse takes the whole chip
static std::atomic<int> g_open_pulses[64];
static std::mutex g_hist_mu;      // (re)allocation of a shared tile-cost history (RtsTileHist)

extern "C" int rts_destroy(RtsHandle c)
{
    if (!c) return RTS_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (rts_lap_on() && c->lap_n[0]) {
        static const char* names[8] = {"host prep", "upload + fill", "scene place", "buffers + tile order", "events + trace launch", "hipSetDevice", "resolve previous", ""};
        fprintf(stderr, "[rts lap] handle %p, %llu pulses begun, us per pulse:", (void*)c, (unsigned long long)c->lap_n[0]);
        for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.1f |", names[k], c->lap_s[k] / (double)c->lap_n[0] * 1e6);
        fprintf(stderr, "\n");
    }
    if (c->pulse_open || c->spec_pending) { c->pulse_open = false; c->spec_pending = false; g_open_pulses[c->device & 63]--; }
    rts_comm_cache_forget(c);
    if (c->scene && --c->scene->refs == 0) { c->scene->release(); delete c->scene; }
    c->scene = nullptr;
    if (c->hist && --c->hist->refs == 0) { c->hist->d.release(); delete c->hist; }
    c->hist = nullptr;
    c->d_verts_world.release(); c->d_normals_world.release();
    c->d_params.release();
    c->d_leaves.release(); c->d_sort_tmp.release(); c->d_rx.release(); c->d_recv.release(); c->d_all.release();
    c->d_block_counters.release(); c->d_timeline.release(); c->d_tile_cost.release(); c->d_tile_key.release(); c->d_tile_key_sorted.release(); c->d_tile_id.release(); c->d_tile_order.release(); c->d_tile_ctr.release(); c->d_dir_hist.release(); c->d_pmask.release(); c->d_child.release(); c->d_rk64.release(); c->d_rk64_sorted.release(); c->d_hit_prim.release(); c->d_hit_t.release(); c->d_stack_ovf.release(); c->d_il_list.release(); c->d_rec_tmp.release();
    c->d_xcd.release();
    c->d_rk.release(); c->d_rk_sorted.release(); c->d_ri.release(); c->d_ri_sorted.release(); c->d_rx_rays.release(); c->d_rx_paths.release();
    c->d_rx_angles.release(); c->d_rx_slots.release(); c->d_all_rays.release(); c->d_all_paths.release(); c->d_all_angles.release();
    c->d_akeys.release(); c->d_akeys_sorted.release(); c->d_aidx.release(); c->d_aidx_sorted.release(); c->d_ghead.release(); c->d_gid.release();
    c->d_gsum.release(); c->d_gmin.release(); c->d_gkey.release(); c->d_gpath.release(); c->d_grow.release(); c->d_gcount.release(); c->d_delay.release(); c->d_phase.release();
    c->d_pathmatch.release(); c->d_rcs.release(); c->d_rcsval.release(); c->d_cube_own.release(); c->d_doppler_own.release();
    (void)hipStreamSynchronize(c->tstream);
    if (--c->gate->refs == 0) { (void)hipStreamDestroy(c->gate->tstream); delete c->gate; }
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_rx) (void)hipHostFree(c->pin_rx);
    if (c->mirror.host) (void)hipHostFree(c->mirror.host);
    for (int i = 0; i < 9; i++) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_spec) (void)hipEventDestroy(c->ev_spec);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(c->ev_coop[i]);
    if (c->cstream) { (void)hipStreamSynchronize(c->cstream); (void)hipStreamDestroy(c->cstream); }
    (void)hipStreamDestroy(c->stream);
    delete c;
    return RTS_OK;
}

#define CHECK_HANDLE(c) do { if (!(c)) { rts_set_error("null handle"); return RTS_ERR_INVALID; } RTS_HIP(hipSetDevice((c)->device)); } while (0)
// entry points that consume a pulse's results complete a pulse that was begun but not yet ended
static int rts_spec_resolve(RtsContext* c);
#define CHECK_CLOSED(c) do { if ((c)->pulse_open) { int rc_ = rts_trace_pulse_end(c); if (rc_ != RTS_OK) return rc_; } \
                             if ((c)->spec_pending) { int rc_ = rts_spec_resolve(c); if (rc_ != RTS_OK) return rc_; } } while (0)

extern "C" int rts_link_handles(RtsHandle a, RtsHandle b)
{
    if (!a || !b || a == b) { rts_set_error("rts_link_handles: needs two distinct handles"); return RTS_ERR_INVALID; }
    if (a->device != b->device) { rts_set_error("rts_link_handles: handles live on different devices (%d, %d)", a->device, b->device); return RTS_ERR_INVALID; }
    if (a->gate == b->gate) return RTS_OK;
    if (a->gate->refs > 1 && b->gate->refs > 1) { rts_set_error("rts_link_handles: both handles already belong to (different) groups"); return RTS_ERR_INVALID; }
    if (a->pulse_open || b->pulse_open) { rts_set_error("rts_link_handles: a pulse is in flight"); return RTS_ERR_INVALID; }
    RtsContext* joiner = b->gate->refs == 1 ? b : a;     // the handle that is still alone adopts the other's trace stream
    RtsContext* host = joiner == b ? a : b;
    RTS_HIP(hipSetDevice(a->device));
    RTS_HIP(hipStreamSynchronize(joiner->tstream));
    (void)hipStreamDestroy(joiner->gate->tstream); delete joiner->gate;
    joiner->gate = host->gate; joiner->gate->refs++; joiner->tstream = joiner->gate->tstream; joiner->tstream_now = joiner->tstream;
    return RTS_OK;
}

// The aggregation groups received rays by a packed (receiver, path) key of D x ceil(log2(targets + 1)) + ceil(log2(receivers))
// bits: one 64-bit sort when it fits, two or three stable passes when it does not (rts_post.hip: wide keys).  A configuration
// beyond 256 bits would be refused HERE, when the scene or the receivers are set, not in the middle of a pulse loop.
static int check_key_width(uint32_t depth, uint32_t n_targets, uint32_t n_rx, const char* who)
{
    uint32_t B = 1; while (((uint64_t)1 << B) < (uint64_t)n_targets + 1) B++;
    uint32_t RXB = 1; while (((uint64_t)1 << RXB) < (uint64_t)std::max<uint32_t>(n_rx, 1)) RXB++;
    if (depth == 0) B = 0;
    if ((uint64_t)depth * B + RXB > 256) {            // (cannot happen within the other limits: 16 x 8 + 16 = 144 bits)
        rts_set_error("%s: %u targets x depth %u (max_refl + max_refr) with %u receivers needs a %u-bit (receiver, path) aggregation key; the limit is 256 bits",
                      who, n_targets, depth, n_rx, depth * B + RXB);
        return RTS_ERR_UNSUPPORTED;
    }
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- scene
extern "C" int rts_set_scene(RtsHandle c, const RtsMesh* meshes, uint32_t n_targets)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    { int rc = check_key_width(c->depth, n_targets, c->n_rx, "rts_set_scene"); if (rc != RTS_OK) return rc; }
    if (n_targets && !meshes) { rts_set_error("rts_set_scene: null meshes"); return RTS_ERR_INVALID; }
    if (n_targets > 254) { rts_set_error("rts_set_scene: more than 254 targets"); return RTS_ERR_UNSUPPORTED; }
    std::vector<RtsMeshHost> mh(n_targets);
    uint64_t nt = 0, nv = 0, nn = 0;
    for (uint32_t t = 0; t < n_targets; t++) {
        const RtsMesh& m = meshes[t];
        if ((m.n_triangles && !m.triangles) || (m.n_vertices && !m.vertices) || (m.n_normals && !m.normals)) { rts_set_error("rts_set_scene: target %u has null arrays", t); return RTS_ERR_INVALID; }
        mh[t].n_tris = m.n_triangles; mh[t].n_verts = m.n_vertices; mh[t].n_normals = m.n_normals;
        mh[t].tri_base = (uint32_t)nt; mh[t].vert_base = (uint32_t)nv; mh[t].normal_base = (uint32_t)nn;
        mh[t].refl_coeff = m.refl_coeff; mh[t].refr_index = m.refr_index;
        mh[t].perface = m.n_normals > m.n_vertices;                              // triangle_mesh.cu:178
        nt += m.n_triangles; nv += m.n_vertices; nn += m.n_normals;
        if (nt > 0x7ffffff0ULL || nv > 0x7ffffff0ULL || nn > 0x7ffffff0ULL) { rts_set_error("rts_set_scene: scene too large"); return RTS_ERR_UNSUPPORTED; }
    }
    std::vector<uint32_t> vidx(3*nt), nidx(3*nt), vtarg(nv), ntarg(nn), ptarg(nt);
    std::vector<double> verts(3*nv), normals(3*nn);
    for (uint32_t t = 0; t < n_targets; t++) {
        const RtsMesh& m = meshes[t]; const RtsMeshHost& h = mh[t];
        for (uint32_t i = 0; i < m.n_triangles; i++) {
            for (int k = 0; k < 3; k++) {
                uint32_t v = m.triangles[3*(size_t)i + k];
                if (v >= m.n_vertices) { rts_set_error("rts_set_scene: target %u triangle %u references vertex %u >= %u", t, i, v, m.n_vertices); return RTS_ERR_INVALID; }
                vidx[3*((size_t)h.tri_base + i) + k] = h.vert_base + v;
                // dbuf_normals is indexed by the vertex index (triangle_mesh.cu:170-172), or by the
                // primitive index for per-face ("rect") normals (:178-180)
                uint32_t ni = h.perface ? i : v;
                if (m.n_normals == 0) ni = 0; else if (ni >= m.n_normals) { rts_set_error("rts_set_scene: target %u normal index %u >= %u", t, ni, m.n_normals); return RTS_ERR_INVALID; }
                nidx[3*((size_t)h.tri_base + i) + k] = h.normal_base + ni;
            }
            ptarg[(size_t)h.tri_base + i] = t;
        }
        if (m.n_triangles && m.n_normals == 0 && c->params.interpolate_smooth) { rts_set_error("rts_set_scene: target %u has no normals but interpolate_smooth is set", t); return RTS_ERR_INVALID; }
        for (uint32_t i = 0; i < m.n_vertices; i++) { vtarg[(size_t)h.vert_base + i] = t; for (int k = 0; k < 3; k++) verts[3*((size_t)h.vert_base + i) + k] = m.vertices[3*(size_t)i + k]; }
        for (uint32_t i = 0; i < m.n_normals; i++) { ntarg[(size_t)h.normal_base + i] = t; for (int k = 0; k < 3; k++) normals[3*((size_t)h.normal_base + i) + k] = m.normals[3*(size_t)i + k]; }
    }
    // ---- the new scene object (swapped in at the end; every failure path below leaves the handle's old scene alone)
    struct SceneGuard { RtsScene* s; ~SceneGuard() { if (s) { s->release(); delete s; } } } guard{new RtsScene()};
    RtsScene* ns = guard.s; ns->device = c->device;
    RTS_HIP(hipStreamSynchronize(c->stream));
    RTS_HIP(ns->d_tri_vidx.reserve(3*nt + 1)); RTS_HIP(ns->d_tri_nidx.reserve(3*nt + 1)); RTS_HIP(ns->d_vert_targ.reserve(nv + 1)); RTS_HIP(ns->d_norm_targ.reserve(nn + 1));
    RTS_HIP(ns->d_prim_targ.reserve(nt + 1)); RTS_HIP(ns->d_verts_local.reserve(3*nv + 1)); RTS_HIP(ns->d_normals_local.reserve(3*nn + 1));
    if (nt) { RTS_HIP(hipMemcpy(ns->d_tri_vidx.p, vidx.data(), sizeof(uint32_t)*3*nt, hipMemcpyHostToDevice)); RTS_HIP(hipMemcpy(ns->d_tri_nidx.p, nidx.data(), sizeof(uint32_t)*3*nt, hipMemcpyHostToDevice));
              RTS_HIP(hipMemcpy(ns->d_prim_targ.p, ptarg.data(), sizeof(uint32_t)*nt, hipMemcpyHostToDevice)); }
    if (nv) { RTS_HIP(hipMemcpy(ns->d_vert_targ.p, vtarg.data(), sizeof(uint32_t)*nv, hipMemcpyHostToDevice)); RTS_HIP(hipMemcpy(ns->d_verts_local.p, verts.data(), sizeof(double)*3*nv, hipMemcpyHostToDevice)); }
    if (nn) { RTS_HIP(hipMemcpy(ns->d_norm_targ.p, ntarg.data(), sizeof(uint32_t)*nn, hipMemcpyHostToDevice)); RTS_HIP(hipMemcpy(ns->d_normals_local.p, normals.data(), sizeof(double)*3*nn, hipMemcpyHostToDevice)); }
    ns->meshes = mh; ns->n_prims = (uint32_t)nt; ns->n_verts = (uint32_t)nv; ns->n_normals = (uint32_t)nn;

    // ---- static target-space hierarchy, one per mesh; leaf slots name GLOBAL primitive ids.  Two builders, same node format:
    //   device (default) top-down binned SAH over split references, built on the GPU level by level (rts_lbvh.hip): milliseconds;
    //          RTS_DEVICE_TREE=lbvh: the Morton / Karras tree of round 2 instead
    //   host   RTS_BUILDER=host / RtsParams.flags & RTS_FLAG_HOST_BUILD: the same algorithm with a greedier reference splitter on
    //          one host thread per mesh (rts_sah.cpp): ~1 s per 100 k triangles; traces a lone C3 pulse 7 % faster (its slowest
    //          tile runs through the pole fans of the synthetic airframe), the pipelined benchmark, the dense control, spheres
    //          and C4 within 1-4 %
    const auto t_build0 = std::chrono::steady_clock::now();
    bool device_build = (c->params.flags & RTS_FLAG_HOST_BUILD) == 0;
    { const char* e = getenv("RTS_BUILDER"); if (e) device_build = (strcmp(e, "host") != 0); }
    // extra references for triangles whose boxes are mostly empty (rts_sah.cpp, rts_lbvh.hip): an average triangle gets about
    // 1 + budget references where cutting pays (2 measured best for both builders)
    double split_budget = 2.0;
    { const char* e = getenv("RTS_SPLIT_BUDGET"); if (e) { const double v = atof(e); if (v >= 0 && v <= 8) split_budget = v; } }
    if (device_build) {
        // The device builder sizes its bins by the references of a level and closes within 256 levels or gives up: out of memory, or
        // a mesh it does not close on, must not fail the scene -- the host builder takes over (RTS_DEVICE_BUILD_FALLBACK=0:
        // the error is returned, for tests of the builder itself).
        int rc = rts_lbvh_build_device(c, ns, vidx, mh, split_budget);
        if (rc == RTS_OK && getenv("RTS_DEBUG_FAIL_DEVICE_BUILD")) { rts_set_error("rts_set_scene: device builder failure injected (RTS_DEBUG_FAIL_DEVICE_BUILD)"); rc = RTS_ERR_HIP; }      // tests: the fall-back below, with the builder's state to clean up
        if (rc != RTS_OK) {
            const char* e = getenv("RTS_DEVICE_BUILD_FALLBACK");
            if (rc == RTS_ERR_INVALID || (e && e[0] == '0')) return rc;
            (void)hipGetLastError();                                     // (a failed allocation leaves a sticky error code behind)
            fprintf(stderr, "[rts] device hierarchy build failed (%s): falling back to the host builder\n", rts_last_error());
            ns->d_nodes4.release(); ns->d_leaf_prim.release(); ns->blas.clear(); ns->n_nodes = 0; ns->n_leaves = 0;
            device_build = false;
        } else ns->builder = 1;
    }
    if (!device_build) {
        std::vector<RtsNode4> nodes4; std::vector<uint32_t> leaf_prim; std::vector<RtsBlasInfo> blas(n_targets);
        {   // the meshes are independent: one host thread each (at most 16 at a time), results concatenated in target order
            std::vector<std::vector<RtsNode4>> pn(n_targets); std::vector<std::vector<uint32_t>> pl(n_targets); std::vector<int> prc(n_targets, RTS_OK);
            for (uint32_t t0 = 0; t0 < n_targets; t0 += 16) {
                std::vector<std::thread> pool;
                for (uint32_t t = t0; t < std::min(n_targets, t0 + 16); t++)
                    pool.emplace_back([&, t]() { prc[t] = rts_sah_build(meshes[t].vertices, meshes[t].triangles, meshes[t].n_triangles, split_budget, pn[t], pl[t], blas[t]); });
                for (auto& th : pool) th.join();
            }
            for (uint32_t t = 0; t < n_targets; t++) {
                if (prc[t] != RTS_OK) return prc[t];
                const int32_t node_base = (int32_t)nodes4.size(), leaf_base = (int32_t)leaf_prim.size();
                for (RtsNode4 nd : pn[t]) {
                    for (int k = 0; k < 4; k++) { int32_t& ch = nd.child[k]; if (ch == 0x7fffffff) continue; if (ch >= 0) ch += node_base; else ch = ~(~ch + leaf_base); }
                    nodes4.push_back(nd);
                }
                for (uint32_t lp : pl[t]) leaf_prim.push_back(lp + mh[t].tri_base);
                if (blas[t].root >= 0) blas[t].root += node_base;
            }
        }
        RTS_HIP(ns->d_nodes4.reserve(nodes4.size() + 1)); RTS_HIP(ns->d_leaf_prim.reserve(leaf_prim.size() + 1));
        if (!nodes4.empty()) RTS_HIP(hipMemcpy(ns->d_nodes4.p, nodes4.data(), sizeof(RtsNode4)*nodes4.size(), hipMemcpyHostToDevice));
        if (!leaf_prim.empty()) RTS_HIP(hipMemcpy(ns->d_leaf_prim.p, leaf_prim.data(), sizeof(uint32_t)*leaf_prim.size(), hipMemcpyHostToDevice));
        ns->blas = blas; ns->n_nodes = (uint32_t)nodes4.size(); ns->n_leaves = (uint32_t)leaf_prim.size();
        ns->builder = 0;
    }
    // Leaf records are kept per PRIMITIVE, not per leaf slot (a triangle cut into several references has several slots): the
    // nodes' leaf children are rewritten from ~slot to ~primitive here, once, whichever builder ran; the slot form is parked in
    // the record's unused words for rts_get_bvh.  A third of the per-pulse leaf refresh and of the leaf bytes on C3 (300 k slots
    // for 100 k triangles), and of the 240 MB per handle on BASELINE configs[3].
    if (ns->n_nodes) { k_children_to_prims<<<(ns->n_nodes + 255) / 256, 256, 0, c->stream>>>(ns->d_nodes4.p, ns->n_nodes, ns->d_leaf_prim.p); RTS_HIP(hipGetLastError()); RTS_HIP(hipStreamSynchronize(c->stream)); }
    {   // the octant versions of the node records (k_node_versions): RTS_NODE_VERSIONS=0: none; they need 1 KB per node -- a scene whose
        // versions would not fit a quarter of the device's free memory goes without (the kernel then walks d_nodes4 as before)
        const char* e = getenv("RTS_NODE_VERSIONS"); const bool want = !(e && e[0] == '0');
        const char* km = getenv("RTS_VERSION_KEY"); const int key_mode = (km && strcmp(km, "corner") == 0) ? 0 : 1;      // (measured, profiles/r05a_versions_ab.log: by centre the dense control visits 5.9 % more nodes than the sorted walk, by entry corner 9.6 %)
        size_t free_b = 0, total_b = 0; (void)hipMemGetInfo(&free_b, &total_b);
        const size_t need = (size_t)ns->n_nodes * 8 * sizeof(RtsNode4);
        if (want && ns->n_nodes && ns->n_nodes < (1u << 28) && need < free_b / 4) {
            if (ns->d_nodes4v.reserve((size_t)ns->n_nodes * 8) == hipSuccess) {
                k_node_versions<<<(ns->n_nodes * 8u + 255u) / 256u, 256, 0, c->stream>>>(ns->d_nodes4.p, ns->n_nodes, ns->d_nodes4v.p, key_mode);
                RTS_HIP(hipGetLastError()); RTS_HIP(hipStreamSynchronize(c->stream));
            } else (void)hipGetLastError();
        }
    }
    ns->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();

    // ---- swap it in
    if (--c->scene->refs == 0) { c->scene->release(); delete c->scene; }
    c->scene = ns; guard.s = nullptr;
    return rts_attach_scene(c);
}

// Per-handle buffers sized by the scene the handle now points at; placement and tile history start afresh.
static int rts_attach_scene(RtsContext* c)
{
    const RtsScene* sc = c->scene; const uint32_t n_targets = (uint32_t)sc->meshes.size();
    RTS_HIP(c->d_leaves.reserve((size_t)sc->n_prims + 1));      // one record per primitive (k_children_to_prims)
    RTS_HIP(c->d_verts_world.reserve(3*(size_t)sc->n_verts + 1)); RTS_HIP(c->d_normals_world.reserve(3*(size_t)sc->n_normals + 1));
    {   // device image of the pinned block's [lc | motion | td]
        const size_t span = offsetof(RtsPinned, cnt);
        RTS_HIP(c->d_params.reserve(span));
        c->p_lc = reinterpret_cast<RtsLaunchConsts*>(c->d_params.p + offsetof(RtsPinned, lc));
        c->p_motion = reinterpret_cast<RtsTargetMotion*>(c->d_params.p + offsetof(RtsPinned, motion));
        c->p_targets = reinterpret_cast<RtsTargetDev*>(c->d_params.p + offsetof(RtsPinned, td));
        c->rcs_uploaded = false;
    }
    c->verts_world_valid = false; c->order_sum_valid = false;
    if (c->hist->refs.load() > 1) { c->hist->refs--; c->hist = new RtsTileHist(); }      // (a handle that leaves a shared scene leaves the shared history)
    c->motion.assign(n_targets, RtsTargetMotion{}); c->motion_valid = false; c->bvh_valid = false; c->hist->n = 0; c->hist->any = false; c->hist->head_hint_valid = false; c->tile_cost_pending = false; c->tile_last_valid = false;
    return RTS_OK;
}

// rts_share_scene: dst drops its own scene and uses src's (same device): the meshes, the hierarchy and its leaf order then
// exist once however many handles keep pulses in flight, and the hierarchy is built once.
extern "C" int rts_share_scene(RtsHandle dst, RtsHandle src)
{
    if (!dst || !src || dst == src) { rts_set_error("rts_share_scene: needs two distinct handles"); return RTS_ERR_INVALID; }
    if (dst->device != src->device) { rts_set_error("rts_share_scene: handles live on different devices (%d, %d)", dst->device, src->device); return RTS_ERR_INVALID; }
    if (dst->params.interpolate_smooth && !src->params.interpolate_smooth) {
        for (const RtsMeshHost& m : src->scene->meshes) if (m.n_tris && m.n_normals == 0) { rts_set_error("rts_share_scene: the scene has a mesh without normals but the receiving handle interpolates normals"); return RTS_ERR_INVALID; }
    }
    RtsContext* c = dst;
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    RTS_HIP(hipStreamSynchronize(c->stream));
    if (dst->scene == src->scene) return RTS_OK;
    if (--dst->scene->refs == 0) { dst->scene->release(); delete dst->scene; }
    dst->scene = src->scene; dst->scene->refs++;
    { int rc = rts_attach_scene(dst); if (rc != RTS_OK) return rc; }
    if (dst->share_history && src->share_history && dst->params.width == src->params.width) {      // ... and the tile-cost history of the scene's handles (RtsTileHist)
        if (--dst->hist->refs == 0) { dst->hist->d.release(); delete dst->hist; }
        dst->hist = src->hist; dst->hist->refs++;
    }
    return RTS_OK;
}

extern "C" int rts_scene_info(RtsHandle c, RtsSceneInfo* out)
{
    if (!c || !out) { rts_set_error("rts_scene_info: null argument"); return RTS_ERR_INVALID; }
    const RtsScene* sc = c->scene;
    memset(out, 0, sizeof(*out));
    out->n_targets = (uint32_t)sc->meshes.size(); out->n_prims = sc->n_prims; out->n_nodes = sc->n_nodes; out->n_leaves = sc->n_leaves;
    out->handles_sharing = (uint32_t)sc->refs.load(); out->builder = sc->builder; out->build_ms = sc->build_ms;
    out->shared_device_bytes = sc->device_bytes(); out->version_bytes = sc->d_nodes4v.cap * sizeof(RtsNode4);
    out->handle_device_bytes = c->d_leaves.cap * sizeof(RtsLeafTri) + c->d_verts_world.cap * 8 + c->d_normals_world.cap * 8 + c->d_params.cap;
    return RTS_OK;
}

extern "C" int rts_set_receivers(RtsHandle c, const RtsReceiverSphere* rx, uint32_t n_rx)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (n_rx && !rx) { rts_set_error("rts_set_receivers: null array"); return RTS_ERR_INVALID; }
    if (n_rx > 65535) { rts_set_error("rts_set_receivers: more than 65535 receivers"); return RTS_ERR_UNSUPPORTED; }
    { int rc = check_key_width(c->depth, (uint32_t)c->scene->meshes.size(), n_rx, "rts_set_receivers"); if (rc != RTS_OK) return rc; }
    std::vector<RtsRxDev> h(n_rx);
    for (uint32_t i = 0; i < n_rx; i++) {
        const RtsReceiverSphere& r = rx[i];
        const double vals[8] = { r.centre[0], r.centre[1], r.centre[2], r.radius, r.min_theta, r.max_theta, r.min_phi, r.max_phi };
        for (int k = 0; k < 8; k++) if (!std::isfinite(vals[k])) { rts_set_error("rts_set_receivers: receiver %u has a non-finite field", i); return RTS_ERR_INVALID; }
        // normalise_angle (ray_tracer.cu:53-57) walks in steps of 2 pi: keep the walk short
        if (std::fabs(r.min_theta) > 1e4 || std::fabs(r.max_theta) > 1e4 || std::fabs(r.min_phi) > 1e4 || std::fabs(r.max_phi) > 1e4) { rts_set_error("rts_set_receivers: receiver %u window angle magnitude > 1e4 rad", i); return RTS_ERR_INVALID; }
        h[i] = RtsRxDev{ r.centre[0], r.centre[1], r.centre[2], r.radius, r.min_theta, r.max_theta, r.min_phi, r.max_phi };
    }
    // A caller in a pulse loop sets the receivers before every pulse (the reference recomputes the capture spheres per pulse,
    // ray_tracer.cpp:894-918): an unchanged set costs nothing, a changed one is uploaded from pinned staging ON THE HANDLE'S
    // STREAM, ahead of the next trace -- no stream drain, no blocking copy (the handle has no pulse in flight here, so the
    // previous upload from the staging has long completed: its pulse was waited for).
    if (c->n_rx == n_rx && c->rx_host.size() == n_rx && (n_rx == 0 || memcmp(c->rx_host.data(), h.data(), sizeof(RtsRxDev) * n_rx) == 0)) return RTS_OK;
    if (n_rx <= 1024) {
        if (c->pin_rx_cap < n_rx) {
            if (c->pin_rx) { RTS_HIP(hipStreamSynchronize(c->stream)); (void)hipHostFree(c->pin_rx); c->pin_rx = nullptr; c->pin_rx_cap = 0; }
            const uint32_t cap = std::max<uint32_t>(64u, n_rx);
            RTS_HIP(hipHostMalloc((void**)&c->pin_rx, sizeof(RtsRxDev) * cap, hipHostMallocDefault)); c->pin_rx_cap = cap;
        }
        if (c->d_rx.cap < (size_t)n_rx + 1) { RTS_HIP(hipStreamSynchronize(c->stream)); RTS_HIP(c->d_rx.reserve(n_rx + 1)); }
        if (n_rx) { memcpy(c->pin_rx, h.data(), sizeof(RtsRxDev) * n_rx); RTS_HIP(hipMemcpyAsync(c->d_rx.p, c->pin_rx, sizeof(RtsRxDev) * n_rx, hipMemcpyHostToDevice, c->stream)); }
    } else {
        RTS_HIP(hipStreamSynchronize(c->stream));
        RTS_HIP(c->d_rx.reserve(n_rx + 1));
        RTS_HIP(hipMemcpy(c->d_rx.p, h.data(), sizeof(RtsRxDev)*n_rx, hipMemcpyHostToDevice));
    }
    c->rx_host = h; c->n_rx = n_rx;
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- launch constants
// sph_to_cart (ray_tracer.cu:132-139) and the launch-uniform part of ray_generation
// (ray_tracer.cu:155-156, 167-169 step factors, 173-175 Rot, 186-196 orth_vec and Rot1),
// evaluated once on the host in the reference's expression order.
static inline dvec3 host_sph_to_cart(double azi, double ele) { dvec3 c; c.x = std::cos(azi)*std::cos(ele); c.y = std::sin(azi)*std::cos(ele); c.z = std::sin(ele); return c; }

static void fill_launch_constants(RtsLaunchConsts& a, const RtsPulse& p, uint32_t W)
{
    const double spx = p.tx_span[0], spy = p.tx_span[1], spz = p.tx_span[2];
    const double dx = p.tx_dir[0], dy = p.tx_dir[1];
    a.ox = p.ray_origin[0]; a.oy = p.ray_origin[1]; a.oz = p.ray_origin[2];
    const dvec3 beamStart = host_sph_to_cart(-spx/2, -spy/2);
    const dvec3 beamEnd = host_sph_to_cart(spx/2, spy/2);
    a.bsx = beamStart.x; a.bsy = beamStart.y; a.bsz = beamStart.z;
    const dvec3 w1 = host_sph_to_cart(dx, dy);
    a.w1x = w1.x; a.w1y = w1.y; a.w1z = w1.z;
    a.stx = a.sty = a.stz = 0;
    if (W > 1) {
        a.stx = (((beamEnd.x*(1 + spz)) - beamStart.x)/(W - 1));
        a.sty = ((beamEnd.y - beamStart.y)/(W - 1));
        a.stz = ((beamEnd.z - beamStart.z)/(W - 1));
    }
    const double Rot[3][3] = {{std::cos(dx), -std::sin(dx), 0}, {std::sin(dx), std::cos(dx), 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.rot[3*i + j] = Rot[i][j];
    dvec3 rotated; rotated.x = 0; rotated.y = 0; rotated.z = 0;
    rotated.x += Rot[0][1]; rotated.y += Rot[1][1]; rotated.z += Rot[2][1];
    const double on = std::sqrt(rotated.x*rotated.x + rotated.y*rotated.y + rotated.z*rotated.z);
    const dvec3 orth_vec = mk3(rotated.x/on, rotated.y/on, rotated.z/on);
    const double Rot1[3][3] = {
        {std::cos(dy) + orth_vec.x*orth_vec.x*(1 - std::cos(dy)), orth_vec.x*orth_vec.y*(1 - std::cos(dy)) + orth_vec.z*std::sin(dy), orth_vec.x*orth_vec.z*(1 - std::cos(dy)) - orth_vec.y*std::sin(dy)},
        {orth_vec.y*orth_vec.x*(1 - std::cos(dy)) - orth_vec.z*std::sin(dy), std::cos(dy) + orth_vec.y*orth_vec.y*(1 - std::cos(dy)), orth_vec.y*orth_vec.z*(1 - std::cos(dy)) + orth_vec.x*std::sin(dy)},
        {orth_vec.z*orth_vec.x*(1 - std::cos(dy)) + orth_vec.y*std::sin(dy), orth_vec.z*orth_vec.y*(1 - std::cos(dy)) - orth_vec.x*std::sin(dy), std::cos(dy) + orth_vec.z*orth_vec.z*(1 - std::cos(dy))}};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a.rot1[3*i + j] = Rot1[i][j];
}

// Frame and extent of the primary-ray mask (RtsMaskFrame): b = direction of the beam's middle, (u, v) an orthonormal complement;
// the directions of a launch are the central projection of the (convex) lattice box, so the perspective coordinates of its
// eight corners bound those of every ray.
static void fill_mask_frame(RtsLaunchConsts& a, bool enable)
{
    RtsMaskFrame& f = a.mask; memset(&f, 0, sizeof(f));
    if (!enable || a.W < 2) return;
    dvec3 c[8]; dvec3 sum = mk3(0, 0, 0);
    for (int k = 0; k < 8; k++) {
        c[k] = rts_lattice_dir(a, (k & 1) ? a.W - 1 : 0, (k & 2) ? a.W - 1 : 0, (k & 4) ? a.W - 1 : 0);
        const double n = len3(c[k]); if (!(n > 0) || !std::isfinite(n)) return;
        c[k] = mk3(c[k].x / n, c[k].y / n, c[k].z / n); sum = add3(sum, c[k]);
    }
    const double sn = len3(sum); if (!(sn > 1e-6)) return;
    const dvec3 b = mk3(sum.x / sn, sum.y / sn, sum.z / sn);
    dvec3 ref = std::fabs(b.z) < 0.9 ? mk3(0, 0, 1) : mk3(1, 0, 0);
    dvec3 u = cross3(ref, b); const double un = len3(u); u = mk3(u.x / un, u.y / un, u.z / un);
    const dvec3 v = cross3(b, u);
    double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300;
    for (int k = 0; k < 8; k++) {
        const double w = dot3(c[k], b); if (!(w > 0.5)) return;                  // beam wider than ~120 degrees: no mask
        const double uu = dot3(c[k], u) / w, vv = dot3(c[k], v) / w;
        u0 = std::min(u0, uu); u1 = std::max(u1, uu); v0 = std::min(v0, vv); v1 = std::max(v1, vv);
    }
    const double du = std::max(u1 - u0, 1e-9), dv = std::max(v1 - v0, 1e-9);
    u0 -= 0.002 * du; u1 += 0.002 * du; v0 -= 0.002 * dv; v1 += 0.002 * dv;   // rays exactly on the rim stay inside the bitmap
    f.bx = (float)b.x; f.by = (float)b.y; f.bz = (float)b.z; f.ux = (float)u.x; f.uy = (float)u.y; f.uz = (float)u.z; f.vx = (float)v.x; f.vy = (float)v.y; f.vz = (float)v.z;
    const double n_fit = std::floor(std::min(u1 - u0, v1 - v0) / RTS_MASK_MIN_CELL);
    if (!(n_fit >= 32.0)) return;                                                // beam narrower than 32 cells of the minimum size: no mask
    const uint32_t n = (uint32_t)std::min<double>(RTS_MASK_N, n_fit) & ~31u;     // whole 32-bit words per row
    f.u0 = (float)u0; f.v0 = (float)v0; f.inv_du = (float)(n / (u1 - u0)); f.inv_dv = (float)(n / (v1 - v0));
    f.n = n;
}

// Per-target placement constants of a pulse: inverse rotation, world bounds of the placed hierarchy, error slack.
static int fill_target_placement(const RtsContext* c, uint32_t t, RtsTargetDev& td)
{
    const RtsTargetMotion& m = c->motion[t]; const RtsBlasInfo& b = c->scene->blas[t];
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (m.has_rotation) for (int k = 0; k < 9; k++) R[k] = m.rotation[k];
    for (int k = 0; k < 9; k++) if (!std::isfinite(R[k])) { rts_set_error("rts_trace_pulse: target %u has a non-finite rotation", t); return RTS_ERR_INVALID; }
    // the hierarchy is rigid: R must be a rotation up to rounding (the reference's are products of float-trig
    // axis rotations, ray_tracer.cpp:95-118, orthonormal to ~1e-7)
    double dev = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double g = 0; for (int k = 0; k < 3; k++) g += R[3*k + i] * R[3*k + j];
        dev = std::max(dev, std::fabs(g - (i == j ? 1.0 : 0.0)));
    }
    if (!(dev < 1e-3)) { rts_set_error("rts_trace_pulse: target %u: rotation is not orthonormal (|R^T R - I| = %.3g); only rigid placements are supported", t, dev); return RTS_ERR_UNSUPPORTED; }
    const double det = R[0]*(R[4]*R[8] - R[5]*R[7]) - R[1]*(R[3]*R[8] - R[5]*R[6]) + R[2]*(R[3]*R[7] - R[4]*R[6]);
    const double id = 1.0 / det;
    td.rinv[0] = (R[4]*R[8] - R[5]*R[7]) * id; td.rinv[1] = (R[2]*R[7] - R[1]*R[8]) * id; td.rinv[2] = (R[1]*R[5] - R[2]*R[4]) * id;
    td.rinv[3] = (R[5]*R[6] - R[3]*R[8]) * id; td.rinv[4] = (R[0]*R[8] - R[2]*R[6]) * id; td.rinv[5] = (R[2]*R[3] - R[0]*R[5]) * id;
    td.rinv[6] = (R[3]*R[7] - R[4]*R[6]) * id; td.rinv[7] = (R[1]*R[6] - R[0]*R[7]) * id; td.rinv[8] = (R[0]*R[4] - R[1]*R[3]) * id;
    td.px = m.position[0]; td.py = m.position[1]; td.pz = m.position[2];
    td.root = b.root;
    // bounding sphere of the placed hierarchy: image of the local box centre, half diagonal stretched by the
    // rotation's deviation from orthonormality, padded by 1e-6 of the world scale
    double cl[3], half2 = 0, wc[3], wmax = 0;
    for (int i = 0; i < 3; i++) { cl[i] = 0.5 * b.lo[i] + 0.5 * b.hi[i]; const double h = 0.5 * (b.hi[i] - b.lo[i]); half2 += h * h; }
    for (int i = 0; i < 3; i++) { wc[i] = R[3*i]*cl[0] + R[3*i + 1]*cl[1] + R[3*i + 2]*cl[2] + m.position[i]; wmax = std::max(wmax, std::fabs(wc[i])); }
    double rad = std::sqrt(half2) * (1.0 + 2.0 * dev); wmax += rad;
    rad = rad * (1.0 + 1e-6) + wmax * 1e-6 + 1e-30;
    td.cx = wc[0]; td.cy = wc[1]; td.cz = wc[2]; td.r2 = rad * rad;
    // Extra origin slack of the target-space slab test.  What it has to cover is the one error that scales with the WORLD
    // coordinates: the exact test runs on vw = fl(fl(R v) + p), which is off the ideal R v + p by <= 2^-53 |vw| per
    // coordinate (one rounding of the final sum; the partial sums live at target scale), i.e. <= sqrt(3) 2^-53 wmax along a
    // target axis = 1.2e-9 m at Earth-centred coordinates (6.4e6 m).  Everything else -- the mapping R^-1 (o - p), o - p
    // itself (Sterbenz-exact or correctly rounded), the f32 narrowing of the mapped origin -- is relative to |o - p| and
    // is inside the 3e-7 |o'| of rts_slab_setup.  2^-49 wmax is that bound times 9.  (Round 1 used 4e-9 wmax: harmless
    // near the origin, but 2.6 cm on every slab plane of a mesh of 10-30 cm triangles at 6.4e6 m.)
    td.ew = (float)(wmax * c->ew_rel) + 1.0e-30f;
    if (b.root < 0 || !std::isfinite(td.cx + td.cy + td.cz + td.r2)) td.root = -1;
    return RTS_OK;
}

// Device buffers of a launch of up to n launch indices (grown, never shrunk).  Also the first touch of the big slabs:
// doing it before the first pulse keeps multi-GB hipMalloc calls out of a caller's timed or latency-critical region.
extern "C" int rts_reserve(RtsHandle c, uint64_t n_rays)
{
    CHECK_HANDLE(c);
    const uint64_t W3 = (uint64_t)c->params.width * c->params.width * c->params.width;
    const uint64_t n = std::min<uint64_t>(n_rays ? n_rays : W3, W3);
    const uint32_t chains = c->params.max_refr ? 3u : 1u;
    if (n * chains > 0xfffffff0ULL) { rts_set_error("rts_reserve: rays x chains exceeds 2^32"); return RTS_ERR_UNSUPPORTED; }
    const size_t threads = (size_t)c->n_cu * 64 * RTS_BLOCK;             // upper bound of any launch's grid
    const uint32_t H = c->params.max_refl + 1;
    RTS_HIP(c->d_recv.reserve((size_t)n * chains + 1));
    RTS_HIP(c->d_tile_ctr.reserve(RTS_ZERO_WORDS + RTS_MASK_WORDS + 64)); c->p_counters = reinterpret_cast<unsigned long long*>(c->d_tile_ctr.p + RTS_OFF_COUNTERS);
    RTS_HIP(c->d_dir_hist.reserve((size_t)(c->params.max_refr ? 3 * H : std::max<uint32_t>(c->params.max_refl, 1)) * 3 * n + 4));
    const size_t coop_threads = c->coop_frac > 0.0 ? (size_t)c->coop_grid_max * RTS_BLOCK : 0;
    if (c->params.max_refr) RTS_HIP(c->d_child.reserve(2 * (threads + coop_threads)));
    RTS_HIP(c->d_stack_ovf.reserve((size_t)RTS_STACK_OVF * ((size_t)c->n_cu * 1024 + coop_threads)));
    RTS_HIP(c->d_block_counters.reserve(((size_t)c->n_cu * 64 + c->coop_grid_max) * 8));
    const size_t n_tiles = (size_t)((n + RTS_WTILE - 1) / RTS_WTILE), n_hist = (size_t)((W3 + RTS_WTILE - 1) / RTS_WTILE);
    RTS_HIP(c->d_tile_ctr.reserve(RTS_ZERO_WORDS + RTS_MASK_WORDS + 64)); c->p_counters = reinterpret_cast<unsigned long long*>(c->d_tile_ctr.p + RTS_OFF_COUNTERS); RTS_HIP(c->d_tile_cost.reserve(n_tiles)); RTS_HIP(c->d_tile_key.reserve(n_tiles)); RTS_HIP(c->d_tile_key_sorted.reserve(n_tiles));
    RTS_HIP(c->d_tile_id.reserve(n_tiles)); RTS_HIP(c->d_tile_order.reserve(n_tiles));
    if (c->hist->n != (uint32_t)n_hist) {
        std::lock_guard<std::mutex> lk(g_hist_mu);      // (handles that share the history may be driven from different threads)
        RTS_HIP(hipDeviceSynchronize()); RTS_HIP(c->hist->d.reserve(n_hist)); RTS_HIP(hipMemset(c->hist->d.p, 0, sizeof(uint32_t) * n_hist));      // (blocking: the table may be shared with handles on other streams)
        c->hist->n = (uint32_t)n_hist; c->hist->any = false; c->tile_cost_pending = false;
    }
    if (c->params.flags & RTS_FLAG_KEEP_ALL_RAYS) { RTS_HIP(c->d_all.reserve((size_t)n * chains + 1)); RTS_HIP(c->d_hit_prim.reserve((size_t)n * H + 1)); RTS_HIP(c->d_hit_t.reserve((size_t)n * H + 1)); }
    // touch the two slabs the trace kernel writes sparsely, so that their pages exist before the first launch
    RTS_HIP(hipMemsetAsync(c->d_recv.p, 0, sizeof(RtsEndRecord) * ((size_t)n * chains + 1), c->stream));
    RTS_HIP(hipMemsetAsync(c->d_dir_hist.p, 0, sizeof(float) * c->d_dir_hist.cap, c->stream));
    RTS_HIP(hipStreamSynchronize(c->stream));
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- launch
extern "C" int rts_trace_pulse(RtsHandle c, const RtsPulse* p)
{
    int rc = rts_trace_pulse_begin(c, p);
    return rc != RTS_OK ? rc : rts_trace_pulse_end(c);
}

// Everything of a pulse up to and including the trace kernel, left in flight on the handle's stream.
extern "C" int rts_trace_pulse_begin(RtsHandle c, const RtsPulse* p)
{
    if (!c) { rts_set_error("null handle"); return RTS_ERR_INVALID; }
    RtsLapTimer lt(c);
    CHECK_HANDLE(c);
    lt.lap(5);
    if (!p) { rts_set_error("rts_trace_pulse: null pulse"); return RTS_ERR_INVALID; }
    if (c->pulse_open) { rts_set_error("rts_trace_pulse_begin: the previous pulse of this handle was begun but not ended"); return RTS_ERR_INVALID; }
    if (c->spec_pending) { int rc_ = rts_spec_resolve(c); if (rc_ != RTS_OK) return rc_; }
    lt.lap(6);
    const uint32_t W = c->params.width;
    const uint64_t total = (uint64_t)W * W * W;
    uint64_t first = p->ray_first, count = p->ray_count ? p->ray_count : (total > first ? total - first : 0);
    uint32_t il_tile = 0, il_parts = 0, il_part = 0;
    const bool il_list = p->interleave_parts == RTS_INTERLEAVE_LIST;      // the tiles dealt to this handle (rts_set_tile_list)
    if (il_list) {
        if (c->il_list_tile == 0 || p->interleave_tile != c->il_list_tile) { rts_set_error("rts_trace_pulse: RTS_INTERLEAVE_LIST with tile %u, but the handle's tile list has %u tiles of %u launch indices (rts_set_tile_list)", p->interleave_tile, c->il_list_n, c->il_list_tile); return RTS_ERR_INVALID; }
        il_tile = c->il_list_tile; il_parts = RTS_INTERLEAVE_LIST; il_part = c->il_list_gen;
    } else if (p->interleave_parts > 1) {
        il_tile = p->interleave_tile; il_parts = p->interleave_parts; il_part = p->interleave_part;
        if (il_tile == 0 || il_part >= il_parts) { rts_set_error("rts_trace_pulse: bad interleave (tile %u, part %u of %u)", il_tile, il_part, il_parts); return RTS_ERR_INVALID; }
    }
    if (first > total || count > total - first) { rts_set_error("rts_trace_pulse: ray range [%llu, +%llu) outside W^3 = %llu", (unsigned long long)first, (unsigned long long)count, (unsigned long long)total); return RTS_ERR_INVALID; }
    const uint32_t n_targets = (uint32_t)c->scene->meshes.size();
    hipStream_t st = c->stream;
    c->agg_valid = false; c->agg_pending.valid = false; c->n_recv = 0;
    c->mirror.want = false; c->mirror.recv_valid = false; c->mirror.agg_valid = false; c->v_recv_have = 0;

    // ---- scene placement: only when a target actually moved
    RTS_HIP(hipEventRecord(c->ev[0], st));
    bool moved = !c->bvh_valid;
    if (p->motion) {
        if (!c->motion_valid || memcmp(c->motion.data(), p->motion, sizeof(RtsTargetMotion)*n_targets) != 0) moved = true;
        if (n_targets) memcpy(c->motion.data(), p->motion, sizeof(RtsTargetMotion)*n_targets);
        c->motion_valid = true;
    } else if (!c->motion_valid) {
        for (auto& m : c->motion) memset(&m, 0, sizeof(m));
        c->motion_valid = true; moved = true;
    }
    c->stats.bvh_rebuilt = 0;
    if (moved) {
        for (uint32_t t = 0; t < n_targets; t++) for (int k = 0; k < 3; k++)
            if (!std::isfinite(c->motion[t].position[k]) || !std::isfinite(c->motion[t].velocity[k])) { rts_set_error("rts_trace_pulse: target %u has a non-finite position/velocity", t); return RTS_ERR_INVALID; }
        if (n_targets > 256) { rts_set_error("rts_trace_pulse: more than 256 targets"); return RTS_ERR_UNSUPPORTED; }
        // pinned staging is safe to rewrite: every earlier upload precedes the previous launch's trace kernel, whose
        // completion the host already waited for (received-count readback)
        RtsTargetDev* td = c->pin->td;
        for (uint32_t t = 0; t < n_targets; t++) {
            td[t].reflCoeff = c->scene->meshes[t].refl_coeff; td[t].vx = c->motion[t].velocity[0]; td[t].vy = c->motion[t].velocity[1]; td[t].vz = c->motion[t].velocity[2];
            td[t].tri_base = c->scene->meshes[t].tri_base; td[t].perface_normals = c->scene->meshes[t].perface ? 1u : 0u; td[t].refrIndex = c->scene->meshes[t].refr_index;
            int rc = fill_target_placement(c, t, td[t]); if (rc != RTS_OK) { c->bvh_valid = false; c->motion_valid = false; return rc; }
            c->pin->motion[t] = c->motion[t];
        }
        c->bvh_valid = false;                                     // (until the upload and the placement kernels below have been enqueued: an early return in between must not leave the device with the old placement)
        // (uploaded below, together with the launch constants: one copy; the placement kernels follow it)
    }

    // ---- per-pulse buffers
    if (il_list) {           // every listed tile is whole, except the last tile of the range when it is listed (the list is ascending: it is the last entry)
        const uint64_t range_tiles = (count + il_tile - 1) / il_tile;
        if (c->il_list_n == 0) count = 0;
        else if (c->il_list_last >= range_tiles) { rts_set_error("rts_trace_pulse: the handle's tile list names tile %u, the range has %llu tiles of %u launch indices", c->il_list_last, (unsigned long long)range_tiles, il_tile); return RTS_ERR_INVALID; }
        else {
            uint64_t cnt = (uint64_t)c->il_list_n * il_tile;
            if (c->il_list_last == range_tiles - 1) cnt -= range_tiles * il_tile - count;
            count = cnt;
        }
    } else if (il_parts > 1) {      // number of launch indices of the range that fall into this part's tiles
        const uint64_t stride = (uint64_t)il_tile * il_parts, full = count / stride, rem = count % stride;
        const uint64_t lo = (uint64_t)il_part * il_tile;
        count = full * il_tile + (rem > lo ? std::min<uint64_t>(rem - lo, il_tile) : 0);
    }
    const uint32_t n = (uint32_t)count;
    c->ray_first = first; c->n_rays = n;
    const bool keep_all = (c->params.flags & RTS_FLAG_KEEP_ALL_RAYS) != 0;
    const bool count_trav = (c->params.flags & RTS_FLAG_COUNT_TRAVERSAL) != 0;
    const bool shared_gpu = c->grid_spare_forced || g_open_pulses[c->device & 63].load() > 0;
    const int grid_mult = c->grid_mult, grid_spare = shared_gpu ? c->grid_spare : 0;   // blocks per CU: 4 = exactly the resident set (waves draw tiles from a queue); block slots left free
    // the trace kernel's blocks are persistent and four of them fill a CU's register file: leave a few block slots free so
    // that the short kernels of the neighbouring pulses (other streams) are not locked out for the whole launch
    uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)n + RTS_BLOCK - 1) / RTS_BLOCK, (uint64_t)std::max<int>((c->n_cu * grid_mult - grid_spare * c->n_cu / 256) * (256 / RTS_BLOCK), c->n_cu));   // (grid_spare: block slots per 256 CUs; 160 measured best with three pulses in flight: 0.709 vs 0.735 ms/pulse at 64)
    if (grid == 0) grid = 1;
    RtsTraceArgs a; memset(&a, 0, sizeof(a));
    RtsLaunchConsts& lc = c->last_lc; memset(&lc, 0, sizeof(lc));
    fill_launch_constants(lc, *p, W);
    lc.ray_first = first; lc.W = W; lc.il_tile = il_tile; lc.il_parts = il_parts; lc.il_part = il_part; lc.il_list = il_list ? c->d_il_list.p : nullptr;
    if (W >= 2) {                                                     // branch-free magic number of the division by W (libdivide's u32 scheme)
        const uint32_t fl = 31u - (uint32_t)__builtin_clz(W);
        if ((W & (W - 1)) == 0) { lc.w_magic = 0; lc.w_more = fl - 1; }
        else {
            const uint64_t k2 = 1ULL << (32 + fl); uint64_t pm = k2 / W; const uint64_t rem = k2 - pm * W;
            pm += pm; const uint64_t tr = rem + rem;
            if (tr >= W || tr < rem) pm += 1;
            lc.w_magic = (uint32_t)(1 + pm); lc.w_more = fl;
        }
    }
    const bool pre_filter = c->use_pmask && !c->pre_dense;           // (a launch where most rays hit pays for the filter and skips nothing)
    fill_mask_frame(lc, pre_filter && c->scene->n_prims > 0);
    for (int k = 0; k < 3; k++) { lc.f_bs[k] = (float)(&lc.bsx)[k]; lc.f_st[k] = (float)(&lc.stx)[k]; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) lc.f_m[3 * i + j] = (float)(lc.rot1[3 * i] * lc.rot[j] + lc.rot1[3 * i + 1] * lc.rot[3 + j] + lc.rot1[3 * i + 2] * lc.rot[6 + j]);
    lt.lap(0);
    // ---- the pulse's parameters in ONE upload: launch constants, and -- when a target moved -- the placements behind them
    c->pin->lc = lc;
    RTS_HIP(hipMemcpyAsync(c->d_params.p, &c->pin->lc, moved && n_targets ? offsetof(RtsPinned, td) + sizeof(RtsTargetDev) * n_targets : sizeof(RtsLaunchConsts), hipMemcpyHostToDevice, st));
    a.lc = c->p_lc;
    // ONE fill per pulse: the draw counters of both kernels, the order's head words and bins, the launch's 16 counters and --
    // behind them, when this pulse has one -- the primary-ray mask
    RTS_HIP(c->d_tile_ctr.reserve(RTS_ZERO_WORDS + RTS_MASK_WORDS + 64));
    RTS_HIP(hipMemsetAsync(c->d_tile_ctr.p, 0, sizeof(uint32_t) * (((RTS_ZERO_WORDS + (lc.mask.n ? (size_t)lc.mask.n * lc.mask.n / 32u + 1u : 0u)) + 63u) & ~(size_t)63u), st));      // (a whole number of 256-byte pieces: the runtime splits an odd-sized fill into two kernels)
    uint32_t* const pmask = c->d_tile_ctr.p + RTS_ZERO_WORDS;
    lt.lap(1);
    { int rc = rts_scene_place(c, lc, moved, pmask); if (rc != RTS_OK) return rc; }      // placement (a target moved) + the primary-ray mask: one pass over the leaves for both
    if (moved) { RTS_STAGE(c, "scene_place"); c->bvh_valid = true; c->stats.bvh_rebuilt = 1; }
    RTS_HIP(hipEventRecord(c->ev[1], st));
    lt.lap(2);
    a.ray_first = first; a.n_rays = n; a.W = W; a.max_refl = c->params.max_refl; a.smooth = c->params.interpolate_smooth ? 1u : 0u;
    a.n_prims = c->scene->n_prims; a.n_targets = n_targets; a.n_rx = c->n_rx; a.keep_all = keep_all ? 1u : 0u;
    a.max_refr = c->params.max_refr; a.rows = a.max_refr ? c->params.max_refl + 3 : 1;
    const uint32_t chains = a.max_refr ? 3u : 1u;
    if ((uint64_t)n * chains > 0xfffffff0ULL) { rts_set_error("rts_trace_pulse: rays x chains exceeds 2^32"); return RTS_ERR_UNSUPPORTED; }
    a.total_threads = grid * RTS_BLOCK;
    a.async_idle0 = c->async_idle0; a.async_idle1 = c->async_idle1; a.async_age = c->async_age;
    a.coop_spread = c->coop_spread;
    a.coop_walk_steps_lo = std::min(c->coop_walk_steps_lo, c->coop_walk_steps);
    a.coop_walk_steps = c->coop_walk_steps; a.coop_min_cost = c->coop_walk_steps ? std::min<uint32_t>(c->coop_floor, 1875u) : 0u;      // (nothing shorter than 50 us is looked at; RTS_COOP_STEPS=0: every tile is flagged)
    const uint32_t coop_threads = c->coop_frac > 0.0 ? c->coop_grid_max * RTS_BLOCK : 0u;      // the cooperative kernel's rows of the per-thread slabs
    a.slab_threads = a.total_threads + coop_threads;
    RTS_HIP(c->d_recv.reserve((size_t)n * chains + 1));
    RTS_HIP(c->d_dir_hist.reserve((size_t)(a.max_refr ? 3 * (c->params.max_refl + 1) : std::max<uint32_t>(c->params.max_refl, 1)) * 3 * n + 4));
    if (a.max_refr) RTS_HIP(c->d_child.reserve((size_t)2 * a.slab_threads));
    RTS_HIP(c->d_stack_ovf.reserve((size_t)RTS_STACK_OVF * a.slab_threads));
    RTS_HIP(c->d_block_counters.reserve(((size_t)grid + c->coop_grid_max) * 8));
    if (keep_all) {
        RTS_HIP(c->d_all.reserve((size_t)n * chains + 1)); RTS_HIP(c->d_hit_prim.reserve((size_t)n * (c->params.max_refl + 1) + 1)); RTS_HIP(c->d_hit_t.reserve((size_t)n * (c->params.max_refl + 1) + 1));
        rts_fill_i32(st, c->d_hit_prim.p, -2, (size_t)n * (c->params.max_refl + 1));
        RTS_HIP(hipMemsetAsync(c->d_hit_t.p, 0, sizeof(float) * (size_t)n * (c->params.max_refl + 1), st));
    }
    a.pmask = lc.mask.n ? pmask : nullptr; a.pre_filter = pre_filter ? 1u : 0u;
    a.nodes4 = c->scene->d_nodes4.p; a.nodes4v = (c->node_versions && !c->async_idle0) ? c->scene->d_nodes4v.p : nullptr; a.stack_lds = c->stack_lds; a.leaves = c->d_leaves.p; a.tri_nidx = c->scene->d_tri_nidx.p; a.normals = c->d_normals_world.p;
    a.targets = c->p_targets; a.rx = c->d_rx.p;
    a.recv_records = c->d_recv.p; a.all_records = c->d_all.p; a.counters = c->p_counters; a.block_counters = c->d_block_counters.p; a.dir_hist = c->d_dir_hist.p;
    a.hit_prim = c->d_hit_prim.p; a.hit_t = c->d_hit_t.p; a.stack_ovf = c->d_stack_ovf.p; a.child = c->d_child.p;
    {   // longest-tile-first order from what this handle's earlier launches measured per global tile (rts_post.hip)
        const int lpt = c->tile_lpt ? 1 : 0;
        const uint32_t n_tiles = (n + RTS_WTILE - 1) / RTS_WTILE;
        const uint64_t sig[4] = {n, first, ((uint64_t)il_parts << 32) | il_tile, il_part};
        const bool aligned = first % RTS_WTILE == 0 && (il_parts <= 1 || il_tile % RTS_WTILE == 0);
        a.rx_window_screen = c->rx_window_screen ? 1u : 0u;
        a.coop_versions = c->coop_versions ? 1u : 0u;
        a.batch_dead = aligned ? (uint32_t)c->batch_dead : 0u;      // (a wave tile must be 64 CONSECUTIVE launch indices for the tile-level screen: rts_tile_maybe)
        const uint32_t n_hist = (uint32_t)((total + RTS_WTILE - 1) / RTS_WTILE);
        RTS_HIP(c->d_tile_ctr.reserve(RTS_ZERO_WORDS + RTS_MASK_WORDS + 64)); c->p_counters = reinterpret_cast<unsigned long long*>(c->d_tile_ctr.p + RTS_OFF_COUNTERS); a.counters = c->p_counters;
        a.tile_ctr = c->d_tile_ctr.p;
        if (lpt && aligned && n_tiles > grid * (RTS_BLOCK / RTS_WTILE)) {
            if (c->hist->n != n_hist) {
                std::lock_guard<std::mutex> lk(g_hist_mu);
                RTS_HIP(hipDeviceSynchronize()); RTS_HIP(c->hist->d.reserve(n_hist)); RTS_HIP(hipMemset(c->hist->d.p, 0, sizeof(uint32_t) * n_hist));      // (blocking, once: the table may be shared with handles on other streams)
                c->hist->n = n_hist; c->hist->any = false; c->tile_cost_pending = false;
            }
            if (c->tile_cost_pending || c->hist->any) {
                c->coop_big_now = (il_parts > 1 && !shared_gpu && c->coop_big_part > c->coop_big) ? c->coop_big_part : c->coop_big;
                int rc = rts_tile_order_build(c, c->tile_cost_sig, c->tile_cost_pending, sig, n_tiles, grid * (RTS_BLOCK / RTS_WTILE)); if (rc != RTS_OK) return rc;
                a.xcd_seg = c->xcd_affine_now ? c->d_xcd.p : nullptr;
                a.tile_order = c->d_tile_order.p; a.tile_head = c->coop_frac > 0.0 ? c->d_tile_ctr.p + RTS_OFF_HEAD + 2 : nullptr; a.tile_head_all = a.tile_head; c->hist->any = true;
                a.tile_live = c->d_tile_ctr.p + RTS_OFF_LIVE;      // (written by the order build when it counts bins; else it stays at the fill's 0 = unknown)
            }
            const bool merged_all = c->tile_cost_pending && (c->tile_cost_sig[0] + RTS_WTILE - 1) / RTS_WTILE >= n_tiles && c->d_tile_cost.cap >= n_tiles;      // k_tile_merge read AND cleared the records
            RTS_HIP(c->d_tile_cost.reserve(n_tiles));
            if (!merged_all) RTS_HIP(hipMemsetAsync(c->d_tile_cost.p, 0, sizeof(uint32_t) * n_tiles, st));
            a.tile_cost = c->d_tile_cost.p;
            c->tile_cost_pending = true; memcpy(c->tile_cost_sig, sig, sizeof(sig)); memcpy(c->tile_last_sig, sig, sizeof(sig)); c->tile_last_valid = true;
        } else { c->tile_cost_pending = false; c->tile_last_valid = false; }      // (costs of an unaligned or single-sweep launch are not recorded)
    }
    const char* tl_path = count_trav ? getenv("RTS_TIMELINE") : nullptr;      // debug: dump the block/tile timeline of this launch
    const size_t cnt_tl = (size_t)grid * 2 + 2 * (size_t)((n + RTS_WTILE - 1) / RTS_WTILE);          // [grid][2] block ticks, [tiles] durations, [tiles] start ticks
    if (tl_path) { RTS_HIP(c->d_timeline.reserve(cnt_tl + 1)); RTS_HIP(hipMemsetAsync(c->d_timeline.p, 0, sizeof(unsigned long long) * (cnt_tl + 1), st)); a.timeline = c->d_timeline.p; }
    c->tl_blocks = 0;
    if (!tl_path && !count_trav && c->timeline_blocks) {      // debug, product builds: when every block of this launch started and ended (printed by rts_trace_pulse_end)
        RTS_HIP(c->d_timeline.reserve((size_t)grid * 2 + 1)); RTS_HIP(hipMemsetAsync(c->d_timeline.p, 0, sizeof(unsigned long long) * ((size_t)grid * 2 + 1), st)); a.timeline = c->d_timeline.p; c->tl_blocks = grid;
    }
    c->last_args = a;

    lt.lap(3);
    // ---- trace
    RTS_STAGE(c, "pre-trace");
    // The trace kernel's stream: the handle's low-priority trace stream when other pulses share the GPU (their short kernels then get
    // in beside it), the handle's OWN stream when none does -- a pulse on its own is one dependent chain, and every hop between two
    // streams costs it 17-25 us of event hand-over (profiles/r04_inflight1_pulse_timeline.log: two hops per pulse)
    c->tstream_now = (c->trace_own_stream && !shared_gpu && c->gate->refs.load() == 1) ? st : c->tstream;
    if (c->tstream_now != st) {
        RTS_HIP(hipEventRecord(c->ev[8], st));                   // scene + per-pulse buffers of this handle are ready
        RTS_HIP(hipStreamWaitEvent(c->tstream_now, c->ev[8], 0));
    }
    RTS_HIP(hipEventRecord(c->ev[2], c->tstream_now));
    // the cooperative kernel (tiles at the head of the cost order, one launch index per wave): its grid follows the head count
    // of the handle's previous order build (the count of THIS build is on the device; a grid too small or too large only costs
    // balance, every unit is drawn from a queue); it came home with that launch's counters
    // Whether there is a cooperative kernel at all is decided HERE, from that earlier count (both kernels must agree on
    // who traces the head of the order): no head last time -> none now, and the ordinary kernel traces every tile.
    unsigned coop_grid = 0;
    if (a.tile_head) {
        if (c->hist->head_hint_valid) c->n_head_hint = c->hist->head_hint;      // (the latest head count of ANY handle that shares the history)
        const bool grouped = a.nodes4v && a.coop_versions && !keep_all && !a.max_refr;      // (rts_trace_dispatch: the cooperative kernel that walks the octant versions holds 64 / RTS_COOP_GROUP rays per unit)
        const uint64_t units = (grouped ? (uint64_t)RTS_COOP_GROUP : 64ULL) * c->n_head_hint;
        if (units == 0) a.tile_head = nullptr;
        else coop_grid = (unsigned)std::min<uint64_t>(c->coop_grid_max, std::max<uint64_t>(16, (units + 3) / 4));
        c->last_args = a;
    }
    if (c->debug_coop) fprintf(stderr, "[rts] launch: n_rays %u grid %u head hint %u coop grid %u walk-steps threshold %u min cost %u order %d\n", n, grid, c->n_head_hint, coop_grid, a.coop_walk_steps, a.coop_min_cost, a.tile_order ? 1 : 0);
    c->last_coop_grid = coop_grid;
    int rc = rts_trace_launch(c, a, count_trav, coop_grid);
    if (rc != RTS_OK) return rc;
    RTS_STAGE(c, "k_trace");
    RTS_HIP(hipEventRecord(c->ev[3], c->tstream_now));
    if (c->tstream_now != st) RTS_HIP(hipStreamWaitEvent(st, c->ev[3], 0));      // everything later on this handle's stream follows its trace
    lt.lap(4);
    // (the eight counters were written into the pinned block by k_sum_counters itself: no copy)
    c->pulse_open = true; g_open_pulses[c->device & 63]++;
    if (tl_path) {
        RTS_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(cnt_tl + 2);
        h[0] = grid; h[1] = (n + RTS_WTILE - 1) / RTS_WTILE;
        RTS_HIP(hipMemcpy(h.data() + 2, c->d_timeline.p, sizeof(unsigned long long) * cnt_tl, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(tl_path, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
    return RTS_OK;
}

// Waits for the pulse begun on this handle, then orders + expands its received rays (left in flight).
// counters of a finished launch -> the handle's statistics and the hints its next launch uses
static void rts_pulse_account(RtsContext* c, const unsigned long long* cnt)
{
    const uint32_t n = c->n_rays;
    RtsStats& s = c->stats;
    s.rays = n; s.segments = cnt[1]; s.shaded = cnt[2]; s.received = cnt[0]; s.node_visits = cnt[3]; s.tri_tests = cnt[4]; s.stack_overflows = (uint32_t)cnt[5];
    s.n_prims = c->scene->n_prims; s.n_nodes = c->scene->n_nodes;
    s.walked_segments = cnt[11]; s.cost_records_dropped = (uint32_t)cnt[12];
    if (c->debug_coop && (cnt[12] || cnt[14])) fprintf(stderr, "[rts] clocks: %llu cost records dropped; shader clock ran backwards on %llu tiles (XCC mask 0x%llx)\n", cnt[12], cnt[14] & 0x00ffffffffffffffULL, cnt[14] >> 56);
    s.coop_tiles = c->last_coop_grid ? (uint32_t)std::min<unsigned long long>(std::min<unsigned long long>(cnt[7], (n + RTS_WTILE - 1) / RTS_WTILE), 16384ull) : 0u;      // (the bounds k_trace applies to the order's head count)
    c->pre_dense = 2 * s.shaded > (uint64_t)n;                      // next launch of this handle: pre-filter only if most launch indices hit nothing
    s.ms_scene = s.ms_trace = s.ms_compact = s.ms_aggregate = 0;
    c->stats_pending = true;
    c->recv_hint = cnt[0]; c->recv_hint_valid = true;
}

extern "C" int rts_trace_pulse_end(RtsHandle c)
{
    CHECK_HANDLE(c);
    if (!c->pulse_open) { rts_set_error("rts_trace_pulse_end: no pulse in flight on this handle"); return RTS_ERR_INVALID; }
    c->pulse_open = false; g_open_pulses[c->device & 63]--;
    hipStream_t st = c->stream;
    const bool keep_all = (c->params.flags & RTS_FLAG_KEEP_ALL_RAYS) != 0;
    unsigned long long* cnt = c->pin->cnt;
    RTS_HIP(rts_stream_wait(c, st));                // the one host sync of the launch: the received count sizes what follows
    if (cnt[13]) { rts_set_error("rts_trace_pulse: %llu counter rows of the launch were never written by their blocks (counting build)", cnt[13]); return RTS_ERR_HIP; }
    if (cnt[6]) { rts_set_error("rts_trace_pulse: traversal stack overflow / malformed BVH guard tripped on %llu waves", cnt[6]); return RTS_ERR_HIP; }
    c->n_recv = cnt[0]; c->n_head_hint = (uint32_t)cnt[7]; c->hist->head_hint = c->n_head_hint; c->hist->head_hint_valid = true;
    if (c->tl_blocks) {      // RTS_TIMELINE_BLOCKS: the launch as a bulk (every block resident) and a tail (rts_get_block_timeline; profiles/r05h_batch_launch.log)
        std::vector<unsigned long long> h((size_t)c->tl_blocks * 2);
        (void)hipMemcpy(h.data(), c->d_timeline.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> s0, s1; for (uint32_t i = 0; i < c->tl_blocks; i++) { s0.push_back(h[2 * i]); s1.push_back(h[2 * i + 1]); }
        std::sort(s0.begin(), s0.end()); std::sort(s1.begin(), s1.end());
        const size_t m = s0.size(); const unsigned long long t0 = s0[0];
        const size_t q[8] = {0, m / 2, m - 1, 0, m / 10, m / 2, m * 9 / 10, m - 1};
        for (int k = 0; k < 8; k++) c->tl_summary[k] = (double)((k < 3 ? s0[q[k]] : s1[q[k]]) - t0) / 100.0;      // us after the first block's start (100 MHz counter)
        c->tl_summary[8] = (double)m;
        if (c->debug_coop) fprintf(stderr, "[rts] blocks of handle %p (us after the first start): start p50 %.1f max %.1f | end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", (void*)c,
                                   c->tl_summary[1], c->tl_summary[2], c->tl_summary[3], c->tl_summary[4], c->tl_summary[5], c->tl_summary[6], c->tl_summary[7]);
    }
    if (c->debug_coop && c->d_xcd.p && c->d_tile_ctr.p) {      // debug: what the head rule of this launch's order build saw (blocking read-backs)
        unsigned long long sums[2] = {0, 0};
        (void)hipMemcpy(&sums[0], c->d_xcd.p + 32, sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&sums[1], c->d_tile_ctr.p + RTS_OFF_HEAD, sizeof(unsigned long long), hipMemcpyDeviceToHost);
        uint32_t live_w = 0; (void)hipMemcpy(&live_w, c->d_tile_ctr.p + RTS_OFF_LIVE, sizeof(uint32_t), hipMemcpyDeviceToHost);
        fprintf(stderr, "[rts] end: handle %p head count %llu coop grid %u | cost sum persisted %llu, this build's %llu | live word %u of %u tiles\n", (void*)c, cnt[7], c->last_coop_grid, sums[0], sums[1], live_w, (c->n_rays + RTS_WTILE - 1) / RTS_WTILE);
        if (c->tile_cost_pending && c->d_tile_cost.p) {          // the cost records this launch wrote (merged by the next order build)
            const uint32_t nt = (uint32_t)((c->tile_cost_sig[0] + RTS_WTILE - 1) / RTS_WTILE);
            std::vector<uint32_t> h(nt);
            (void)hipMemcpy(h.data(), c->d_tile_cost.p, sizeof(uint32_t) * nt, hipMemcpyDeviceToHost);
            unsigned long long sum = 0, big = 0, flagged = 0; uint32_t mx = 0, shown = 0;
            for (uint32_t j = 0; j < nt; j++) { const uint32_t v = h[j] & 0x3fffffffu; sum += v; if (v > mx) mx = v; if (h[j] >> 31) flagged++; if (v > (1u << 26)) { big++; if (shown++ < 6) fprintf(stderr, "[rts]    tile %u record 0x%08x\n", j, h[j]); } }
            fprintf(stderr, "[rts]    records of this launch: sum %llu max %u, %llu above 2^26, %llu flagged LONG WALKS\n", sum, mx, big, flagged);
        }
    }

    // ---- order + expand the received rays (and the keep-all buffers); left in flight on the stream
    RTS_HIP(hipEventRecord(c->ev[4], st));
    int rc = rts_post_order_and_expand(c); if (rc != RTS_OK) return rc;
    if (keep_all) { rc = rts_post_expand_all(c); if (rc != RTS_OK) return rc; }
    if (c->mirror.want) { rc = rts_post_mirror_received(c); if (rc != RTS_OK) return rc; }
    RTS_HIP(hipEventRecord(c->ev[5], st));

    c->agg_timed = false; c->fin_timed = false;
    rts_pulse_account(c, cnt);
    return RTS_OK;
}

extern "C" int rts_get_stats(RtsHandle c, RtsStats* out)
{
    if (!c || !out) { rts_set_error("rts_get_stats: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    if (c->stats_pending) {                          // stage timers are resolved lazily: reading them drains the stream
        RTS_HIP(hipSetDevice(c->device));
        RTS_HIP(hipStreamSynchronize(c->stream));
        RtsStats& s = c->stats; float ms = 0;
        RTS_HIP(hipEventElapsedTime(&s.ms_scene, c->ev[0], c->ev[1]));
        RTS_HIP(hipEventElapsedTime(&s.ms_trace, c->ev[2], c->ev[3]));
        RTS_HIP(hipEventElapsedTime(&s.ms_compact, c->ev[4], c->ev[5]));
        s.ms_aggregate = 0;
        if (c->fin_timed || c->agg_timed) { RTS_HIP(hipEventElapsedTime(&ms, c->ev[6], c->ev[7])); s.ms_aggregate = ms; }
        c->stats_pending = false;
    }
    *out = c->stats; return RTS_OK;
}

// Lane statistics of the last launch's walk (RTS_FLAG_COUNT_TRAVERSAL builds; zeros otherwise): out[0] lane-steps ISSUED (64 x the
// longest walk of every bounce round of every tile), out[1] of them issued to lanes that were in the round at all, out[2]
// walk steps actually taken.  1 - out[1]/out[0]: what lanes that left their tile early cost (re-packing rays between rounds could
// recover at most this); (out[1] - out[2])/out[0]: what waiting for the round's slowest lane costs.
// When the persistent blocks of the handle's last launch started and ended (a handle created with RTS_TIMELINE_BLOCKS=1; product builds): out[0..2] first / median / last
// block START, out[3..7] first / 10th percentile / median / 90th percentile / last block END, all in microseconds after the first start (the 100 MHz counter), out[8] the blocks.
// A launch is a BULK -- every block resident; the median end -- and a tail of the few tiles that are one ray's long chain of dependent steps.
extern "C" int rts_get_block_timeline(RtsHandle c, double* out, uint32_t n)
{
    if (!c || !out) { rts_set_error("rts_get_block_timeline: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    if (!c->timeline_blocks || c->tl_summary[8] == 0.0) { rts_set_error("rts_get_block_timeline: the handle records no block timeline (create it with RTS_TIMELINE_BLOCKS=1, product build) or has not traced yet"); return RTS_ERR_INVALID; }
    for (uint32_t k = 0; k < n && k < 9; k++) out[k] = c->tl_summary[k];
    return RTS_OK;
}

extern "C" int rts_get_lane_stats(RtsHandle c, uint64_t* out3)
{
    if (!c || !out3) { rts_set_error("rts_get_lane_stats: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    for (int k = 0; k < 3; k++) out3[k] = c->pin->cnt[8 + k];
    return RTS_OK;
}

// ... and the split VERDICT r4 #3 asked for: out[0..2] as rts_get_lane_stats, out[3] segments that walked at all, out[4] lane-steps issued to
// lanes that are in their tile's bounce round but never started a walk in it (a primary the pre-filter or every bounding sphere cleared, in a
// tile other lanes of which walk): out[4] / out[0] is what packing live launch indices of several tiles into dense waves could recover at most;
// (out[1] - out[2] - out[4]) / out[0] what waiting for the round's slowest WALKING lane costs.  n: capacity of out (<= 5 values are written).
extern "C" int rts_get_walk_stats(RtsHandle c, uint64_t* out, uint32_t n)
{
    if (!c || !out) { rts_set_error("rts_get_walk_stats: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    const int src[5] = {8, 9, 10, 11, 15};
    for (uint32_t k = 0; k < n && k < 5u; k++) out[k] = c->pin->cnt[src[k]];
    return RTS_OK;
}

extern "C" int rts_received_count(RtsHandle c, uint64_t* count)
{
    if (!c || !count) { rts_set_error("rts_received_count: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    *count = c->n_recv; return RTS_OK;
}

extern "C" int rts_get_received(RtsHandle c, PerRayData* rays, int32_t* paths, double* rcs_angles, uint64_t* slots, uint64_t capacity)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    const uint64_t R = c->n_recv; const uint32_t D = c->depth;
    if (capacity < R) { rts_set_error("rts_get_received: capacity %llu < %llu received rays", (unsigned long long)capacity, (unsigned long long)R); return RTS_ERR_CAPACITY; }
    if (R == 0) return RTS_OK;
    RTS_HIP(hipStreamSynchronize(c->stream));
    if (rays) RTS_HIP(hipMemcpy(rays, c->d_rx_rays.p, sizeof(PerRayData)*R, hipMemcpyDeviceToHost));
    if (paths && D) RTS_HIP(hipMemcpy(paths, c->d_rx_paths.p, sizeof(int32_t)*R*D, hipMemcpyDeviceToHost));
    if (rcs_angles && D) RTS_HIP(hipMemcpy(rcs_angles, c->d_rx_angles.p, sizeof(double)*2*R*D, hipMemcpyDeviceToHost));
    if (slots) RTS_HIP(hipMemcpy(slots, c->d_rx_slots.p, sizeof(uint64_t)*R, hipMemcpyDeviceToHost));
    return RTS_OK;
}

extern "C" int rts_get_all_rays(RtsHandle c, PerRayData* results, int32_t* targ_intersect, double* rcs_angle, int32_t* hit_prim, float* hit_t, uint64_t capacity)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!(c->params.flags & RTS_FLAG_KEEP_ALL_RAYS)) { rts_set_error("rts_get_all_rays: handle was not created with RTS_FLAG_KEEP_ALL_RAYS"); return RTS_ERR_INVALID; }
    const uint64_t n1 = c->n_rays, n = n1 * c->last_args.rows; const uint32_t D = c->depth, H = c->params.max_refl + 1;
    if (capacity < n) { rts_set_error("rts_get_all_rays: capacity too small (%llu rows)", (unsigned long long)n); return RTS_ERR_CAPACITY; }
    if (n == 0) return RTS_OK;
    RTS_HIP(hipStreamSynchronize(c->stream));
    if (results) RTS_HIP(hipMemcpy(results, c->d_all_rays.p, sizeof(PerRayData)*n, hipMemcpyDeviceToHost));
    if (targ_intersect && D) RTS_HIP(hipMemcpy(targ_intersect, c->d_all_paths.p, sizeof(int32_t)*n*D, hipMemcpyDeviceToHost));
    if (rcs_angle && D) RTS_HIP(hipMemcpy(rcs_angle, c->d_all_angles.p, sizeof(double)*2*n*D, hipMemcpyDeviceToHost));
    if (hit_prim) RTS_HIP(hipMemcpy(hit_prim, c->d_hit_prim.p, sizeof(int32_t)*n1*H, hipMemcpyDeviceToHost));
    if (hit_t) RTS_HIP(hipMemcpy(hit_t, c->d_hit_t.p, sizeof(float)*n1*H, hipMemcpyDeviceToHost));
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- finalise + aggregate
extern "C" int rts_finalise_uniform(RtsHandle c, const double* rcs_per_target, double wavelength, double gt, double gr, double carrier, double cspeed)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    std::vector<double> ones;
    if (!rcs_per_target) { ones.assign(c->scene->meshes.size() + 1, 1.0); rcs_per_target = ones.data(); }
    RTS_HIP(hipEventRecord(c->ev[6], c->stream));
    int rc = rts_post_finalise(c, rcs_per_target, wavelength, gt, gr, carrier, cspeed); if (rc != RTS_OK) return rc;
    RTS_HIP(hipEventRecord(c->ev[7], c->stream));
    c->fin_timed = true; c->stats_pending = true;
    c->agg_valid = false;      // (the mirror keeps the set AS RECEIVED for the rest of the pulse: rts_received_view)
    return RTS_OK;
}

static int rts_aggregate_impl(RtsContext* c, double cspeed, double carrier, uint64_t recv_index_base)
{
    const uint64_t R = c->n_recv;
    c->agg_pending.valid = false;                                       // (an unread table of an earlier call is dropped)
    c->groups.clear(); c->recv_index_base = recv_index_base;
    if (R == 0) { c->agg_valid = true; return RTS_OK; }
    RTS_HIP(c->d_delay.reserve(R)); RTS_HIP(c->d_phase.reserve(R)); RTS_HIP(c->d_pathmatch.reserve(R));
    if (!c->fin_timed) RTS_HIP(hipEventRecord(c->ev[6], c->stream));
    c->agg_delay_in = false;                                            // (delay / phase sums start at zero: no fills)
    const int32_t max_path = (int32_t)c->scene->meshes.size() - 1, max_rx = c->n_rx ? (int32_t)c->n_rx - 1 : 0;
    const bool use_rows = recv_index_base == RTS_BASE_USE_ROWS;
    c->agg_base_local = use_rows ? 0 : (int64_t)recv_index_base;
    int rc = rts_aggregate_device(c, max_path, max_rx, c->d_rx_paths.p, R, c->depth, cspeed, carrier, use_rows ? 0 : recv_index_base, c->d_rx_rays.p,
                                  c->d_delay.p, c->d_phase.p, c->d_pathmatch.p, &c->groups, nullptr, nullptr, nullptr, INT32_MAX, use_rows ? c->d_rx_slots.p : nullptr);
    c->agg_delay_in = true;
    if (rc != RTS_OK) return rc;
    // (the rays' power / Doppler on the device are the group values now; the mirror still holds the set as it was received, and rts_received_view keeps serving it)
    if (c->mirror.want) { rc = rts_post_mirror_aggregated(c); if (rc != RTS_OK) return rc; }
    RTS_HIP(hipEventRecord(c->ev[7], c->stream));
    c->agg_timed = true; c->stats_pending = true;
    c->agg_valid = true;
    return RTS_OK;
}

extern "C" int rts_aggregate(RtsHandle c, double cspeed, double carrier, uint64_t recv_index_base)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    return rts_aggregate_impl(c, cspeed, carrier, recv_index_base);
}

// ------------------------------------------------------------------------------------- a pulse's post-processing without the host in it
// rts_trace_pulse_end_uniform = rts_trace_pulse_end + rts_finalise_uniform (+ rts_cube_accumulate) + rts_aggregate for a caller that
// needs no host callbacks between them ("device-side received count feeding the post kernels").  When the handle's
// previous pulse received no more than 3/4 of what the one-block ordering kernels take (4 096 rays with 32-bit sort keys, 2 048 with
// 64-bit ones) the whole chain is ENQUEUED behind the trace at once, sized for that capacity; its kernels read the received count the trace kernel left on the device and do nothing if it exceeds that
// capacity.  The count, the statistics and the group table come home with the first call that asks (rts_received_count,
// rts_get_stats, rts_group_count, ...; the handle's next rts_trace_pulse_begin at the latest): if the count did exceed the
// capacity the chain is run again then, the ordinary way.  Otherwise -- no history yet, KEEP_ALL, a large received set -- the
// call is the four calls it stands for.
static int rts_post_chain(RtsContext* c, bool ordered = false)      // ordered: rts_trace_pulse_end has ordered + expanded the received set already (its events ev[4] / ev[5] stand)
{
    const RtsSpecParams& q = c->spec;
    hipStream_t st = c->stream;
    const bool keep_all = (c->params.flags & RTS_FLAG_KEEP_ALL_RAYS) != 0;
    int rc = RTS_OK;
    // (one block does in 150 us what seven launches -- four of them many blocks wide -- do in 92 us + six launch gaps for BASELINE configs[2]'s
    // ~1 900 received rays; for a few hundred rays it is the other way round: sequential pulses, one kernel against seven, C3 1.025 / 0.980 ms,
    // C2 (400 rays) 0.339 / 0.354, configs[4] (100) 0.955 / 0.978, profiles/r04_post_one_ab.log -- so the choice follows the handle's last count)
    if (!ordered && q.mode == 0 && c->recv_dev && c->post_one && c->recv_hint_valid && c->recv_hint <= c->post_one_max && c->post_small && !keep_all && !c->mirror.want && c->n_recv <= c->spec_cap) {
        // the speculative chain as ONE kernel (rts_post.hip: k_post_all): sized for the capacity, the count from the device
        RTS_HIP(hipEventRecord(c->ev[4], st)); RTS_HIP(hipEventRecord(c->ev[5], st)); RTS_HIP(hipEventRecord(c->ev[6], st));
        c->agg_pending.valid = false; c->groups.clear();
        rc = rts_post_all_small(c, (uint32_t)c->n_recv, q, true); if (rc != RTS_OK) return rc;
        RTS_HIP(hipEventRecord(c->ev[7], st));
        c->fin_timed = true; c->agg_timed = true; c->stats_pending = true; c->agg_valid = true;
        return RTS_OK;
    }
    if (!ordered) {
        RTS_HIP(hipEventRecord(c->ev[4], st));
        rc = rts_post_order_and_expand(c); if (rc != RTS_OK) return rc;
        if (keep_all) { rc = rts_post_expand_all(c); if (rc != RTS_OK) return rc; }
        if (c->mirror.want) { rc = rts_post_mirror_received(c); if (rc != RTS_OK) return rc; }
        RTS_HIP(hipEventRecord(c->ev[5], st));
    }
    if (q.mode == 1) return RTS_OK;                                     // rts_received_prefetch: the received set goes home, the caller finalises it
    RTS_HIP(hipEventRecord(c->ev[6], st));
    rc = rts_post_finalise(c, q.rcs.data(), q.wl, q.gt, q.gr, q.carrier, q.cspeed); if (rc != RTS_OK) return rc;
    c->fin_timed = true; c->agg_valid = false;
    if (q.cube_pulse >= 0) { rc = rts_cube_accumulate_device(c, (uint32_t)q.cube_pulse, q.cspeed, q.carrier); if (rc != RTS_OK) return rc; }
    rc = rts_aggregate_impl(c, q.cspeed, q.carrier, q.base);
    return rc;
}

static int rts_spec_resolve(RtsContext* c)
{
    if (!c->spec_pending) return RTS_OK;
    c->spec_pending = false; g_open_pulses[c->device & 63]--;
    RTS_HIP(hipSetDevice(c->device));
    const unsigned long long* cnt = c->pin->cnt;
    RTS_HIP(rts_stream_wait(c, c->stream));
    if (cnt[13]) { rts_set_error("rts_trace_pulse: %llu counter rows of the launch were never written by their blocks (counting build)", cnt[13]); return RTS_ERR_HIP; }      // (as rts_trace_pulse_end does: ADVICE r4)
    if (cnt[6]) { rts_set_error("rts_trace_pulse: traversal stack overflow / malformed BVH guard tripped on %llu waves", cnt[6]); return RTS_ERR_HIP; }
    c->n_recv = cnt[0]; c->n_head_hint = (uint32_t)cnt[7]; c->hist->head_hint = c->n_head_hint; c->hist->head_hint_valid = true;
    c->recv_hint = cnt[0]; c->recv_hint_valid = true;                   // (the next pulse's choices -- speculate at all, one kernel or seven -- follow THIS pulse's count, not the handle's first)
    rts_pulse_account(c, cnt);
    if (c->n_recv > c->spec_cap) {                                      // more rays than the speculative chain was sized for: it did nothing; the ordinary chain now
        c->agg_pending.valid = false;
        return rts_post_chain(c);
    }
    if (c->n_recv == 0) { c->agg_pending.valid = false; c->groups.clear(); c->agg_valid = true; return RTS_OK; }
    c->agg_pending.R = (uint32_t)c->n_recv; c->agg_pending.spec = std::min<uint32_t>((uint32_t)c->n_recv, RTS_PIN_GROUPS);
    return RTS_OK;
}

extern "C" int rts_trace_pulse_end_uniform(RtsHandle c, const double* rcs_per_target, double wavelength, double gt, double gr, double carrier, double cspeed,
                                           int32_t cube_pulse, uint64_t recv_index_base)
{
    CHECK_HANDLE(c);
    if (!c->pulse_open) { rts_set_error("rts_trace_pulse_end_uniform: no pulse in flight on this handle"); return RTS_ERR_INVALID; }
    if (cube_pulse >= 0 && (!c->cube_set || (uint32_t)cube_pulse >= c->cube_params.n_pulses)) { rts_set_error("rts_trace_pulse_end_uniform: no cube attached, or pulse %d outside it", cube_pulse); return RTS_ERR_INVALID; }
    RtsSpecParams& q = c->spec;
    const size_t nt = c->scene->meshes.size();
    q.rcs.assign(nt + 1, 1.0); if (rcs_per_target) for (size_t t = 0; t < nt; t++) q.rcs[t] = rcs_per_target[t];
    q.wl = wavelength; q.gt = gt; q.gr = gr; q.carrier = carrier; q.cspeed = cspeed; q.cube_pulse = cube_pulse; q.base = recv_index_base; q.mode = 0;
    const bool keep_all = (c->params.flags & RTS_FLAG_KEEP_ALL_RAYS) != 0;
    bool narrow_key = true;
    {   // capacity of a speculative chain: the smaller of its two one-block sorts (row keys: 32 bits without refraction chains; (receiver, path) keys: D x B + RXB bits)
        uint32_t B = 1; while (((uint64_t)1 << B) < (uint64_t)(c->scene->meshes.size() + 1)) B++;
        uint32_t RXB = 1; while (((uint64_t)1 << RXB) < (uint64_t)std::max<uint32_t>(c->n_rx, 1u)) RXB++;
        const uint32_t key_bits = (c->depth ? c->depth * B : 0u) + RXB;
        c->spec_cap = (c->last_args.max_refr == 0 && key_bits < 32u) ? RTS_SMALL_CAP32 : RTS_SMALL_CAP64;
        // a key beyond 64 bits (e.g. 16 bounces among >= 8 targets) is sorted as two or three words by the GENERAL chain, whose
        // kernels take the received count from the host (rts_post.hip: only the one-block path reads it on the device): such a
        // handle never speculates
        narrow_key = key_bits <= 64u;
    }
    const bool speculate = c->post_small && c->spec_enabled && narrow_key && !keep_all && c->n_rays > 0 && c->recv_hint_valid && c->recv_hint <= ((uint64_t)c->spec_cap * 3ull) / 4ull;
    if (!speculate) {
        int rc = rts_trace_pulse_end(c); if (rc != RTS_OK) return rc;
        return rts_post_chain(c, true);
    }
    c->pulse_open = false;                                              // (the pulse stays counted as open on its device until it is resolved)
    c->spec_pending = true;
    c->agg_timed = false; c->fin_timed = false;
    c->n_recv = c->spec_cap; c->recv_dev = c->p_counters;              // sizes for the capacity, the count itself from the device
    // ... on the TRACE stream, behind the trace kernel: enqueued on the handle's other stream -- which waits for the trace through
    // an event -- every launch call of the chain blocked (0.22 ms per pulse in the submitting thread)
    int rc;
    if (c->spec_on_trace_stream && c->tstream_now != c->stream) {
        hipStream_t own = c->stream; c->stream = c->tstream_now;
        rc = rts_post_chain(c);
        RTS_HIP(hipEventRecord(c->ev_spec, c->tstream_now)); c->stream = own;
        RTS_HIP(hipStreamWaitEvent(c->stream, c->ev_spec, 0));          // (what the handle enqueues next on its own stream comes after the chain)
    } else rc = rts_post_chain(c);                                      // (the handle's own stream already waits for the trace: rts_trace_pulse_begin)
    c->recv_dev = nullptr; c->n_recv = 0;
    if (rc != RTS_OK) { c->spec_pending = false; g_open_pulses[c->device & 63]--; return rc; }
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- the received set at home without copy calls
// rts_received_prefetch: for a caller that needs the received rays on the HOST (the simulator's RCS / gain callbacks,
// ray_tracer.cpp:1198-1256).  Behind the pulse's trace -- and, when the handle's previous pulse received few rays, without
// waiting for it: the kernels take the count from the device, like rts_trace_pulse_end_uniform's chain -- the received set is
// ordered, expanded and STORED BY A KERNEL into the handle's pinned host mirror; rts_received_view then waits once and hands out
// pointers into it.  No history yet, a large set, KEEP_ALL: the call only marks the pulse, and the ordinary (blocking)
// rts_trace_pulse_end that the first accessor runs feeds the mirror.
extern "C" int rts_received_prefetch(RtsHandle c)
{
    CHECK_HANDLE(c);
    if (!c->pulse_open) { rts_set_error("rts_received_prefetch: no pulse in flight on this handle"); return RTS_ERR_INVALID; }
    const bool keep_all = (c->params.flags & RTS_FLAG_KEEP_ALL_RAYS) != 0;
    const uint32_t cap = c->last_args.max_refr == 0 ? RTS_SMALL_CAP32 : RTS_SMALL_CAP64;
    { int rc = rts_mirror_reserve(c, cap); if (rc != RTS_OK) return rc; }
    c->mirror.want = true;
    const bool speculate = c->post_small && c->spec_enabled && !keep_all && c->n_rays > 0 && c->recv_hint_valid && c->recv_hint <= ((uint64_t)cap * 3ull) / 4ull;
    if (!speculate) return RTS_OK;
    c->spec.mode = 1; c->spec_cap = cap;
    c->pulse_open = false; c->spec_pending = true;                     // (the pulse stays counted as open on its device until it is resolved)
    c->agg_timed = false; c->fin_timed = false;
    c->n_recv = cap; c->recv_dev = c->p_counters;
    const int rc = rts_post_chain(c);
    c->recv_dev = nullptr; c->n_recv = 0;
    if (rc != RTS_OK) { c->spec_pending = false; g_open_pulses[c->device & 63]--; return rc; }
    return RTS_OK;
}

extern "C" int rts_received_view(RtsHandle c, const PerRayData** rays, const int32_t** paths, const double** rcs_angles, const uint64_t** slots, uint64_t* count)
{
    CHECK_HANDLE(c);
    if (!count) { rts_set_error("rts_received_view: null count"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    const uint64_t R = c->n_recv; const uint32_t D = c->depth;
    *count = R;
    if (rays) *rays = nullptr; if (paths) *paths = nullptr; if (rcs_angles) *rcs_angles = nullptr; if (slots) *slots = nullptr;
    if (R == 0) return RTS_OK;
    RTS_HIP(rts_stream_wait(c, c->stream));
    const RtsHostMirror& m = c->mirror;
    if (m.recv_valid && R <= m.cap) {
        if (rays) *rays = (const PerRayData*)(m.host + m.o_rays); if (paths) *paths = (const int32_t*)(m.host + m.o_paths);
        if (rcs_angles) *rcs_angles = (const double*)(m.host + m.o_angles); if (slots) *slots = (const uint64_t*)(m.host + m.o_slots);
        return RTS_OK;
    }
    // no mirror of this set (not asked for, or larger than the mirror): copies into storage the handle keeps.  Each array is read from the
    // device ONCE per pulse (v_recv_have): pointers handed out earlier stay valid and keep their content -- the records as they were at the
    // pulse's first call, i.e. AS RECEIVED when that call came before rts_finalise_values (a second call used to re-read records
    // the finalisation had changed, and rts_aggregated_view's fallback overwrote them with group values)
    if (rays) { if (!(c->v_recv_have & 1u)) { c->v_rays.resize(R); RTS_HIP(hipMemcpy(c->v_rays.data(), c->d_rx_rays.p, sizeof(PerRayData) * R, hipMemcpyDeviceToHost)); c->v_recv_have |= 1u; } *rays = c->v_rays.data(); }
    if (paths && D) { if (!(c->v_recv_have & 2u)) { c->v_paths.resize(R * D); RTS_HIP(hipMemcpy(c->v_paths.data(), c->d_rx_paths.p, sizeof(int32_t) * R * D, hipMemcpyDeviceToHost)); c->v_recv_have |= 2u; } *paths = c->v_paths.data(); }
    if (rcs_angles && D) { if (!(c->v_recv_have & 4u)) { c->v_angles.resize(2 * R * D); RTS_HIP(hipMemcpy(c->v_angles.data(), c->d_rx_angles.p, sizeof(double) * 2 * R * D, hipMemcpyDeviceToHost)); c->v_recv_have |= 4u; } *rcs_angles = c->v_angles.data(); }
    if (slots) { if (!(c->v_recv_have & 8u)) { c->v_slots.resize(R); RTS_HIP(hipMemcpy(c->v_slots.data(), c->d_rx_slots.p, sizeof(uint64_t) * R, hipMemcpyDeviceToHost)); c->v_recv_have |= 8u; } *slots = c->v_slots.data(); }
    return RTS_OK;
}

// The per-received-ray update of ray_tracer.cpp:1219-1253 when the factors come from the simulator's callbacks: the caller has
// formed every received ray's final power and Doppler shift on the host (from rts_received_view's records); they replace the
// traced values on the device, in received order.  Enqueued (the values are copied out of the caller's arrays before the call
// returns); rts_aggregate follows.
extern "C" int rts_finalise_values(RtsHandle c, const double* power, const double* doppler, uint64_t count)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (count != c->n_recv) { rts_set_error("rts_finalise_values: %llu values for %llu received rays", (unsigned long long)count, (unsigned long long)c->n_recv); return RTS_ERR_INVALID; }
    if (count == 0) return RTS_OK;
    if (!power || !doppler) { rts_set_error("rts_finalise_values: null array"); return RTS_ERR_INVALID; }
    RTS_HIP(hipEventRecord(c->ev[6], c->stream));
    RtsHostMirror& m = c->mirror;
    if (!m.host) { int rc = rts_mirror_reserve(c, RTS_SMALL_CAP32); if (rc != RTS_OK) return rc; }
    if (count <= m.cap) {
        // (the staging is free: what read it last -- this handle's previous pulse -- was waited for before this pulse was begun)
        memcpy(m.host + m.o_vpower, power, sizeof(double) * count); memcpy(m.host + m.o_vdoppler, doppler, sizeof(double) * count);
        int rc = rts_post_set_values(c, (const double*)(m.dev + m.o_vpower), (const double*)(m.dev + m.o_vdoppler)); if (rc != RTS_OK) return rc;
    } else {                                                            // a set beyond the mirror: blocking uploads into scratch the aggregation overwrites later
        RTS_HIP(c->d_delay.reserve(count)); RTS_HIP(c->d_phase.reserve(count));
        RTS_HIP(hipMemcpyAsync(c->d_delay.p, power, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        RTS_HIP(hipMemcpyAsync(c->d_phase.p, doppler, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
        int rc = rts_post_set_values(c, c->d_delay.p, c->d_phase.p); if (rc != RTS_OK) return rc;
        RTS_HIP(hipStreamSynchronize(c->stream));
    }
    RTS_HIP(hipEventRecord(c->ev[7], c->stream));
    c->fin_timed = true; c->stats_pending = true;
    c->agg_valid = false;      // (the mirror keeps the set AS RECEIVED for the rest of the pulse: rts_received_view)
    return RTS_OK;
}

// Per-ray outputs of the last rts_aggregate (what rs::kernel_wrapper leaves in h_rx_results_arr[].power / .doppler, h_delay_arr,
// h_phase_arr, h_pathMatch): pointers into the host mirror when the pulse was prefetched (the call waits for the handle's stream
// once), copies otherwise.  Valid until the handle's next rts_trace_pulse_begin.
extern "C" int rts_aggregated_view(RtsHandle c, const double** power, const double** doppler, const double** delay, const double** phase, const int32_t** path_match, uint64_t* count)
{
    CHECK_HANDLE(c);
    if (!count) { rts_set_error("rts_aggregated_view: null count"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    if (!c->agg_valid) { rts_set_error("rts_aggregated_view: call rts_aggregate first"); return RTS_ERR_INVALID; }
    const uint64_t R = c->n_recv;
    *count = R;
    if (power) *power = nullptr; if (doppler) *doppler = nullptr; if (delay) *delay = nullptr; if (phase) *phase = nullptr; if (path_match) *path_match = nullptr;
    if (R == 0) return RTS_OK;
    RTS_HIP(rts_stream_wait(c, c->stream));
    const RtsHostMirror& m = c->mirror;
    if (m.agg_valid && R <= m.cap) {
        if (power) *power = (const double*)(m.host + m.o_apower); if (doppler) *doppler = (const double*)(m.host + m.o_adoppler);
        if (delay) *delay = (const double*)(m.host + m.o_adelay); if (phase) *phase = (const double*)(m.host + m.o_aphase); if (path_match) *path_match = (const int32_t*)(m.host + m.o_apm);
        return RTS_OK;
    }
    if (power || doppler) {
        c->v_agg_rays.resize(R); RTS_HIP(hipMemcpy(c->v_agg_rays.data(), c->d_rx_rays.p, sizeof(PerRayData) * R, hipMemcpyDeviceToHost));      // (scratch of its own: rts_received_view's records stay as they were)
        c->v_apower.resize(R); c->v_adoppler.resize(R);
        for (uint64_t i = 0; i < R; i++) { c->v_apower[i] = c->v_agg_rays[i].power; c->v_adoppler[i] = c->v_agg_rays[i].doppler; }
        if (power) *power = c->v_apower.data(); if (doppler) *doppler = c->v_adoppler.data();
    }
    if (delay) { c->v_adelay.resize(R); RTS_HIP(hipMemcpy(c->v_adelay.data(), c->d_delay.p, sizeof(double) * R, hipMemcpyDeviceToHost)); *delay = c->v_adelay.data(); }
    if (phase) { c->v_aphase.resize(R); RTS_HIP(hipMemcpy(c->v_aphase.data(), c->d_phase.p, sizeof(double) * R, hipMemcpyDeviceToHost)); *phase = c->v_aphase.data(); }
    if (path_match) { c->v_apm.resize(R); RTS_HIP(hipMemcpy(c->v_apm.data(), c->d_pathmatch.p, sizeof(int32_t) * R, hipMemcpyDeviceToHost)); *path_match = c->v_apm.data(); }
    return RTS_OK;
}

extern "C" int rts_group_count(RtsHandle c, uint32_t* count)
{
    if (!c || !count) { rts_set_error("rts_group_count: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    if (!c->agg_valid) { rts_set_error("rts_group_count: call rts_aggregate first"); return RTS_ERR_INVALID; }
    { int rc = rts_aggregate_fetch(c, &c->groups); if (rc != RTS_OK) return rc; }
    *count = (uint32_t)c->groups.size(); return RTS_OK;
}

extern "C" int rts_get_groups(RtsHandle c, RtsGroup* groups, uint32_t capacity)
{
    if (!c || (!groups && capacity)) { rts_set_error("rts_get_groups: null argument"); return RTS_ERR_INVALID; }
    CHECK_CLOSED(c);
    if (!c->agg_valid) { rts_set_error("rts_get_groups: call rts_aggregate first"); return RTS_ERR_INVALID; }
    { int rc = rts_aggregate_fetch(c, &c->groups); if (rc != RTS_OK) return rc; }
    if (capacity < c->groups.size()) { rts_set_error("rts_get_groups: capacity too small"); return RTS_ERR_CAPACITY; }
    if (!c->groups.empty()) memcpy(groups, c->groups.data(), sizeof(RtsGroup)*c->groups.size());
    return RTS_OK;
}

extern "C" int rts_get_aggregated(RtsHandle c, PerRayData* rays, double* delay, double* phase, int32_t* path_match, uint64_t capacity)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!c->agg_valid) { rts_set_error("rts_get_aggregated: call rts_aggregate first"); return RTS_ERR_INVALID; }
    const uint64_t R = c->n_recv;
    if (capacity < R) { rts_set_error("rts_get_aggregated: capacity too small"); return RTS_ERR_CAPACITY; }
    if (R == 0) return RTS_OK;
    RTS_HIP(hipStreamSynchronize(c->stream));
    if (rays) RTS_HIP(hipMemcpy(rays, c->d_rx_rays.p, sizeof(PerRayData)*R, hipMemcpyDeviceToHost));
    if (delay) RTS_HIP(hipMemcpy(delay, c->d_delay.p, sizeof(double)*R, hipMemcpyDeviceToHost));
    if (phase) RTS_HIP(hipMemcpy(phase, c->d_phase.p, sizeof(double)*R, hipMemcpyDeviceToHost));
    if (path_match) RTS_HIP(hipMemcpy(path_match, c->d_pathmatch.p, sizeof(int32_t)*R, hipMemcpyDeviceToHost));
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- complex return cube
extern "C" int rts_cube_attach(RtsHandle c, const RtsCubeParams* p, void* device_ptr)
{
    CHECK_HANDLE(c);
    if (!p || p->n_rx == 0 || p->n_pulses == 0 || p->n_bins == 0 || !(p->dt > 0) || !std::isfinite(p->t0)) { rts_set_error("rts_cube_attach: bad parameters"); return RTS_ERR_INVALID; }
    const size_t doubles = 2 * (size_t)p->n_rx * p->n_pulses * p->n_bins;
    RTS_HIP(hipStreamSynchronize(c->stream));
    c->cube_params = *p;
    if (device_ptr) c->cube = (double*)device_ptr;          // caller-owned (and caller-zeroed) device memory
    else { RTS_HIP(c->d_cube_own.reserve(doubles)); c->cube = c->d_cube_own.p; RTS_HIP(hipMemset(c->cube, 0, sizeof(double) * doubles)); }
    c->cube_set = true;
    return RTS_OK;
}

extern "C" int rts_cube_accumulate(RtsHandle c, uint32_t pulse_index, double cspeed, double carrier)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!c->cube_set) { rts_set_error("rts_cube_accumulate: call rts_cube_attach first"); return RTS_ERR_INVALID; }
    if (pulse_index >= c->cube_params.n_pulses) { rts_set_error("rts_cube_accumulate: pulse %u >= %u", pulse_index, c->cube_params.n_pulses); return RTS_ERR_INVALID; }
    return rts_cube_accumulate_device(c, pulse_index, cspeed, carrier);
}

extern "C" int rts_cube_accumulate_paths(RtsHandle c, uint32_t pulse_index)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!c->cube_set) { rts_set_error("rts_cube_accumulate_paths: call rts_cube_attach first"); return RTS_ERR_INVALID; }
    if (!c->agg_valid) { rts_set_error("rts_cube_accumulate_paths: call rts_aggregate for this pulse first (the groups' power, delay and phase are its results)"); return RTS_ERR_INVALID; }
    if (pulse_index >= c->cube_params.n_pulses) { rts_set_error("rts_cube_accumulate_paths: pulse %u >= %u", pulse_index, c->cube_params.n_pulses); return RTS_ERR_INVALID; }
    return rts_cube_accumulate_paths_device(c, pulse_index, c->agg_base_local);
}

extern "C" int rts_cube_doppler(RtsHandle c, uint32_t n_fft, void* device_out)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!c->cube_set) { rts_set_error("rts_cube_doppler: call rts_cube_attach first"); return RTS_ERR_INVALID; }
    if (n_fft < 2 || n_fft > 4096 || (n_fft & (n_fft - 1)) != 0 || n_fft < c->cube_params.n_pulses) {
        rts_set_error("rts_cube_doppler: n_fft = %u must be a power of two in [max(2, n_pulses = %u), 4096]", n_fft, c->cube_params.n_pulses); return RTS_ERR_INVALID; }
    const size_t doubles = 2 * (size_t)c->cube_params.n_rx * n_fft * c->cube_params.n_bins;
    if (device_out) c->doppler = (double*)device_out;
    else { RTS_HIP(c->d_doppler_own.reserve(doubles)); c->doppler = c->d_doppler_own.p; }
    c->doppler_n = n_fft;
    return rts_cube_doppler_device(c, n_fft, c->doppler);
}

extern "C" int rts_cube_doppler_get(RtsHandle c, double* host_out, uint64_t capacity_doubles)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!c->cube_set || !c->doppler || !host_out) { rts_set_error("rts_cube_doppler_get: no transform (rts_cube_doppler) / null output"); return RTS_ERR_INVALID; }
    const size_t doubles = 2 * (size_t)c->cube_params.n_rx * c->doppler_n * c->cube_params.n_bins;
    if (capacity_doubles < doubles) { rts_set_error("rts_cube_doppler_get: capacity too small"); return RTS_ERR_CAPACITY; }
    RTS_HIP(hipStreamSynchronize(c->stream));
    RTS_HIP(hipMemcpy(host_out, c->doppler, sizeof(double) * doubles, hipMemcpyDeviceToHost));
    return RTS_OK;
}

extern "C" int rts_cube_get(RtsHandle c, double* host_out, uint64_t capacity_doubles)
{
    CHECK_HANDLE(c);
    CHECK_CLOSED(c);
    if (!c->cube_set || !host_out) { rts_set_error("rts_cube_get: no cube / null output"); return RTS_ERR_INVALID; }
    const size_t doubles = 2 * (size_t)c->cube_params.n_rx * c->cube_params.n_pulses * c->cube_params.n_bins;
    if (capacity_doubles < doubles) { rts_set_error("rts_cube_get: capacity too small"); return RTS_ERR_CAPACITY; }
    RTS_HIP(hipStreamSynchronize(c->stream));
    RTS_HIP(hipMemcpy(host_out, c->cube, sizeof(double) * doubles, hipMemcpyDeviceToHost));
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- several GPUs: the plan of an interval
// ---------------------------------------------------------------------------------------------------------------------
// Ray sharding dealt by last-seen cost (include/rts_amd.h: rts_tile_records_get / _set, rts_deal_tiles, rts_set_tile_list)
extern "C" int rts_set_tile_list(RtsHandle c, uint32_t tile, const uint32_t* tile_ids, uint32_t n_ids)
{
    CHECK_HANDLE(c);
    if (c->pulse_open) { rts_set_error("rts_set_tile_list: a pulse of this handle is in flight"); return RTS_ERR_INVALID; }
    if (c->spec_pending) { int rc_ = rts_spec_resolve(c); if (rc_ != RTS_OK) return rc_; }
    // the cost records of the last launch are indexed through the list in force: into the history before it goes
    { int rc = rts_tile_costs_flush(c); if (rc != RTS_OK) return rc; }
    if (tile == 0) { c->il_list_n = 0; c->il_list_tile = 0; c->il_list_gen++; c->tile_last_valid = false; return RTS_OK; }      // no list any more
    if ((n_ids && !tile_ids) || tile % RTS_WTILE != 0) { rts_set_error("rts_set_tile_list: tile must be a positive multiple of %d launch indices (got %u)", RTS_WTILE, tile); return RTS_ERR_INVALID; }
    if (n_ids == 0) { c->il_list_n = 0; c->il_list_tile = tile; c->il_list_last = 0; c->il_list_gen++; c->tile_last_valid = false; return RTS_OK; }      // an EMPTY list: this worker was dealt nothing, its launches trace no launch index
    for (uint32_t k = 1; k < n_ids; k++) if (tile_ids[k] <= tile_ids[k - 1]) { rts_set_error("rts_set_tile_list: tile ids must be ascending and unique (entry %u: %u after %u)", k, tile_ids[k], tile_ids[k - 1]); return RTS_ERR_INVALID; }
    const uint64_t total = (uint64_t)c->params.width * c->params.width * c->params.width;
    if ((uint64_t)tile_ids[n_ids - 1] * tile >= total) { rts_set_error("rts_set_tile_list: tile %u of %u launch indices lies beyond W^3 = %llu", tile_ids[n_ids - 1], tile, (unsigned long long)total); return RTS_ERR_INVALID; }
    if ((uint64_t)n_ids * tile > 0xffffffffull) { rts_set_error("rts_set_tile_list: more than 2^32 launch indices"); return RTS_ERR_INVALID; }
    RTS_HIP(hipStreamSynchronize(c->stream));                      // (kernels still reading the old list through their launch constants: k_expand of the last pulse)
    RTS_HIP(c->d_il_list.reserve(n_ids));
    RTS_HIP(hipMemcpy(c->d_il_list.p, tile_ids, sizeof(uint32_t) * n_ids, hipMemcpyHostToDevice));
    c->il_list_n = n_ids; c->il_list_tile = tile; c->il_list_last = tile_ids[n_ids - 1]; c->il_list_gen++;
    c->tile_last_valid = false;
    return RTS_OK;
}

extern "C" int rts_tile_records_get(RtsHandle c, uint32_t* records, uint32_t n)
{
    CHECK_HANDLE(c);
    if (c->pulse_open) { rts_set_error("rts_tile_records_get: a pulse of this handle is in flight"); return RTS_ERR_INVALID; }
    const uint64_t total = (uint64_t)c->params.width * c->params.width * c->params.width;
    if (!records || n != (uint32_t)((total + RTS_WTILE - 1) / RTS_WTILE)) { rts_set_error("rts_tile_records_get: n must be ceil(W^3 / %d) = %llu", RTS_WTILE, (unsigned long long)((total + RTS_WTILE - 1) / RTS_WTILE)); return RTS_ERR_INVALID; }
    { int rc = rts_tile_costs_flush(c); if (rc != RTS_OK) return rc; }
    RTS_HIP(c->d_rec_tmp.reserve(n));
    { int rc = rts_tile_records_masked(c, c->d_rec_tmp.p, n); if (rc != RTS_OK) return rc; }
    RTS_HIP(hipMemcpyAsync(records, c->d_rec_tmp.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    RTS_HIP(hipStreamSynchronize(c->stream));
    return RTS_OK;
}

extern "C" int rts_tile_records_set(RtsHandle c, const uint32_t* records, uint32_t n)
{
    CHECK_HANDLE(c);
    if (c->pulse_open) { rts_set_error("rts_tile_records_set: a pulse of this handle is in flight"); return RTS_ERR_INVALID; }
    const uint64_t total = (uint64_t)c->params.width * c->params.width * c->params.width;
    if (!records || n != (uint32_t)((total + RTS_WTILE - 1) / RTS_WTILE)) { rts_set_error("rts_tile_records_set: n must be ceil(W^3 / %d) = %llu", RTS_WTILE, (unsigned long long)((total + RTS_WTILE - 1) / RTS_WTILE)); return RTS_ERR_INVALID; }
    RTS_HIP(hipStreamSynchronize(c->stream));
    RTS_HIP(c->hist->d.reserve(n));
    RTS_HIP(hipMemcpy(c->hist->d.p, records, sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    c->hist->n = n; c->hist->any = true; c->tile_cost_pending = false; c->order_sum_valid = false;      // (records of the last launch not merged yet are superseded)
    return RTS_OK;
}

// Longest-first dealing of plan tiles to `parts` workers.  Deterministic: ties go to the lower tile number, then to the lower worker.
extern "C" int rts_deal_tiles(const uint32_t* records, uint32_t n_records, uint64_t total_rays, uint32_t tile, uint32_t parts, uint32_t* part_of_tile, uint64_t* cost_of_part)
{
    if (!records || !part_of_tile || parts == 0 || tile == 0 || tile % RTS_WTILE != 0 || total_rays == 0) { rts_set_error("rts_deal_tiles: bad arguments (tile must be a positive multiple of %d, parts >= 1)", RTS_WTILE); return RTS_ERR_INVALID; }
    if ((uint64_t)n_records != (total_rays + RTS_WTILE - 1) / RTS_WTILE) { rts_set_error("rts_deal_tiles: n_records must be ceil(total_rays / %d)", RTS_WTILE); return RTS_ERR_INVALID; }
    const uint64_t n_plan64 = (total_rays + tile - 1) / tile;
    if (n_plan64 > 0xffffffffull) { rts_set_error("rts_deal_tiles: too many tiles"); return RTS_ERR_INVALID; }
    const uint32_t n_plan = (uint32_t)n_plan64, per = tile / RTS_WTILE;
    std::vector<uint64_t> cost(n_plan); std::vector<uint32_t> n_long(n_plan, 0);
    for (uint32_t t = 0; t < n_plan; t++) {
        uint64_t v = 0; const uint64_t w0 = (uint64_t)t * per, w1 = std::min<uint64_t>(w0 + per, n_records);
        for (uint64_t w = w0; w < w1; w++) { v += records[w] & 0x3fffffffu; n_long[t] += (records[w] >> 30) ? 1u : 0u; }      // (LONG or LONGISH walks: what a part's head rule may hand to the cooperative kernel)
        cost[t] = v;                                               // 0: a tile nobody traced yet
    }
    std::vector<uint32_t> order, heads; order.reserve(n_plan);
    for (uint32_t t = 0; t < n_plan; t++) if (cost[t]) (n_long[t] ? heads : order).push_back(t);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
    std::stable_sort(heads.begin(), heads.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
    std::vector<std::pair<uint64_t, uint32_t>> heap(parts);
    for (uint32_t r = 0; r < parts; r++) heap[r] = {0, r};
    std::vector<uint64_t> n_of(parts, 0);
    // the tiles that hold LONG or LONGISH WALKS (bits 31 / 30 of a wave tile's record: the cooperative kernel's candidates) first, and by COUNT: a part's time follows the number of cooperative
    // tiles it holds more closely than their recorded cost (BASELINE configs[3] dealt by cost alone: 72 .. 239 of them per eighth, 0.77 .. 1.02 ms;
    // profiles/r05_c4_as_rank.log) -- longest first, each to the worker that holds the fewest of them so far (ties: the least cost)
    {
        std::vector<uint64_t> nl(parts, 0);
        for (uint32_t t : heads) {
            uint32_t best = 0;
            for (uint32_t r = 1; r < parts; r++) if (nl[r] < nl[best] || (nl[r] == nl[best] && heap[r].first < heap[best].first)) best = r;
            part_of_tile[t] = best; nl[best] += n_long[t]; heap[best].first += cost[t]; n_of[best]++;
        }
    }
    // the other tiles WITH a record: longest first, each to the worker with the least cost so far -- min-heap of (load, worker)
    auto cmp = [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (uint32_t k = 0; k < (uint32_t)order.size(); k++) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        std::pair<uint64_t, uint32_t>& top = heap.back();
        part_of_tile[order[k]] = top.second; top.first += cost[order[k]]; n_of[top.second]++;
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
    std::vector<uint64_t> load(parts, 0);
    for (const auto& h : heap) load[h.second] = h.first;
    // the tiles WITHOUT a record (the first interval, a partial table, records a launch dropped): nothing is known about their cost, so their
    // COUNT is balanced -- in ascending tile order, each to the worker holding the fewest tiles so far (as cost-1 entries of the
    // heap above they all went to the lightest workers: 64 tiles, records {500, 100, 50}, three workers: 0 / ~6 / ~55 of the 61)
    for (uint32_t t = 0; t < n_plan; t++) {
        if (cost[t]) continue;
        uint32_t best = 0;
        for (uint32_t r = 1; r < parts; r++) if (n_of[r] < n_of[best]) best = r;
        part_of_tile[t] = best; n_of[best]++; load[best] += 1;
    }
    if (cost_of_part) for (uint32_t r = 0; r < parts; r++) cost_of_part[r] = load[r];
    return RTS_OK;
}

static uint64_t plan_part_count(uint64_t total, uint32_t tile, uint32_t parts, uint32_t part)
{
    if (parts <= 1) return total;
    const uint64_t stride = (uint64_t)tile * parts, full = total / stride, rem = total % stride, lo = (uint64_t)part * tile;
    return full * tile + (rem > lo ? std::min<uint64_t>(rem - lo, tile) : 0);
}

extern "C" int rts_plan_cpi(uint64_t total_rays, uint32_t n_pulses, uint32_t rank, uint32_t world, uint32_t mode, uint32_t min_items,
                            uint32_t tile, RtsPlanItem* out, uint32_t capacity, uint32_t* n_out)
{
    if (tile == 0) tile = RTS_PLAN_TILE;
    if (!n_out || world == 0 || rank >= world || mode > RTS_SHARD_PULSES_WHOLE) { rts_set_error("rts_plan_cpi: bad argument (rank %u of %u, mode %u)", rank, world, mode); return RTS_ERR_INVALID; }
    std::vector<RtsPlanItem> plan;
    auto item = [&](uint32_t pulse, uint32_t parts, uint32_t part) { RtsPlanItem it; memset(&it, 0, sizeof(it)); it.pulse = pulse; it.ray_first = 0; it.ray_count = total_rays;
                                                                     if (parts > 1) { it.interleave_tile = tile; it.interleave_parts = parts; it.interleave_part = part; } return it; };
    if (mode == RTS_SHARD_RAYS) {
        for (uint32_t k = 0; k < n_pulses; k++) plan.push_back(item(k, world, rank));
    } else if (mode == RTS_SHARD_PULSES_WHOLE && n_pulses >= world) {
        // whole pulses only, in contiguous runs (a handle's consecutive pulses are consecutive in time: its cost history fits): the first
        // n_pulses % world workers trace one pulse more.  A part of a pulse is a launch of another SHAPE -- its tile order starts from nothing,
        // and on a short interval that costs more than the imbalance of one pulse (profiles/r05d_as_rank_pulses_*.log)
        const uint32_t base = n_pulses / world, left = n_pulses - base * world;
        const uint32_t first = rank * base + std::min(rank, left), count = base + (rank < left ? 1u : 0u);
        for (uint32_t i = 0; i < count; i++) plan.push_back(item(first + i, 1, 0));
    } else {
        const uint32_t base = n_pulses / world, left = n_pulses - base * world;
        for (uint32_t i = 0; i < base; i++) plan.push_back(item(rank * base + i, 1, 0));
        if (left) {                                      // the left-over pulse this worker helps with, and who else does
            const uint32_t mine = (uint32_t)((uint64_t)rank * left / world);
            uint32_t group = 0, index = 0;
            for (uint32_t r = 0; r < world; r++) if ((uint32_t)((uint64_t)r * left / world) == mine) { if (r == rank) index = group; group++; }
            plan.push_back(item(base * world + mine, group, index));
        }
    }
    // refinement: the oldest item is split in two (every other one of its tiles each) until there are min_items
    while (!plan.empty() && plan.size() < min_items) {
        const RtsPlanItem it = plan.front();
        const uint32_t parts = it.interleave_parts > 1 ? it.interleave_parts : 1u, part = it.interleave_parts > 1 ? it.interleave_part : 0u;
        if (parts > 0x3fffffffu || plan_part_count(it.ray_count, tile, 2 * parts, part + parts) == 0) break;      // nothing left to split off
        plan.erase(plan.begin());
        plan.push_back(item(it.pulse, 2 * parts, part)); plan.push_back(item(it.pulse, 2 * parts, part + parts));
    }
    *n_out = (uint32_t)plan.size();
    if (out) { if (capacity < plan.size()) { rts_set_error("rts_plan_cpi: capacity %u < %zu items", capacity, plan.size()); return RTS_ERR_CAPACITY; }
               if (!plan.empty()) memcpy(out, plan.data(), sizeof(RtsPlanItem) * plan.size()); }
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- several GPUs: sum of the return cubes
// RCCL is loaded on first use (dlopen): librts_amd.so has no link-time dependency on it, and a process that already holds
// a copy (torch.distributed) shares that one.
namespace {
typedef struct ncclComm* rccl_comm_t;
struct RcclApi {
    void* lib = nullptr; bool tried = false;
    int (*CommInitAll)(rccl_comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (tried) return lib != nullptr;
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
        CommInitAll = (int (*)(rccl_comm_t*, int, const int*))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (int (*)(rccl_comm_t))dlsym(lib, "ncclCommDestroy");
        AllReduce = (int (*)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t))dlsym(lib, "ncclAllReduce");
        GroupStart = (int (*)())dlsym(lib, "ncclGroupStart"); GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
        GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd) { dlclose(lib); lib = nullptr; }
        return lib != nullptr;
    }
};
RcclApi g_rccl;
// Communicators are created once per set of devices and kept (ncclCommInitAll costs hundreds of milliseconds on eight GPUs, an
// all-reduce of a cube half a millisecond); an entry lives as long as one of the handles that used it (rts_destroy ->
// rts_comm_cache_forget).
struct CommEntry { std::vector<int> devs; std::vector<rccl_comm_t> comms; std::vector<RtsContext*> users; };
std::mutex g_comm_mu;
std::vector<CommEntry> g_comm_cache;
__global__ void k_add_f64(double* __restrict__ dst, const double* __restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}
// the communicators for the devices of `hs` (in handle order), created on first use; null on failure (*rc_out = RCCL's code)
const std::vector<rccl_comm_t>* comm_cache_get(RtsHandle* hs, uint32_t n, int* rc_out)
{
    std::vector<int> devs(n);
    for (uint32_t i = 0; i < n; i++) devs[i] = hs[i]->device;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    CommEntry* e = nullptr;
    for (auto& x : g_comm_cache) if (x.devs == devs) { e = &x; break; }
    if (!e) {
        CommEntry ne; ne.devs = devs; ne.comms.assign(n, nullptr);
        const int rc = g_rccl.CommInitAll(ne.comms.data(), (int)n, devs.data());
        if (rc != 0) { *rc_out = rc; return nullptr; }
        g_comm_cache.push_back(std::move(ne)); e = &g_comm_cache.back();
    }
    for (uint32_t i = 0; i < n; i++) if (std::find(e->users.begin(), e->users.end(), hs[i]) == e->users.end()) e->users.push_back(hs[i]);
    *rc_out = 0;
    return &e->comms;
}
}  // namespace

// rts_destroy: the handle no longer keeps any communicator set alive; a set without users is destroyed
void rts_comm_cache_forget(RtsContext* c)
{
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (size_t k = 0; k < g_comm_cache.size();) {
        CommEntry& e = g_comm_cache[k];
        e.users.erase(std::remove(e.users.begin(), e.users.end(), c), e.users.end());
        if (e.users.empty()) { for (rccl_comm_t q : e.comms) if (q) (void)g_rccl.CommDestroy(q); g_comm_cache.erase(g_comm_cache.begin() + k); }
        else k++;
    }
}

// transport 0: RCCL when the handles sit on distinct devices and librccl loads, else peer copies; 1: RCCL or an error (with
// ONE handle: a one-rank communicator and all-reduce -- the identity, but the dlopen, the symbol table and the call sequence
// run, which is how the RCCL path is exercised on a single GPU); 2: peer copies.
extern "C" int rts_cube_reduce(RtsHandle* hs, uint32_t n, int transport)
{
    if (!hs || n == 0) { rts_set_error("rts_cube_reduce: no handles"); return RTS_ERR_INVALID; }
    for (uint32_t i = 0; i < n; i++) {
        if (!hs[i] || !hs[i]->cube_set) { rts_set_error("rts_cube_reduce: handle %u has no cube (rts_cube_attach)", i); return RTS_ERR_INVALID; }
        const RtsCubeParams &a = hs[0]->cube_params, &b = hs[i]->cube_params;
        if (a.n_rx != b.n_rx || a.n_pulses != b.n_pulses || a.n_bins != b.n_bins) { rts_set_error("rts_cube_reduce: handle %u has a cube of another shape", i); return RTS_ERR_INVALID; }
        for (uint32_t j = 0; j < i; j++) if (hs[j]->cube == hs[i]->cube) { rts_set_error("rts_cube_reduce: handles %u and %u share one cube buffer (nothing to add)", j, i); return RTS_ERR_INVALID; }
    }
    const size_t doubles = 2 * (size_t)hs[0]->cube_params.n_rx * hs[0]->cube_params.n_pulses * hs[0]->cube_params.n_bins;
    for (uint32_t i = 0; i < n; i++) { RtsContext* c = hs[i]; CHECK_CLOSED(c); RTS_HIP(hipSetDevice(c->device)); RTS_HIP(hipStreamSynchronize(c->stream)); }
    if (n == 1 && transport != 1) return RTS_OK;
    bool distinct = true;
    for (uint32_t i = 0; i < n; i++) for (uint32_t j = 0; j < i; j++) if (hs[i]->device == hs[j]->device) distinct = false;
    if (transport != 2 && distinct && g_rccl.load()) {
        // one communicator per device, all in this process; one all-reduce (sum, f64) on each handle's stream inside a group
        int rc = 0;
        const std::vector<rccl_comm_t>* comms = comm_cache_get(hs, n, &rc);
        if (comms) {
            // From here on there is no falling back: the all-reduce is in place, and after a failure part-way some cubes may
            // already hold partial sums -- adding them again over peer copies would count them twice.
            rc = g_rccl.GroupStart();
            for (uint32_t i = 0; i < n && rc == 0; i++) { (void)hipSetDevice(hs[i]->device); rc = g_rccl.AllReduce(hs[i]->cube, hs[i]->cube, doubles, 8 /* ncclFloat64 */, 0 /* ncclSum */, (*comms)[i], hs[i]->stream); }
            const int rc2 = g_rccl.GroupEnd(); if (rc == 0) rc = rc2;
            for (uint32_t i = 0; i < n; i++) { (void)hipSetDevice(hs[i]->device); (void)hipStreamSynchronize(hs[i]->stream); }
            if (rc == 0) return RTS_OK;
            rts_set_error("rts_cube_reduce: RCCL all-reduce failed (%s); the cubes may hold partial sums", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
            return RTS_ERR_HIP;
        }
        if (transport == 1) { rts_set_error("rts_cube_reduce: ncclCommInitAll failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"); return RTS_ERR_HIP; }
        // no communicator, nothing issued: peer copies
    } else if (transport == 1) { rts_set_error("rts_cube_reduce: RCCL requested but %s", distinct ? "librccl could not be loaded" : "two handles share a device"); return RTS_ERR_UNSUPPORTED; }
    if (n == 1) return RTS_OK;
    // peer copies: everything is added into handle 0's cube in handle order (bit-reproducible), then copied back out
    RtsContext* c0 = hs[0];
    RTS_HIP(hipSetDevice(c0->device));
    DevBuf<double> tmp; RTS_HIP(tmp.reserve(doubles));
    struct FreeTmp { DevBuf<double>& t; ~FreeTmp() { t.release(); } } ft{tmp};
    for (uint32_t i = 1; i < n; i++) {
        RTS_HIP(hipMemcpyPeerAsync(tmp.p, c0->device, hs[i]->cube, hs[i]->device, sizeof(double) * doubles, c0->stream));
        k_add_f64<<<(unsigned)((doubles + 255) / 256), 256, 0, c0->stream>>>(c0->cube, tmp.p, doubles);
        RTS_HIP(hipGetLastError());
    }
    RTS_HIP(hipStreamSynchronize(c0->stream));
    for (uint32_t i = 1; i < n; i++) RTS_HIP(hipMemcpyPeer(hs[i]->cube, hs[i]->device, c0->cube, c0->device, sizeof(double) * doubles));
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- host-side group algebra
// Merge partial group tables (one per GPU): same (rx, path) => sums add, min_ray takes the minimum.
// Groups are combined in input order, output sorted by (rx, path) so the result does not depend
// on how many tables were concatenated beyond f64 rounding of the sums.
extern "C" int rts_merge_groups(const RtsGroup* in, uint32_t n_in, uint32_t depth, RtsGroup* out, uint32_t* n_out)
{
    if ((!in && n_in) || !n_out) { rts_set_error("rts_merge_groups: null argument"); return RTS_ERR_INVALID; }
    if (depth > RTS_MAX_DEPTH) { rts_set_error("rts_merge_groups: depth > %d", RTS_MAX_DEPTH); return RTS_ERR_INVALID; }
    typedef std::vector<int32_t> Key;
    std::map<Key, RtsGroup> m;
    for (uint32_t i = 0; i < n_in; i++) {
        Key k(1 + RTS_MAX_DEPTH); k[0] = in[i].rx; for (int d = 0; d < RTS_MAX_DEPTH; d++) k[1 + d] = in[i].path[d];
        auto it = m.find(k);
        if (it == m.end()) m.emplace(k, in[i]);
        else {
            RtsGroup& g = it->second;
            g.n += in[i].n; g.sum_sqrt_power += in[i].sum_sqrt_power; g.sum_delay += in[i].sum_delay; g.sum_phase += in[i].sum_phase; g.sum_doppler += in[i].sum_doppler;
            g.min_ray = std::min(g.min_ray, in[i].min_ray); g.direct = g.direct | in[i].direct;
        }
    }
    if (out) { if (*n_out < m.size()) { rts_set_error("rts_merge_groups: capacity too small"); *n_out = (uint32_t)m.size(); return RTS_ERR_CAPACITY; }
               uint32_t j = 0; for (auto& kv : m) out[j++] = kv.second; }
    *n_out = (uint32_t)m.size();
    return RTS_OK;
}

// Responses the reference would emit (ray_tracer.cpp:1290-1321) from a complete group table:
//   every non-direct group -> one response at its smallest ray, with the group means
//   (aggregation.cu:88-93); a receiver's direct group -> pathMatch = smallest ray of ANY group
//   at that receiver and sums over ALL its rays (aggregation.cu:56): emitted only if that
//   smallest ray is itself direct, otherwise it collapses onto the reflected group (quirk 9).
extern "C" int rts_groups_to_responses(const RtsGroup* groups, uint32_t n_groups, RtsResponse* out, uint32_t capacity, uint32_t* n_out)
{
    if ((!groups && n_groups) || !n_out) { rts_set_error("rts_groups_to_responses: null argument"); return RTS_ERR_INVALID; }
    struct Tot { double n = 0, sp = 0, dl = 0, ph = 0, dp = 0; uint64_t mn = ~0ULL; bool mn_direct = false; bool has_direct = false; };
    std::map<int32_t, Tot> tot;
    for (uint32_t i = 0; i < n_groups; i++) {
        Tot& t = tot[groups[i].rx];
        t.n += groups[i].n; t.sp += groups[i].sum_sqrt_power; t.dl += groups[i].sum_delay; t.ph += groups[i].sum_phase; t.dp += groups[i].sum_doppler;
        if (groups[i].min_ray < t.mn) { t.mn = groups[i].min_ray; t.mn_direct = groups[i].direct != 0; }
        if (groups[i].direct) t.has_direct = true;
    }
    std::vector<RtsResponse> r;
    for (uint32_t i = 0; i < n_groups; i++) {
        const RtsGroup& g = groups[i];
        RtsResponse q; q.rx = g.rx;
        if (!g.direct) {
            if (!(g.n > 0)) continue;
            double v = g.sum_sqrt_power / g.n;
            q.ray = g.min_ray; q.n = (uint32_t)g.n; q.power = v*v; q.delay = g.sum_delay / g.n; q.phase = g.sum_phase / g.n; q.doppler = g.sum_doppler / g.n;
            r.push_back(q);
        } else {
            const Tot& t = tot[g.rx];
            if (!t.mn_direct || !(t.n > 0)) continue;       // direct response lost (quirk 9) when a reflected ray has the smallest index
            double v = t.sp / t.n;
            q.ray = t.mn; q.n = (uint32_t)t.n; q.power = v*v; q.delay = t.dl / t.n; q.phase = t.ph / t.n; q.doppler = t.dp / t.n;
            r.push_back(q);
        }
    }
    std::sort(r.begin(), r.end(), [](const RtsResponse& a, const RtsResponse& b) { return a.ray < b.ray; });   // sort + unique of pathMatch (:1290-1292)
    *n_out = (uint32_t)r.size();
    if (out) { if (capacity < r.size()) { rts_set_error("rts_groups_to_responses: capacity too small"); return RTS_ERR_CAPACITY; }
               if (!r.empty()) memcpy(out, r.data(), sizeof(RtsResponse)*r.size()); }
    return RTS_OK;
}

// ------------------------------------------------------------------------------------- rs::kernel_wrapper
// Contexts of the handle-less entry points, one per device (the device current on the calling thread at the call), created on
// first use and kept for the life of the process.
static std::map<int, RtsContext*> g_wrapper_ctx;
static std::mutex g_wrapper_mutex;

static int wrapper_context(RtsContext** out)
{
    int cur = 0; if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    std::lock_guard<std::mutex> lock(g_wrapper_mutex);
    auto it = g_wrapper_ctx.find(cur);
    if (it == g_wrapper_ctx.end()) {
        RtsParams p; memset(&p, 0, sizeof(p)); p.width = 1; p.max_refl = 1; p.device = cur;
        RtsContext* c = nullptr;
        int rc = rts_create(&p, &c); if (rc != RTS_OK) return rc;
        it = g_wrapper_ctx.emplace(cur, c).first;
    }
    *out = it->second;
    return RTS_OK;
}

extern "C" int rts_kernel_wrapper_on(RtsHandle h, PerRayData* h_rx_results_arr, int* h_rx_intersects_arr, unsigned int receivedRays,
                                     unsigned int depthTotal, unsigned int MaxThreads, unsigned int MaxBlocks, double cspeed,
                                     double carrier, double* h_npath_arr, double* h_power_arr, double* h_doppler_arr,
                                     double* h_delay_arr, double* h_phase_arr, int* h_pathMatch)
{
    (void)MaxThreads; (void)MaxBlocks;   // launch shapes are chosen by the library (aggregation.cu:142-160 picks them from these)
    if (receivedRays == 0) return RTS_OK;
    if (!h_rx_results_arr || (depthTotal && !h_rx_intersects_arr) || !h_delay_arr || !h_phase_arr || !h_pathMatch) { rts_set_error("rts_kernel_wrapper: null array"); return RTS_ERR_INVALID; }
    RtsContext* c = h;
    if (!c) { int rc = wrapper_context(&c); if (rc != RTS_OK) return rc; }
    else { CHECK_CLOSED(c); c->agg_valid = false; c->agg_pending.valid = false; c->n_recv = 0; c->mirror.recv_valid = false; c->mirror.agg_valid = false; c->mirror.want = false; }      // the handle's own received set is overwritten
    RTS_HIP(hipSetDevice(c->device));
    const size_t R = receivedRays, D = depthTotal;
    RTS_HIP(c->d_rx_rays.reserve(R)); RTS_HIP(c->d_rx_paths.reserve(R*D + 1)); RTS_HIP(c->d_delay.reserve(R)); RTS_HIP(c->d_phase.reserve(R)); RTS_HIP(c->d_pathmatch.reserve(R));
    RTS_HIP(c->d_rx_angles.reserve(3*R + 1));     // scratch: npath / power / doppler initial sums
    hipStream_t st = c->stream;
    RTS_HIP(hipMemcpyAsync(c->d_rx_rays.p, h_rx_results_arr, sizeof(PerRayData)*R, hipMemcpyHostToDevice, st));
    if (D) RTS_HIP(hipMemcpyAsync(c->d_rx_paths.p, h_rx_intersects_arr, sizeof(int)*R*D, hipMemcpyHostToDevice, st));
    RTS_HIP(hipMemcpyAsync(c->d_delay.p, h_delay_arr, sizeof(double)*R, hipMemcpyHostToDevice, st));
    RTS_HIP(hipMemcpyAsync(c->d_phase.p, h_phase_arr, sizeof(double)*R, hipMemcpyHostToDevice, st));
    RTS_HIP(hipMemcpyAsync(c->d_pathmatch.p, h_pathMatch, sizeof(int)*R, hipMemcpyHostToDevice, st));
    double* d_np = nullptr; double* d_pw = nullptr; double* d_dp = nullptr;
    if (h_npath_arr) { d_np = c->d_rx_angles.p; RTS_HIP(hipMemcpyAsync(d_np, h_npath_arr, sizeof(double)*R, hipMemcpyHostToDevice, st)); }
    if (h_power_arr) { d_pw = c->d_rx_angles.p + R; RTS_HIP(hipMemcpyAsync(d_pw, h_power_arr, sizeof(double)*R, hipMemcpyHostToDevice, st)); }
    if (h_doppler_arr) { d_dp = c->d_rx_angles.p + 2*R; RTS_HIP(hipMemcpyAsync(d_dp, h_doppler_arr, sizeof(double)*R, hipMemcpyHostToDevice, st)); }
    int32_t max_path = -1, max_rx = 0;
    for (size_t i = 0; i < R * D; i++) { if (h_rx_intersects_arr[i] < -1) { rts_set_error("rts_kernel_wrapper: path entry < -1"); return RTS_ERR_INVALID; } max_path = std::max(max_path, (int32_t)h_rx_intersects_arr[i]); }
    for (size_t i = 0; i < R; i++) { if (h_rx_results_arr[i].received < 0) { rts_set_error("rts_kernel_wrapper: ray %zu is not a received ray", i); return RTS_ERR_INVALID; } max_rx = std::max(max_rx, h_rx_results_arr[i].received); }
    int rc = rts_aggregate_device(c, max_path, max_rx, c->d_rx_paths.p, R, (uint32_t)D, cspeed, carrier, 0, c->d_rx_rays.p, c->d_delay.p, c->d_phase.p,
                                  c->d_pathmatch.p, nullptr, d_np, d_pw, d_dp, INT32_MIN /* use the caller's h_pathMatch */, nullptr);
    if (rc != RTS_OK) return rc;
    // copy back what the reference copies back (aggregation.cu:169-172)
    RTS_HIP(hipMemcpyAsync(h_rx_results_arr, c->d_rx_rays.p, sizeof(PerRayData)*R, hipMemcpyDeviceToHost, st));
    RTS_HIP(hipMemcpyAsync(h_delay_arr, c->d_delay.p, sizeof(double)*R, hipMemcpyDeviceToHost, st));
    RTS_HIP(hipMemcpyAsync(h_phase_arr, c->d_phase.p, sizeof(double)*R, hipMemcpyDeviceToHost, st));
    RTS_HIP(hipMemcpyAsync(h_pathMatch, c->d_pathmatch.p, sizeof(int)*R, hipMemcpyDeviceToHost, st));
    RTS_HIP(hipStreamSynchronize(st));
    return RTS_OK;
}

extern "C" int rts_kernel_wrapper(PerRayData* h_rx_results_arr, int* h_rx_intersects_arr, unsigned int receivedRays,
                                  unsigned int depthTotal, unsigned int MaxThreads, unsigned int MaxBlocks, double cspeed,
                                  double carrier, double* h_npath_arr, double* h_power_arr, double* h_doppler_arr,
                                  double* h_delay_arr, double* h_phase_arr, int* h_pathMatch)
{
    return rts_kernel_wrapper_on(nullptr, h_rx_results_arr, h_rx_intersects_arr, receivedRays, depthTotal, MaxThreads, MaxBlocks, cspeed, carrier,
                                 h_npath_arr, h_power_arr, h_doppler_arr, h_delay_arr, h_phase_arr, h_pathMatch);
}

// the C++ symbol the reference's caller links against (aggregation.cuh:19-22).  The reference prints and exit(1)s on a CUDA
// error (aggregation.cu:17-27); a library must not end the process, and carrying on with un-aggregated arrays would be worse:
// a failure is a C++ exception.
namespace rs {
void kernel_wrapper(PerRayData* h_rx_results_arr, int* h_rx_intersects_arr, unsigned int receivedRays, unsigned int depthTotal,
                    unsigned int MaxThreads, unsigned int MaxBlocks, double cspeed, double carrier, double* h_npath_arr,
                    double* h_power_arr, double* h_doppler_arr, double* h_delay_arr, double* h_phase_arr, int* h_pathMatch)
{
    int rc = rts_kernel_wrapper(h_rx_results_arr, h_rx_intersects_arr, receivedRays, depthTotal, MaxThreads, MaxBlocks, cspeed, carrier,
                                h_npath_arr, h_power_arr, h_doppler_arr, h_delay_arr, h_phase_arr, h_pathMatch);
    if (rc != RTS_OK) throw std::runtime_error(std::string("rs::kernel_wrapper: ") + rts_last_error());
}
}

// ------------------------------------------------------------------------------------- introspection
extern "C" int rts_get_bvh(RtsHandle c, void* nodes128, uint32_t* leaf_prim, int32_t* roots, uint32_t node_capacity, uint32_t leaf_capacity, uint32_t* n_leaves)
{
    CHECK_HANDLE(c);
    if (n_leaves) *n_leaves = c->scene->n_leaves;
    if ((nodes128 && node_capacity < c->scene->n_nodes) || (leaf_prim && leaf_capacity < c->scene->n_leaves)) { rts_set_error("rts_get_bvh: capacity too small (%u nodes, %u leaves)", c->scene->n_nodes, c->scene->n_leaves); return RTS_ERR_CAPACITY; }
    RTS_HIP(hipStreamSynchronize(c->stream));
    if (nodes128 && c->scene->n_nodes) {
        RTS_HIP(hipMemcpy(nodes128, c->scene->d_nodes4.p, sizeof(RtsNode4)*c->scene->n_nodes, hipMemcpyDeviceToHost));
        RtsNode4* nd = static_cast<RtsNode4*>(nodes128);              // (the builders' form: leaf children as ~slot, see k_children_to_prims)
        for (uint32_t i = 0; i < c->scene->n_nodes; i++) for (int k = 0; k < 4; k++) { nd[i].child[k] = nd[i].pad[k]; nd[i].pad[k] = 0; }
    }
    if (leaf_prim && c->scene->n_leaves) RTS_HIP(hipMemcpy(leaf_prim, c->scene->d_leaf_prim.p, sizeof(uint32_t)*c->scene->n_leaves, hipMemcpyDeviceToHost));
    if (roots) for (size_t t = 0; t < c->scene->blas.size(); t++) roots[t] = c->scene->blas[t].root;
    return RTS_OK;
}

__global__ void k_self_test_math(const float* y, const float* x, float* at, const double* a, const double* b, double* dv, double* sq, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    at[i] = rts_atan2f(y[i], x[i]);
    dv[i] = a[i] / b[i];
    sq[i] = sqrt(fabs(a[i]));
}

extern "C" int rts_self_test_math(RtsHandle c, const float* y, const float* x, float* atan2f_out, const double* a, const double* b,
                                  double* div_out, double* sqrt_out, uint32_t n)
{
    CHECK_HANDLE(c);
    if (n == 0) return RTS_OK;
    float *dy, *dx, *dat; double *da, *db, *ddv, *dsq;
    RTS_HIP(hipMalloc((void**)&dy, 4*n)); RTS_HIP(hipMalloc((void**)&dx, 4*n)); RTS_HIP(hipMalloc((void**)&dat, 4*n));
    RTS_HIP(hipMalloc((void**)&da, 8*n)); RTS_HIP(hipMalloc((void**)&db, 8*n)); RTS_HIP(hipMalloc((void**)&ddv, 8*n)); RTS_HIP(hipMalloc((void**)&dsq, 8*n));
    RTS_HIP(hipMemcpy(dy, y, 4*n, hipMemcpyHostToDevice)); RTS_HIP(hipMemcpy(dx, x, 4*n, hipMemcpyHostToDevice));
    RTS_HIP(hipMemcpy(da, a, 8*n, hipMemcpyHostToDevice)); RTS_HIP(hipMemcpy(db, b, 8*n, hipMemcpyHostToDevice));
    k_self_test_math<<<(n + 255)/256, 256, 0, c->stream>>>(dy, dx, dat, da, db, ddv, dsq, n);
    RTS_HIP(hipGetLastError()); RTS_HIP(hipStreamSynchronize(c->stream));
    RTS_HIP(hipMemcpy(atan2f_out, dat, 4*n, hipMemcpyDeviceToHost)); RTS_HIP(hipMemcpy(div_out, ddv, 8*n, hipMemcpyDeviceToHost)); RTS_HIP(hipMemcpy(sqrt_out, dsq, 8*n, hipMemcpyDeviceToHost));
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dat); (void)hipFree(da); (void)hipFree(db); (void)hipFree(ddv); (void)hipFree(dsq);
    return RTS_OK;
}
