#!/usr/bin/env python3
"""bench.py -- RTS hot path on MI355X: Mrays/s and ms/pulse on the BASELINE.json metric config.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] -- aircraft-like 100 000-triangle mesh, 1 Tx / 4 Rx, W = 216
(10 077 696 launch indices per pulse), maxRefl = 6; the configuration the metric ("Mrays/s ... 100k-tri scene") is quoted
on.  Synthetic mesh, isotropic antennas, RCS 1.  --config c3ecef is the same scene at Earth-centred coordinates
(10 km above the reference's Earth sphere: |position| = 6.388e6 m, the Earth test of ray_tracer.cu:438-476 live),
--config c2 / c2file the 20 480-triangle icosphere / the 10 000-triangle file-mesh sphere of configs[1].

A step is ONE PULSE end to end: target placement for that pulse (the target moves every pulse: world-space vertices,
normals and leaf records are re-placed on the device inside the timed region; the hierarchy itself is static in target
space because the reference's targets are rigid -- the reference has OptiX rebuild its acceleration structure every
pulse), trace of the pulse's W^3 launch indices, ordering + expansion of the received rays, finalisation and group-by
aggregation into the pulse's responses.  A "ray" in Mrays/s is one traced segment (one rtTrace of the reference: primary
or bounce).  Pulses are independent, so each GPU keeps --inflight of them in flight on as many handles.  All of that is
inside the timed region; ms_per_step is wall time / pulses.

N > 1 ranks (one process per GPU): STRONG scaling is the line's `value` -- ONE interval of --steps pulses shared by the N ranks
(BASELINE.json's north star: "rays shard ... RCCL reduce ... strong-scaling efficiency"), `ms_per_step` = the interval's wall time /
--steps.  How it is shared (--shard auto): WHOLE pulses in contiguous runs, the first K mod N ranks one pulse more (rts_plan_cpi, RTS_SHARD_PULSES_WHOLE),
whenever the interval has at least as many pulses as there are ranks; with FEWER pulses than ranks -- the latency of single pulses -- every pulse
is split over all ranks by RAYS (tiles dealt longest-first from the cost records of the warm-up interval, rts_deal_tiles) when a rank's part is at
least 4 M launch indices, else over groups of ranks.  This is measured, not VERDICT r4's rule ("rays below 4 x --inflight pulses per rank"): one GPU
running each rank's plan of a 20-pulse interval in turn (profiles/r05d_as_rank_pulses_*.log) took 3.3-3.7 ms per rank with left-over pulses split and
12-14 ms ray-sharded on configs[2] (11.9 ms on one GPU), 34-45 / 30-68 ms on configs[3] (110 ms): a part of a pulse is a launch of another shape
whose tile schedule starts from nothing, and an eighth of a configs[2] pulse is 0.08 ms of tracing against a 0.07 ms empty launch.
Either way the per-(receiver, path) group tables are exchanged once per interval (one all-gather over RCCL) and the complex return
cube is summed once (one all-reduce), both inside the timed region.  The same job then measures, outside that region and reported
as secondary blocks: `weak` -- every rank --steps whole pulses (an N x steps interval; per-GPU work as at N = 1, no collective in the
data path) -- and `n1` -- rank 0 ALONE tracing the strong interval's --steps pulses while the others wait at a barrier --, from
which `efficiency_vs_n1` = value / (N x n1.value): the line is self-contained.  (--scaling weak makes the weak interval the
headline, as round 4 did.)

roofline: the trace kernel moves ~150 MB of HBM per launch against >10 GB of cache-served traversal bytes, so HBM is not
its bound (reported as secondary fields).  What binds is instruction issue: `bound` names the busier of the two per-CU
pipes measured with rocprofv3 counters on this binary and workload (profiles/<tag>_pmc_<workload>.json, produced by
tools/pmc_collect.sh + tools/pmc_derive.py): VALU issue and the vector-memory return path.  VALU issue is priced with
MEASURED per-class costs (tools/valu_calib.hip -> profiles/r03_valu_calib.json: 2.25-2.44 cycles of a SIMD for f32 add /
mul / fma and simple integer instructions, 4.1-4.3 for f64, min / max, compares, selects, 8 / 16 for rcp), the classes
from the counters plus the static opcode mix of the walk loop (tools/isa_mix.py); `frac` = demanded issue cycles / (1024
SIMDs x 2.4 GHz x serial kernel time) <= 1.  The old convention (4 cycles per instruction) is kept as a secondary field.
The profile carries the hash of the sources it was taken on (rts_build_id): if the loaded library was built from other
sources the roofline is NOT priced (frac null) -- collect the counters again.  The kernel time used is the SERIAL one
(one launch alone on the GPU, HIP events on its stream, measured after the timed region); the timed region itself
overlaps the kernels of --inflight pulses.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
PEAK_CLOCK_HZ = 2.4e9          # max shader clock (MI355X_MICROARCH.md)
N_SIMD, N_CU = 1024, 256
PRI = 1.0e-3                   # pulse repetition interval of the synthetic CPI
PMC_TAG = "r05"                # profiles/<tag>_pmc_<workload>.json: counters of the committed kernel (tools/pmc_collect.sh + tools/pmc_derive.py)


def pulse_motion(spec, k):
    """target placement of pulse k; the config's interval is n_pulses long (C3: 256 pulses = 51 m of flight), longer runs
    repeat it so that the workload does not drift out of the beam"""
    k = k % max(int(spec.get("n_pulses", 256)), 1)
    if "motion_fn" in spec:                                    # (a placement with a per-pulse rotation: configs[4], the adapter benchmark's sphere)
        return spec["motion_fn"](k)
    out = []
    for m in spec["motion"]:
        v = np.asarray(m["velocity"], np.float64); p0 = np.asarray(m["position"], np.float64)
        out.append(dict(position=tuple(p0 + v * (k * PRI)), velocity=tuple(v)))
    return out


def bounding_sphere_fraction(spec, tx, motion, max_rays=4_000_000):
    """fraction of a pulse's launch indices whose primary ray meets the bounding sphere of (any) target -- SURVEY section 8d asks that
    the beam be chosen so that this is >= 0.5 and that it be stated.  The lattice directions of ray_generation (ray_tracer.cu:155-203,
    restated in numpy: lattice point, normalise, Rot(azimuth), Rot1(elevation about the rotated y axis)) against the spheres around
    the placed meshes; a strided sample of at most max_rays launch indices."""
    W = spec["W"]; total = W ** 3
    spx, spy, spz = tx["span"]; az, el = tx["dir"]
    s2c = lambda a, e: np.array([math.cos(a) * math.cos(e), math.sin(a) * math.cos(e), math.sin(e)])
    bs, be = s2c(-spx / 2, -spy / 2), s2c(spx / 2, spy / 2)
    st = np.array([((be[0] * (1 + spz)) - bs[0]) / (W - 1), (be[1] - bs[1]) / (W - 1), (be[2] - bs[2]) / (W - 1)]) if W > 1 else np.zeros(3)
    idx = np.arange(0, total, max(total // max_rays, 1), dtype=np.int64)
    lx = idx % W; ly = (idx // W) % W; lz = idx // (W * W)
    v = bs[None, :] + np.stack([lx, ly, lz], 1) * st[None, :]
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    Rot = np.array([[math.cos(az), -math.sin(az), 0], [math.sin(az), math.cos(az), 0], [0, 0, 1.0]])
    o = Rot[:, 1] / np.linalg.norm(Rot[:, 1]); c, sn = math.cos(el), math.sin(el)
    Rot1 = np.array([[c + o[0] * o[0] * (1 - c), o[0] * o[1] * (1 - c) + o[2] * sn, o[0] * o[2] * (1 - c) - o[1] * sn],
                     [o[1] * o[0] * (1 - c) - o[2] * sn, c + o[1] * o[1] * (1 - c), o[1] * o[2] * (1 - c) + o[0] * sn],
                     [o[2] * o[0] * (1 - c) + o[1] * sn, o[2] * o[1] * (1 - c) - o[0] * sn, c + o[2] * o[2] * (1 - c)]])
    d = v @ Rot.T; d /= np.linalg.norm(d, axis=1, keepdims=True); d = d @ Rot1.T
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit = np.zeros(len(idx), bool)
    from rts_amd import scenes as S
    for m, mo in zip(spec["meshes"], motion):
        vw, _ = S.world_vertices(m, mo)
        cen = 0.5 * (vw.min(axis=0) + vw.max(axis=0)); rad = float(np.linalg.norm(vw - cen[None, :], axis=1).max())
        q = cen - np.asarray(tx["origin"], np.float64)
        b = d @ q
        hit |= (b > 0) & ((q @ q) - b * b <= rad * rad)
    return float(hit.mean())


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(spec, seconds_target=12.0):
    """The CPU restatement (oracle, BVH mode, every hardware thread, -O3 -march=native build made on this machine) timed on
    whole pulses of the same workload, repeated until ~seconds_target of CPU work.  kind = "port": the reference has no CPU
    path and cannot be built here."""
    os.environ["RTS_ORACLE_NATIVE"] = "1"
    os.environ["RTS_ORACLE_PIN"] = "1"                     # one worker per hardware thread, pinned (oracle/rts_oracle.cpp: orc_trace)
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    # every hardware thread of the machine: the GPU loop above bound this process to the GPU's NUMA node -- half the host on the two-socket boxes of
    # this pool, where 256 unpinned threads then shared 128 hardware threads and a pulse took 498-749 ms from run to run
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass
    threads = len(os.sched_getaffinity(0)) or 1
    sc = H.oracle_scene(O, spec, pulse_motion(spec, 0))
    tx = spec["tx"]; W = spec["W"]; total = W ** 3
    n = max(20000, min(total // 8, 2_000_000))             # the probe that decides "whole pulses or a sample": large enough that starting (and pinning) the threads is not what it measures
    kw = dict(use_bvh=True, threads=threads, debug=False, reuse_buffers=True)
    sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=total // n, n_rays=n, **kw)   # builds the BVH
    t0 = time.time()
    r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=total // n, n_rays=n, **kw)
    rate = r["counters"]["segments"] / max(time.time() - t0, 1e-3)
    seg = 0; t0 = time.time(); reps = 0
    per_pulse = max(r["counters"]["segments"] * total / n, 1)
    if per_pulse / rate <= seconds_target:                 # whole pulses, repeated until ~seconds_target of CPU work
        times = []
        while time.time() - t0 < seconds_target and reps < 256:
            t1 = time.time()
            r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=1, n_rays=total, **kw)
            times.append(time.time() - t1)
            seg += r["counters"]["segments"]; reps += 1
        best = min(times); seg_pulse = seg / reps
        what = "%d whole pulses (%d launch indices each): best %.1f, median %.1f, worst %.1f ms per pulse" % (reps, total, best * 1e3, float(np.median(times)) * 1e3, max(times) * 1e3)
        dt = time.time() - t0
        return dict(value=seg_pulse / best / 1e6, unit="Mrays/s", cores=threads, kind="port", cpu=cpu_model(), mean_value=seg / dt / 1e6,
                    spread=dict(best_ms=best * 1e3, median_ms=float(np.median(times)) * 1e3, worst_ms=max(times) * 1e3, pulses=reps, worst_over_best=max(times) / best, threads_pinned=True),
                    sample="%s (value = best of %d; mean %.1f Mrays/s): %d segments in %.1f s; oracle/rts_oracle.cpp in BVH mode, g++ -O3 -march=native -ffp-contract=off, %d threads on %s"
                           % (what, reps, seg / dt / 1e6, seg, dt, threads, cpu_model()))
    else:                                                  # a strided sample of one pulse
        stride = max(int(per_pulse / (rate * seconds_target)), 1)
        m = total // stride
        r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=stride, n_rays=m, **kw)
        seg = r["counters"]["segments"]
        what = "every %d-th of the %d launch indices of one pulse" % (stride, total)
    dt = time.time() - t0
    return dict(value=seg / dt / 1e6, unit="Mrays/s", cores=threads, kind="port", cpu=cpu_model(),
                sample="%s: %d segments in %.1f s; oracle/rts_oracle.cpp in BVH mode, g++ -O3 -march=native -ffp-contract=off, %d threads on %s"
                       % (what, seg, dt, threads, cpu_model()))


def load_pmc(workload):
    p = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (PMC_TAG, workload))
    if not os.path.exists(p):
        return None, None
    return json.load(open(p)), os.path.relpath(p, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=0, help="W (launch indices per pulse = W^3); 0 = the config's own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--link", action="store_true", help="rts_link_handles: the handles' trace kernels run strictly one at a time")
    ap.add_argument("--fused-post", action="store_true", help="rts_trace_pulse_end_uniform (a pulse's post-processing enqueued behind its trace on the device-side received count, ONE kernel for a small received set since round 4, no host wait for the trace) instead of rts_trace_pulse_end + rts_finalise_uniform + rts_cube_accumulate + rts_aggregate.  Measured (round 4, DESIGN.md section 5): sequential pulses 1.03 against 1.10 ms, three pulses in flight 0.60 against 0.58 -- the default stays the four calls, except with --inflight 1")
    ap.add_argument("--no-bind", action="store_true", help="leave the process's CPU affinity alone (default: the CPUs of the GPU's NUMA node)")
    ap.add_argument("--post-lag", type=int, default=0, choices=(0, 1), help="1: a pulse's group table is collected one pulse later (the submitting thread does not wait for the post-processing it has just enqueued)")
    ap.add_argument("--inflight", type=int, default=3, help="pulses in flight per GPU; 1 = strictly sequential pulses")
    ap.add_argument("--config", default="c3", choices=["c2", "c2file", "c3", "c3ecef", "c3ico", "c4", "c5", "sphere6"], help="c3 = BASELINE configs[2] (the metric's workload); c4 = configs[3]'s scene and size (100 M launch indices per pulse: give --steps 32; --tx both: its two transmitters in turn); c5 = configs[4]: the C3 airframe re-rotated AND translated every pulse, 1024-pulse interval (give --steps 1024), the transmitter tracking it; sphere6 = the scene of the C++ boundary benchmark (tests/adapter/adapter_bench.cpp)")
    ap.add_argument("--tx", default="0", choices=["0", "1", "both"], help="c4: which of configs[3]'s two transmitters; both = the first half of the interval's pulses from transmitter 0, the second half from transmitter 1 (the reference's transmitter loop is the outer one, ray_tracer.cpp:806)")
    ap.add_argument("--as-rank", default="", help="debug, one process: 'r/N' runs the plan rank r of N ranks would run (--shard rays, the default here: its part of EVERY pulse; --shard pulses: its whole pulses and its parts of the left-over ones), without a process group -- one GPU's share of a ray-sharded interval at the pipelined rate, every r in turn gives the critical path of an N-GPU run; value / ms_per_step are that rank's alone")
    ap.add_argument("--deal", default="auto", choices=["auto", "interleave", "cost"], help="auto = cost for a multi-rank job, interleave with --as-rank.  --shard rays: 'cost' = after the warm-up pulses (traced as interleaved parts) the ranks exchange what every tile cost the rank that traced it (ONE all-reduce of a uint32 per 64 launch indices, outside the timed interval: it belongs to the previous interval), adopt the merged table as their tile history and trace the timed interval's pulses as tile lists dealt longest-first from it (rts_deal_tiles, rts_set_tile_list) instead of the static interleave.  With --as-rank the table comes from two whole pulses traced by this process (standing in for the other ranks)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"], help="N > 1: which interval is the line's value.  'strong' (default, the north star's) = ONE interval of --steps pulses shared by the ranks; 'weak' = every rank runs --steps pulses, the interval is N x steps pulses (per-GPU work fixed as N grows -- pulses are independent, ray_tracer.cpp:843, and there is no collective in the data path).  The other one, and rank 0 alone on the strong interval, are measured by the same job and reported as secondary blocks")
    ap.add_argument("--shard", default="auto", choices=["auto", "pulses", "whole", "rays"], help="N > 1, the strong interval: 'whole' = whole pulses only, contiguous runs, some ranks one pulse more; 'pulses' = whole pulses and each left-over pulse split over a group of ranks; 'rays' = every pulse split over all ranks (tiles dealt by cost).  auto: see the top of this file")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1: skip the secondary intervals (weak / strong counterpart and rank 0 alone)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    # stdout carries the JSON line and nothing else: whatever a library prints there (RCCL's version banner under torch.distributed.run
    # made profiles/r04_bench_rccl_world1.json unparsable) goes to stderr from here on; the line is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1); os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # One process per GPU, on the GPU's socket -- BEFORE this process touches the GPU, so that the runtime's threads, its signal
    # pools and the handles' pinned blocks all come to live there: a child process asks the runtime for the device's PCI address
    # (the parent must not initialise HIP for that), sysfs says which NUMA node that is and which CPUs it has.  On the
    # two-socket hosts of this pool the other socket costs a pipelined pulse ~0.05 ms (profiles/r03c_bind_ab.log).
    numa_node = -1
    profiled = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")      # (a profiler's preloaded library has initialised the GPU already: no child process then -- the library call below binds after the fact)
    if not args.no_bind and not profiled:
        try:
            import subprocess
            q = ("import ctypes as C\nh=C.CDLL('libamdhip64.so')\nn=C.c_int(0)\nh.hipGetDeviceCount(C.byref(n))\nb=C.create_string_buffer(64)\n"
                 "print(b.value.decode() if n.value and h.hipDeviceGetPCIBusId(b,64,%d %% n.value)==0 else '')" % local_rank)
            bdf = subprocess.run([sys.executable, "-c", q], capture_output=True, text=True, timeout=60).stdout.strip().lower()
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read()) if bdf else -1
            if node >= 0:
                cpus = set()
                for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                    a, _, b = part.partition("-"); cpus.update(range(int(a), int(b or a) + 1))
                cpus &= os.sched_getaffinity(0)
                if cpus:
                    os.sched_setaffinity(0, cpus); numa_node = node
        except Exception:                                         # (no sysfs, no permission, another platform: the scheduler's choice stands)
            pass
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the RTS hot path")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()       # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dist = None
    # launched by torch.distributed.run (RANK in the environment): the process group is initialised whatever the world size, so
    # that `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` runs init_process_group / all_gather /
    # all_reduce on RCCL with ONE rank -- the only way to execute those lines on a single-GPU box
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import rts_amd
    from rts_amd import api, scenes, multigpu
    rts_amd._lib.require_built()                                  # never builds (see _lib.require_built)
    if not args.no_bind and numa_node < 0 and hasattr(rts_amd._lib.lib(), "rts_bind_host_to_device"):
        import ctypes
        nn = ctypes.c_int(-1)
        if rts_amd._lib.lib().rts_bind_host_to_device(local_rank % max(torch.cuda.device_count(), 1), ctypes.byref(nn)) == 0:
            numa_node = nn.value
    if args.config == "c3":
        spec = scenes.config3(W=args.width or 216)
    elif args.config == "c3ecef":
        spec = scenes.translate(scenes.config3(W=args.width or 216), scenes.ecef_offset(lat=math.pi / 2))
    elif args.config == "c3ico":                                  # the same airframe tessellated without pole fans (secondary line)
        spec = scenes.config3(W=args.width or 216, ico=True)
    elif args.config == "c4":                                     # BASELINE configs[3]: 4 meshes x 250 k triangles, 8 Rx, W = 465, maxRefl = 8 (transmitter 0)
        spec = scenes.config4(W=args.width or 465)
    elif args.config == "c5":                                     # BASELINE configs[4]: per-pulse rotation + translation (ray_tracer.cpp:993-1014), 1024-pulse interval
        spec = scenes.config5(W=args.width or 216)
    elif args.config == "sphere6":
        spec = scenes.config_sphere6(W=args.width or 216)
    elif args.config == "c2":
        spec = scenes.config2(W=args.width or 100)
    else:
        spec = scenes.config2_file(W=args.width or 100)
    if os.environ.get("RTS_BENCH_RX_FAR"):                     # diagnostic: the receivers moved 1 000 km away -- the same launches, nothing received (what a received set costs the pipeline)
        spec["rx"] = [dict(r, centre=tuple(np.add(r["centre"], (1.0e6, 0.0, 0.0)))) for r in spec["rx"]]
    W = spec["W"]; total = W ** 3
    tx = spec["tx"]; wl = spec["c"] / spec["carrier"]
    if args.tx != "0" and "tx_list" not in spec:
        raise SystemExit("--tx %s: configuration %s has one transmitter" % (args.tx, args.config))
    if args.tx == "1":
        tx = spec["tx_list"][1]
    two_tx = args.tx == "both"
    plan_rank, plan_world = (int(args.as_rank.split("/")[0]), int(args.as_rank.split("/")[1])) if args.as_rank else (rank, world)
    if args.as_rank and (world != 1 or not 0 <= plan_rank < plan_world):
        raise SystemExit("--as-rank r/N needs a single process and 0 <= r < N")
    RAY_SHARD_MIN = 4_000_000                                     # launch indices of a rank's part of a pulse below which ray sharding measures launch overhead (top of this file)

    def resolve_shard(n_pulses, w):
        if args.shard != "auto":
            return args.shard
        if args.as_rank:
            return "rays"
        if w > 1 and n_pulses < w:                             # fewer pulses than ranks -- the latency of single pulses: every pulse split over all ranks by rays when the parts are big enough, else over groups of ranks
            return "rays" if total // w >= RAY_SHARD_MIN else "pulses"
        return "whole"

    # the interval being run: the closures below read it (strong / weak / rank 0 alone differ in these five things only)
    cur = dict(n_int=args.steps * (world if (args.scaling == "weak" and world > 1) else 1), n_warm=args.warmup * (world if (args.scaling == "weak" and world > 1) else 1),
               rank=plan_rank, world=plan_world, shard=None, collectives=True)
    cur["shard"] = resolve_shard(cur["n_int"], plan_world) if args.scaling == "strong" or args.as_rank else ("whole" if args.shard == "auto" else args.shard)
    deal_mode = args.deal if args.deal != "auto" else ("interleave" if args.as_rank else "cost")
    n_int, n_warm = cur["n_int"], cur["n_warm"]

    def half_of():
        return (cur["n_int"] + 1) // 2                            # --tx both: pulses [0, half) of transmitter 0, then [0, n_int - half) of transmitter 1

    def tx_of(k_rel, motion):
        """transmitter of the interval's k_rel-th pulse; a configuration with tx_track aims the boresight at the (first) target's
        position of that pulse (the reference reads the transmitter's rotation per pulse, ray_tracer.cpp:888)"""
        t = tx if not two_tx else spec["tx_list"][0 if k_rel < half_of() else 1]
        if spec.get("tx_track"):
            d = np.asarray(motion[0]["position"], np.float64) - np.asarray(t["origin"], np.float64)
            t = dict(t, dir=(math.atan2(d[1], d[0]), math.atan2(d[2], math.hypot(d[0], d[1]))))
        return t

    def pulse_of(k_rel):
        """pulse number (placement) of the interval's k_rel-th pulse: with two transmitters each runs through the same pulses"""
        return k_rel if not two_tx or k_rel < half_of() else k_rel - half_of()

    # --inflight handles take the pulses in turn; they SHARE one copy of the immutable scene (hierarchy built once)
    trs = []
    t_scene0 = time.perf_counter()
    for i in range(max(args.inflight, 1)):
        t = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank, count_traversal=bool(os.environ.get("RTS_BENCH_COUNT")))      # (RTS_BENCH_COUNT=1: the counting build in the pipelined loop -- diagnostics, e.g. RTS_DEBUG_COOP's clock check)
        if i == 0:
            t.set_scene(spec["meshes"])
        else:
            t.share_scene(trs[0])
        t.set_receivers(spec["rx"])
        t.reserve(0)                                          # set-up, not warm-up: device buffers exist before the first pulse
        if trs and args.link:
            trs[0].link(t)
        trs.append(t)
    scene_setup_s = time.perf_counter() - t_scene0

    # complex return cube [rx][pulse][range bin] (derived product; the dense buffer that is all-reduced over RCCL)
    n_bins = 1024
    r0 = 2.0 * float(np.linalg.norm(np.asarray(tx["origin"], np.float64) - np.asarray(spec["motion"][0]["position"], np.float64)))
    cube_t0 = (r0 - 150.0) / spec["c"]; cube_dt = 300.0 / spec["c"] / n_bins
    has_dop = hasattr(rts_amd._lib.lib(), "rts_cube_doppler")     # (an older library named by RTS_AMD_LIB, A/B runs: torch.fft then, outside the timed region)
    cubes = {}                                                 # one cube per interval LENGTH this job runs (strong: --steps rows, weak: N x steps): the all-reduce and the transform of an interval move its own rows only
    cube = dop = None; n_fft = 0

    def attach_cube(rows):
        """the return cube [rx][rows][bins] of an interval of `rows` pulses, attached to every handle (each pulse owns one row, so the handles share it), and its
        range-Doppler map: zero-padded power-of-two transform over the pulse axis (rts_cube_doppler)"""
        nonlocal cube, dop, n_fft
        rows = max(int(rows), 1)
        if rows not in cubes:
            c_ = torch.zeros((len(spec["rx"]), rows, n_bins), dtype=torch.complex128, device="cuda")
            nf = 1 << max(rows - 1, 1).bit_length()
            d_ = torch.zeros((c_.shape[0], nf, n_bins), dtype=torch.complex128, device="cuda") if (nf <= 4096 and has_dop) else None
            cubes[rows] = (c_, d_, nf)
        cube, dop, n_fft = cubes[rows]
        for t in trs:
            t.cube_attach(cube.shape[0], cube.shape[1], n_bins, cube_t0, cube_dt, device_ptr=cube.data_ptr())
    attach_cube(max(n_int, n_warm))

    dealt = {}                                               # --deal cost: {"tile": launch indices per plan tile, "cost": every rank's share by the records} once the tile lists are set

    def plan(n_pulses):
        if dealt:
            return [(k, 0, total, (dealt["tile"], api.INTERLEAVE_LIST, 0)) for k in range(n_pulses)]
        if cur["shard"] == "rays" and cur["world"] > 1:
            p = multigpu.plan_rays(total, n_pulses, cur["rank"], cur["world"])
        elif cur["shard"] == "whole":
            return multigpu.plan_whole(total, n_pulses, cur["rank"], cur["world"])      # (never refined: a rank with fewer pulses than handles leaves a handle idle rather than trace parts)
        else:
            p = multigpu.plan_cpi(total, n_pulses, cur["rank"], cur["world"])
        return multigpu.refine_plan(p, len(trs))

    def prepare_cpi(k0, n_pulses):
        """the interval's INPUTS: this rank's plan and the target placements of its pulses (synthetic data: generated before the
        clock starts, like the scene)"""
        items = plan(n_pulses)
        motions = [pulse_motion(spec, k0 + pulse_of(k)) for (k, _, _, _) in items]
        return items, motions, [tx_of(k, m) for (k, _, _, _), m in zip(items, motions)]

    def run_cpi(k0, n_pulses, prepared=None):
        """pulses k0 .. k0+n_pulses-1 as one coherent processing interval, sharded over the ranks"""
        parts = []; acc = dict(segments=0, shaded=0, received=0, ms_scene=0.0, ms_trace=0.0, ms_post=0.0, launches=0)
        cube.zero_(); torch.cuda.synchronize()
        t_cpi = time.perf_counter()

        hp = acc.setdefault("host_ms", dict(begin=0.0, end_wait=0.0, post_enqueue=0.0, collect_wait=0.0))      # where the submitting thread spends the interval
        skip_post = os.environ.get("RTS_BENCH_SKIP_POST", "")

        def post(t, k):
            """everything after the trace is only ENQUEUED.  One call (rts_trace_pulse_end_uniform): on the device-side received
            count, without waiting for the trace, when the handle's previous pulse received few rays (--fused-post); default: the
            pulse's trace has to be over first (rts_trace_pulse_end reads the count), then three more calls"""
            h0 = time.perf_counter()
            if fused_post:
                t.trace_end_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"], cube_pulse=k, recv_index_base=rts_amd._lib.RTS_BASE_USE_ROWS)
                hp["post_enqueue"] += (time.perf_counter() - h0) * 1e3
                return
            t.trace_end()
            h1 = time.perf_counter()
            if skip_post == "1":                                  # diagnostic (RTS_BENCH_SKIP_POST=1 | fin | cube | agg: INVALID as a bench line): the received set is ordered and expanded, (part of) the rest left out
                hp["end_wait"] += (h1 - h0) * 1e3; return
            if skip_post != "fin":
                t.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
            if skip_post != "cube":
                t.cube_accumulate(k, spec["c"], spec["carrier"])
            if skip_post != "agg":
                t.aggregate(spec["c"], spec["carrier"], rts_amd._lib.RTS_BASE_USE_ROWS, fetch=False)
            h2 = time.perf_counter()
            hp["end_wait"] += (h1 - h0) * 1e3; hp["post_enqueue"] += (h2 - h1) * 1e3

        def collect(t, k):
            """wait for the pulse's post-processing and take its group table"""
            h0 = time.perf_counter()
            groups = t.groups() if skip_post not in ("1", "agg", "fetch") else np.zeros(0, rts_amd._lib.GROUP_DTYPE)      # (fetch: the chain runs, its table is never asked for)
            hp["collect_wait"] += (time.perf_counter() - h0) * 1e3
            st = t.stats_raw()                                # stream already drained by the table fetch
            parts.append(dict(pulse=k, groups=groups, responses=(api.groups_to_responses(groups) if (dist is None or not cur["collectives"]) and not skip_post else None)))      # (one rank: the pulse's responses are formed here, while the next pulses trace, instead of in the interval's tail)
            acc["segments"] += st.segments; acc["shaded"] += st.shaded; acc["received"] += st.received
            acc["ms_scene"] += st.ms_scene; acc["ms_trace"] += st.ms_trace; acc["ms_post"] += st.ms_compact + st.ms_aggregate
            acc["launches"] += 1
            if bench_debug:
                print("pulse %d done at %.3f ms: scene %.3f trace %.3f compact %.3f agg %.3f" % (k, (time.perf_counter() - t_cpi) * 1e3, st.ms_scene, st.ms_trace, st.ms_compact, st.ms_aggregate), file=sys.stderr)

        # The submitting thread never waits for a chain of small kernels it has just enqueued: with --post-lag 1 (default when
        # there are >= 3 handles) the handles hold, in pulse order, [one pulse whose post-processing runs] [handles - 1 pulses
        # tracing]; the oldest pulse's table is collected just before its handle takes a new pulse.  --post-lag 0: the oldest
        # pulse is completed (trace, post-processing, table) before the next one is begun -- every handle traces.
        lag = args.post_lag if len(trs) >= 2 else 0
        fused_post = (args.fused_post or len(trs) == 1) and hasattr(rts_amd._lib.lib(), "rts_trace_pulse_end_uniform")
        pending = []; posted = []
        items, motions, txs = prepared if prepared is not None else prepare_cpi(k0, n_pulses)
        bench_debug = bool(os.environ.get("BENCH_DEBUG"))
        t_cpi = time.perf_counter()                                   # (the interval's clock for BENCH_DEBUG lines)
        for i, (k, first, count, il) in enumerate(items):
            t = trs[i % len(trs)]
            while any(p[0] is t for p in posted):
                collect(*posted.pop(0))
            h0 = time.perf_counter()
            t.trace_begin(txs[i]["origin"], txs[i]["span"], txs[i]["dir"], motions[i], ray_first=first, ray_count=count, interleave=il)
            hp["begin"] += (time.perf_counter() - h0) * 1e3
            if fused_post:                                    # the whole pulse -- placement, trace, post-processing -- is enqueued in one go ...
                post(t, k); posted.append((t, k))
                if len(posted) == len(trs):                   # ... and the oldest pulse's table collected while the newer ones run
                    collect(*posted.pop(0))
                continue
            pending.append((t, k))
            if len(pending) == len(trs) - lag:                # the oldest pulse in flight is taken further while the newer ones run
                tk = pending.pop(0); post(*tk)
                if lag:
                    posted.append(tk)
                else:
                    collect(*tk)
        while pending:
            tk = pending.pop(0); post(*tk); posted.append(tk)
        while posted:
            collect(*posted.pop(0))
        t_a = time.perf_counter()
        cdist = dist if cur["collectives"] else None           # (rank 0 alone on the strong interval: no collective)
        allp = multigpu.exchange_parts(parts, cdist, torch)   # ONE exchange per CPI (RCCL all-gather), inside the timed region
        t_b = time.perf_counter()
        resp = multigpu.merge_cpi(allp, spec["max_refl"])
        t_c = time.perf_counter()
        if cdist is not None:                                 # dense per-receiver return buffers: sum over the ranks
            if args.backend == "nccl":
                dist.all_reduce(torch.view_as_real(cube), op=dist.ReduceOp.SUM)
            else:
                cc = torch.view_as_real(cube).cpu(); dist.all_reduce(cc, op=dist.ReduceOp.SUM); cube.copy_(torch.view_as_complex(cc))
        t_d = time.perf_counter()
        if dop is not None:                                   # slow-time transform of the (summed) cube: the library's LDS-resident FFT, on the handle's stream
            trs[0].cube_doppler(n_fft, device_ptr=dop.data_ptr(), fetch=False)
            torch.cuda.synchronize()
        acc["tail_ms"] = dict(exchange=(t_b - t_a) * 1e3, merge=(t_c - t_b) * 1e3, cube_reduce=(t_d - t_c) * 1e3, doppler_fft=(time.perf_counter() - t_d) * 1e3)
        return acc, resp

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import gc

    def clear_deal():
        if dealt:
            for t in trs:
                t.set_tile_list(0, np.zeros(0, np.uint32))
            dealt.clear()

    def timed_interval(n_int_, n_warm_, shard_, rank_, world_, collectives_=True, participate=True):
        """warm-up + ONE timed interval under the given sharding; returns (acc, resp, seconds) -- seconds = the slowest rank's when the
        interval uses the process group.  participate = False: this rank only keeps the barriers (rank 0 traces alone)."""
        clear_deal()
        cur.update(n_int=n_int_, n_warm=n_warm_, shard=shard_, rank=rank_, world=world_, collectives=collectives_)
        attach_cube(max(n_int_, n_warm_, 3 * len(trs)))
        acc_ = resp_ = None; dt_ = 0.0
        if participate:
            if n_warm_:
                run_cpi(0, n_warm_)
            if deal_mode == "cost" and shard_ == "rays" and world_ > 1:
                if args.as_rank or not collectives_:                  # one process: two whole pulses stand in for what the other ranks measured
                    m0 = pulse_motion(spec, 0); x0 = tx_of(0, m0)
                    for _ in range(2):
                        trs[0].trace(x0["origin"], x0["span"], x0["dir"], m0, ray_first=0, ray_count=total)
                    table = trs[0].tile_records_get()
                else:
                    table = multigpu.exchange_tile_records([t.tile_records_get() for t in trs], dist, torch)
                tile_, ids_, cost_ = multigpu.dealt_tiles(table, total, rank_, world_)
                for t in trs:
                    t.tile_records_set(table); t.set_tile_list(tile_, ids_)
                dealt.update(tile=tile_, cost=[float(x) for x in cost_ / max(float(cost_.mean()), 1.0)], tiles_of_this_rank=int(ids_.shape[0]))
                run_cpi(n_warm_, 3 * len(trs))                # (every handle's first launches over its list: the order build of a new shape, then the cooperative kernel's grid from the head count of the launch before -- three launches per handle until the schedule is the one the interval runs with)
            prepared = prepare_cpi(n_warm_, n_int_)
            # The harness is Python: its cyclic garbage collector, once a few thousand ctypes / numpy objects have been allocated by the
            # loop, makes full passes over everything torch imported (~40 ms each) -- measured as 0.15 ms per pulse in trace_begin at 256
            # pulses, none at 64 (gpurun_out r04h).  Not the product's time: collected once here, then off for the timed interval.
            gc.collect()
            if not os.environ.get("RTS_BENCH_GC"):             # (RTS_BENCH_GC=1: leave the collector on -- the control run of profiles/r04_fresh_processes_gc.log)
                gc.disable()
        sync()
        t0 = time.perf_counter()
        if participate:
            acc_, resp_ = run_cpi(n_warm_, n_int_, prepared)
        sync()
        dt_ = time.perf_counter() - t0
        gc.enable()
        if participate:
            expected = n_int_ if not args.as_rank else len({it[0] for it in prepared[0]})      # (--as-rank: one process stands for one rank, it sees the pulses of that rank's plan only)
            assert len(resp_) == expected, "every pulse of the interval must come back with its responses"
        if dist is not None and collectives_:
            rdev = "cuda" if args.backend == "nccl" else "cpu"
            tt = torch.tensor([dt_], dtype=torch.float64, device=rdev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt_ = float(tt.item())
        return acc_, resp_, dt_

    def job_sums(acc_):
        """(segments, shaded, received) over all ranks"""
        v = [acc_["segments"], acc_["shaded"], acc_["received"]]
        if dist is None:
            return v
        rdev = "cuda" if args.backend == "nccl" else "cpu"
        ss = torch.tensor(v, dtype=torch.float64, device=rdev); dist.all_reduce(ss, op=dist.ReduceOp.SUM)
        return [int(x) for x in ss.tolist()]

    # ---- the line's interval
    acc, resp, dt = timed_interval(n_int, n_warm, cur["shard"], plan_rank, plan_world)
    headline = dict(scaling=args.scaling, shard=cur["shard"], n_int=n_int, deal=dict(dealt) if dealt else None)
    # range-Doppler map of the interval (slow-time FFT of the summed cube): a check of the dense product, outside the timed
    # region -- the hot path ends with the per-pulse responses and the (all-reduced) cube
    # (the transform itself ran inside the timed region, in the interval's tail: rts_cube_doppler; here it is checked against torch.fft)
    ref_rd = torch.fft.fft(cube, n=n_fft, dim=1) if dop is not None else torch.fft.fft(cube, dim=1)
    range_doppler_peak = float((dop if dop is not None else ref_rd).abs().max().item()) if args.steps > 0 else 0.0
    range_doppler_err = float((dop - ref_rd).abs().max().item() / max(float(ref_rd.abs().max().item()), 1e-300)) if dop is not None else None
    assert range_doppler_err is None or range_doppler_err < 1e-9, "rts_cube_doppler disagrees with torch.fft (%g)" % range_doppler_err
    cube_desc = "complex128 [%d rx][%d pulses][%d bins], all-reduced once per interval, then %d-point slow-time FFT in the library (rts_cube_doppler, inside the timed region); range-Doppler peak %.6e, max deviation from torch.fft %.1e (relative)" % (cube.shape[0], cube.shape[1], n_bins, n_fft, range_doppler_peak, range_doppler_err or 0.0)
    seg = acc["segments"]; ms_trace = acc["ms_trace"]; ms_scene = acc["ms_scene"]; ms_post = acc["ms_post"]
    shaded = acc["shaded"]; received = acc["received"]; launches = max(acc["launches"], 1)
    seg_all, shaded_all, received_all = job_sums(acc)              # whole-job aggregates (dt is already the slowest rank's)

    # ---- secondary intervals of a multi-rank job: the other scaling mode, and rank 0 alone on the strong interval
    secondary = {}
    if world > 1 and not args.no_secondary and not args.as_rank:
        other = "weak" if args.scaling == "strong" else "strong"
        o_int = args.steps * (world if other == "weak" else 1); o_warm = min(args.warmup, 2 * len(trs)) * (world if other == "weak" else 1)
        o_shard = "whole" if other == "weak" else resolve_shard(o_int, world)
        acc_o, _, dt_o = timed_interval(o_int, o_warm, o_shard, rank, world)
        seg_o = job_sums(acc_o)[0]
        secondary[other] = dict(value=seg_o / dt_o / 1e6, unit="Mrays/s", ms_per_step=dt_o / args.steps * 1e3, ms_per_pulse_of_the_interval=dt_o / o_int * 1e3, pulses_in_the_interval=o_int, shard=o_shard,
                                note=("every rank traces --steps whole pulses; ms_per_step = wall time / --steps (one pulse on every rank)" if other == "weak" else "ONE interval of --steps pulses shared by the ranks"))
        acc_1, _, dt_1 = timed_interval(args.steps, min(args.warmup, 2 * len(trs)), "whole", 0, 1, collectives_=False, participate=(rank == 0))
        if rank == 0:
            n1_value = acc_1["segments"] / dt_1 / 1e6
            secondary["n1"] = dict(value=n1_value, unit="Mrays/s", ms_per_step=dt_1 / args.steps * 1e3,
                                   note="rank 0 alone, the strong interval's --steps pulses, in this job after the timed region (the other ranks wait at the barrier)")
            strong_value = (seg_all / dt / 1e6) if args.scaling == "strong" else secondary["strong"]["value"]
            secondary["efficiency_vs_n1"] = strong_value / (world * n1_value)

    if rank == 0:
        # ---- un-timed measurements behind the roofline block (rank 0, after the timed region)
        # (1) the SERIAL trace kernel: whole pulses one after the other on one handle; HIP events on the kernel's own stream
        ser = []
        for k in range(12):
            mo = pulse_motion(spec, args.warmup + k); tk = tx_of(0, mo)
            st = trs[0].trace(tk["origin"], tk["span"], tk["dir"], mo)
            if k >= 2:
                ser.append((st["ms_trace"], st["segments"]))
        ms_serial = float(np.mean([s[0] for s in ser])); seg_serial = float(np.mean([s[1] for s in ser]))
        # (1b) the serial launch as a BULK and a tail: when its persistent blocks end (rts_get_block_timeline; a handle created with RTS_TIMELINE_BLOCKS=1, sharing the
        #      scene and its tile-cost history).  Median block end = the bulk: every block resident, the chip full; the rest is a few dozen tiles that are ONE ray's chain of
        #      ~1 000 dependent walk steps each, filled by the next pulse's blocks when pulses are pipelined
        bulk = None
        if hasattr(rts_amd._lib.lib(), "rts_get_block_timeline") and not os.environ.get("RTS_BENCH_COUNT"):
            os.environ["RTS_TIMELINE_BLOCKS"] = "1"
            try:
                trb = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank)
            finally:
                del os.environ["RTS_TIMELINE_BLOCKS"]
            trb.share_scene(trs[0]); trb.set_receivers(spec["rx"])
            bl = []
            for k in range(8):
                mo = pulse_motion(spec, args.warmup + k); tk = tx_of(0, mo)
                st = trb.trace(tk["origin"], tk["span"], tk["dir"], mo)
                if k >= 2:
                    b = trb.block_timeline(); bl.append((b["end_p50"] * 1e-3, b["end_p90"] * 1e-3, b["end_last"] * 1e-3, st["ms_trace"], b["blocks"]))
            trb.close()
            bulk = dict(kernel_ms_bulk=float(np.median([b[0] for b in bl])), block_end_p90_ms=float(np.median([b[1] for b in bl])), block_end_last_ms=float(np.median([b[2] for b in bl])),
                        kernel_ms_events=float(np.median([b[3] for b in bl])), blocks=int(bl[0][4]),
                        what="lone launches of a handle that records when its persistent blocks end (100 MHz counter): kernel_ms_bulk = the median block's end -- every block resident until then; block_end_last_ms = the launch")
        # (2) traversal counts and the primary hit fraction: one pulse of the counting build, one pulse with maxRefl = 1
        #     (every primary hit then spawns exactly one more segment, so hit fraction = (segments - rays) / rays)
        trc = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank, count_traversal=True)
        trc.share_scene(trs[0]); trc.set_receivers(spec["rx"])
        mo = pulse_motion(spec, args.warmup); tk = tx_of(0, mo)
        sc = trc.trace(tk["origin"], tk["span"], tk["dir"], mo)
        trc.close()
        tr1 = api.Tracer(W, 1, 0, spec["smooth"], device=local_rank)
        tr1.share_scene(trs[0]); tr1.set_receivers(spec["rx"])
        s1 = tr1.trace(tk["origin"], tk["span"], tk["dir"], mo)
        tr1.close()
        hit_fraction = (s1["segments"] - s1["rays"]) / max(s1["rays"], 1)
        sphere_fraction = bounding_sphere_fraction(spec, tk, mo)
        walked_per_pulse = float(sc.get("walked_segments", 0))       # segments that entered a hierarchy at all (counting build: the others were cleared by the pre-filter or by the bounding spheres)
        V = sc["node_visits"] / max(sc["segments"], 1); T = sc["tri_tests"] / max(sc["segments"], 1); Hh = sc["shaded"] / max(sc["segments"], 1)
        # (3) dense control: the beam squeezed onto the fuselage, (nearly) every launch index hits and bounces
        dense = None
        if args.config in ("c3", "c3ecef", "c3ico"):
            dtx = dict(tx, span=(0.004, 0.004, 0.1)); dn = []
            for k in range(10):                              # (the beam changes shape: the handle's tile order needs a few launches to follow -- the settled ones are counted)
                st = trs[0].trace(dtx["origin"], dtx["span"], dtx["dir"], pulse_motion(spec, args.warmup + k))
                if k >= 4:
                    dn.append((st["ms_trace"], st["segments"]))
            dense = dict(Gseg_per_s=float(np.mean([s[1] for s in dn]) / np.mean([s[0] for s in dn]) / 1e6),
                         kernel_ms=float(np.mean([s[0] for s in dn])), segments=float(np.mean([s[1] for s in dn])),
                         what="same scene, beam span 0.004 x 0.004 rad: every launch index hits the airframe")

        # ---- roofline from the committed counters of this binary on this workload
        pmc, pmc_src = load_pmc({"c3ecef": "c3", "c3ico": "c3"}.get(args.config, args.config))
        try:
            lib_hash = rts_amd._lib.build_id()
        except AttributeError:                                  # a library from before rts_build_id (RTS_AMD_LIB)
            lib_hash = None
        pmc_stale = pmc is not None and pmc.get("source_hash") != lib_hash
        pmc_ok = pmc is not None and not pmc_stale and world == 1 and ((W == 216 and args.config in ("c3", "c3ecef", "c5")) or (W == 100 and args.config == "c2") or (W == 465 and args.config == "c4" and args.tx == "0"))
        roof = {"kernel": "k_trace", "kernel_ms_serial": ms_serial, "segments_per_launch": seg_serial, "walked_segments_per_launch": walked_per_pulse,
                "walked_Gseg_per_s_serial": walked_per_pulse / max(ms_serial, 1e-9) / 1e6,
                "kernel_ms_overlapped_avg": ms_trace / launches, "gpu_ms_per_launch_timed_region": dt / args.steps * 1e3,
                "hit_fraction": hit_fraction, "nodes_per_segment": V, "tri_tests_per_segment": T, "shaded_per_segment": Hh,
                "dense_control": dense, "serial_launch_anatomy": bulk, "counters_source": pmc_src if pmc_ok else None}
        # SURVEY.md section 8(d), figure (B): 288 B of ray state per segment, 128 B per BVH4 node visit, 80 B per triangle test
        # (leaf record), 96 B per shaded hit -- cache-served bytes; as a fraction of the HBM peak it is a yardstick of
        # traversal rate only and may exceed 1 (secondary field, labelled)
        bytes_per_seg = 288.0 + 128.0 * V + 80.0 * T + 96.0 * Hh
        alg = bytes_per_seg * seg_serial
        roof["secondary_hbm_yardstick"] = {"algorithmic_bytes_per_launch": alg, "bytes_per_segment": bytes_per_seg,
                                           "GBps_if_it_were_hbm": alg / (ms_serial * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms_serial * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "note": "cache-served bytes priced as if they came from HBM (SURVEY 8d figure B): NOT a bound" + (": the 20 MB scene lives in L1/L2" if args.config != "c4" else ": configs[3]'s ~0.5 GB of records do NOT fit the 8 x 4 MB of L2 -- the measured fabric traffic is roofline.traffic (profiles/r04_c4_xcd_affine_pmc.log)")}
        if pmc_ok:
            d = pmc["derived"]; pl = pmc["per_launch"]
            seg_prof = float(pmc.get("segments_per_launch") or seg_serial)
            scale = seg_serial / seg_prof                           # this run's launch vs the profiled one (same workload: ~1)
            t_s = ms_serial * 1e-3
            valu = d["valu_wave_insts"] * scale                     # VALU wave instructions per launch
            td = d["vmem_rd_wave_insts"] * 16.0 * scale             # data cycles of the 256 L1 -> register return paths per launch: 16 clk (64 x 16 B at 64 B/clk) per wave-wide load, counted (SQ_INSTS_VMEM_RD); TD_TD_BUSY is not used: it counts cycles with requests outstanding (0.85 on an empty launch)
            f_valu4 = valu * 4.0 / (N_SIMD * PEAK_CLOCK_HZ * t_s)   # the old convention: every instruction 4 cycles
            vcyc = d.get("valu_issue_cycles_calibrated")
            f_valu = (vcyc * scale / (N_SIMD * PEAK_CLOCK_HZ * t_s)) if vcyc else f_valu4
            f_td = td / (N_CU * PEAK_CLOCK_HZ * t_s)
            hbm = d.get("hbm_bytes_per_launch")
            if f_valu >= f_td:
                roof.update(bound="valu_issue", achieved=(vcyc * scale if vcyc else valu * 4.0) / t_s / 1e9, peak=N_SIMD * PEAK_CLOCK_HZ / 1e9, unit="G SIMD issue cycles/s", frac=f_valu,
                            valu_cycles_per_inst=d.get("valu_cycles_per_inst_calibrated"), valu_classes=d.get("valu_classes"), frac_if_every_inst_cost_4_cycles=f_valu4)
            else:
                roof.update(bound="vmem_issue", achieved=td / t_s / 1e9, peak=N_CU * PEAK_CLOCK_HZ / 1e9, unit="G return-path data cycles/s", frac=f_td)
            if bulk and pmc.get("coop_per_launch"):                # (a launch with a cooperative kernel beside the ordinary one: the counts are both kernels', the bulk the ordinary kernel's)
                bulk.update(note="the cooperative kernel runs beside this (ordinary) kernel to the end of the launch (kernel_ms_events): the launch's instruction counts are not this bulk's")
            elif bulk:                                              # the same counts over the launch's BULK (all blocks resident): how busy the chip is while it is full
                t_b = bulk["kernel_ms_bulk"] * 1e-3
                bulk.update(valu_issue_frac=(vcyc * scale if vcyc else valu * 4.0) / (N_SIMD * PEAK_CLOCK_HZ * t_b), vmem_return_path_frac=td / (N_CU * PEAK_CLOCK_HZ * t_b),
                            note="the launch's instruction counts over its bulk alone (an upper estimate by the tail's share of the instructions: a few dozen of ~26 000 live tiles)")
            t_w = dt / args.steps                                   # GPU time per launch in the timed region (kernels of --inflight pulses overlap)
            roof.update(valu_issue_frac=f_valu, vmem_return_path_frac=f_td, source_hash=lib_hash,
                        timed_region={"valu_issue_frac": (vcyc * scale if vcyc else valu * 4.0) / (N_SIMD * PEAK_CLOCK_HZ * t_w), "vmem_return_path_frac": td / (N_CU * PEAK_CLOCK_HZ * t_w),
                                      "note": "the same per-launch counters over wall time / launches of the timed region: up to --inflight trace kernels share the chip, so a launch costs less wall time than the serial kernel (kernel_ms_serial x launches > ms_per_step x steps is expected unless --link / --inflight 1)"},
                        in_profile={"valu_busy": d.get("valu_busy"), "vmem_return_frac_from_counts": d.get("vmem_return_frac_from_counts"), "td_cycles_with_requests_frac": d.get("td_busy"), "ta_busy": d.get("ta_busy"), "l1_hit_rate": d.get("l1_hit_rate"),
                                    "l2_hit_rate": d.get("l2_hit_rate"), "waves_per_simd_avg": d.get("waves_per_simd_avg"), "active_lanes_per_valu_inst": d.get("active_lanes_per_valu_inst"),
                                    "note": "utilisations formed inside the profiled passes (busy cycles over GRBM_GUI_ACTIVE of the same dispatches)"},
                        traffic=hbm, traffic_source="%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of this binary on this workload; static, not re-measured by this run" % pmc_src,
                        hbm_GBps_measured=(hbm / t_s / 1e9) if hbm else None, hbm_frac_of_peak=(hbm / t_s / 1e9 / HBM_PEAK_GBS) if hbm else None,
                        note="bound = the busier of VALU issue (measured cycles of one of 1024 SIMDs per wave-instruction, by class: profiles/r03_valu_calib.json, r04_isa_mix.json) and the vector-memory return path (16 clk of one of 256 CUs per wave-wide load), both from instruction COUNTS of the profiled launch over this run's serial kernel time at the 2.4 GHz peak clock; HBM carries ~1 % of its peak (hbm_frac_of_peak) -- the scene is cache resident, the kernel is issue bound (DESIGN.md section 5)")
        else:
            roof.update(bound="valu_issue", achieved=None, peak=N_SIMD * PEAK_CLOCK_HZ / 1e9, unit="G SIMD issue cycles/s", frac=None, traffic=None, source_hash=lib_hash,
                        note=("the committed counter profile %s was taken on sources %s, this library is %s: not priced -- run tools/pmc_collect.sh + tools/pmc_derive.py again" % (pmc_src, pmc.get("source_hash"), lib_hash)) if pmc_stale
                        else "no committed counter profile for this configuration (profiles/%s_pmc_*.json cover c3 / c5 at W = 216, c2 at W = 100 and c4 -- transmitter 0 -- at W = 465 on one GPU)" % PMC_TAG)
        out = {
            "metric": "Mrays/s (primary+bounces) & ms/pulse, 100k-tri scene",
            "value": seg_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]%s: %s, 1 Tx / %d Rx, W=%d (%d launch indices/pulse), maxRefl=%d, target moves every pulse (re-placed on the device per pulse; static target-space BVH4)"
                                   % ({"c3": 2, "c3ecef": 2, "c3ico": 2, "c4": 3, "c5": 4}.get(args.config, 1), " at Earth-centred coordinates" if args.config == "c3ecef" else (" (both transmitters in turn: pulses [0, %d) from Tx 0, [%d, %d) from Tx 1; a receiver's noise temperature grows by the signal's once per transmitter, ray_tracer.cpp:829 -- host side, the SOARS adapter's)" % (half_of(), half_of(), args.steps) if two_tx else (" -- NOT a BASELINE configuration: the scene of the C++ boundary benchmark (tests/adapter/adapter_bench.cpp)" if args.config == "sphere6" else (" (target re-rotated and translated every pulse, ray_tracer.cpp:993-1014; the transmitter's boresight tracks it)" if args.config == "c5" else ""))), spec["name"], len(spec["rx"]), W, total, spec["max_refl"]),
                       "rays_per_pulse": total, "pulses_in_the_interval": n_int, "segments_per_pulse": seg_all / max(n_int, 1), "received_per_pulse": received_all / max(n_int, 1),
                       "hit_fraction": hit_fraction, "bounding_sphere_fraction": sphere_fraction, "primary_Mrays_per_s": total * n_int / dt / 1e6,
                       "walked_segments_per_pulse": walked_per_pulse, "walked_Mrays_per_s": walked_per_pulse * (seg_all / max(seg_serial * n_int, 1)) * n_int / dt / 1e6,
                       "walked_note": "segments that entered a target's hierarchy (counting build, one pulse); the rest of segments_per_pulse are primaries the conservative pre-filter or the bounding spheres cleared -- counted as rtTrace calls (SURVEY 8d), but bulk culling, not traversal",
                       "dense_control_Gseg_per_s": (dense or {}).get("Gseg_per_s"),
                       "return_cube": cube_desc,
                       "as_rank": args.as_rank or None, "deal": (dict(headline["deal"], how="tiles dealt longest-first from the cost records of the warm-up interval (one all-reduce), rts_deal_tiles") if headline["deal"] else "static interleave") if headline["shard"] == "rays" and plan_world > 1 else None,
                       "sharding": ("%d-pulse interval over %d ranks (--scaling %s: %s), --shard %s -> %s: " % (n_int, world, args.scaling, "--steps pulses per rank" if args.scaling == "weak" else "--steps pulses in all", args.shard, headline["shard"])) + ("every pulse split over all ranks, tiles of 4096 launch indices %s" % ("dealt longest-first from the warm-up interval's cost records" if headline["deal"] else "interleaved") if headline["shard"] == "rays" else ("whole pulses only (contiguous runs, some ranks one pulse more)" if headline["shard"] == "whole" else "whole pulses, left-over pulses in interleaved 4096-index tiles")) + "; one group-table all-gather + one cube all-reduce per interval",
                       "ms_per_pulse_of_the_interval": dt / max(n_int, 1) * 1e3,
                       "host_numa_node": numa_node, "post_processing_call": "rts_trace_pulse_end_uniform" if ((args.fused_post or len(trs) == 1) and hasattr(rts_amd._lib.lib(), "rts_trace_pulse_end_uniform")) else "rts_trace_pulse_end + rts_finalise_uniform + rts_cube_accumulate + rts_aggregate", "pulses_in_flight": len(trs), "linked": bool(args.link), "scene_setup_s": scene_setup_s,
                       "interval_tail_ms_rank0": acc["tail_ms"],
                       "host_ms_per_pulse_rank0": {k: v / max(acc["launches"], 1) for k, v in acc["host_ms"].items()},
                       "stage_ms_per_launch_rank0": {"scene_placement": ms_scene / launches, "trace": ms_trace / launches, "order+finalise+aggregate": ms_post / launches}},
            "roofline": roof,
        }
        out.update(secondary)                                       # N > 1: "weak" (or "strong"), "n1", "efficiency_vs_n1"
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(spec)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())     # the ONE line of this process's real stdout (everything else went to stderr, see main)
    if api._PY_LAP:
        print("python side of trace_begin, us per call:", {k: round(v / max(api._PY_LAP["n"], 1) * 1e6, 1) for k, v in api._PY_LAP.items() if k != "n"}, file=sys.stderr)
    for t in trs:
        t.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
