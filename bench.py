#!/usr/bin/env python3
"""bench.py -- RTS hot path on MI355X: Mrays/s and ms/pulse on the BASELINE.json metric config.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] -- aircraft-like 100 000-triangle mesh,
1 Tx / 4 Rx, W = 216 (10 077 696 launch indices per pulse), maxRefl = 6; the configuration the
metric ("Mrays/s ... 100k-tri scene") is quoted on.  Synthetic mesh, isotropic antennas, RCS 1.

A step is ONE PULSE end to end: target placement for that pulse (the target moves every pulse: world-space
vertices, normals and leaf records are re-placed on the device inside the timed region; the hierarchy itself is
static in target space because the reference's targets are rigid -- the reference instead has OptiX rebuild its
acceleration structure every pulse), trace of the pulse's W^3 launch indices, ordering +
expansion of the received rays, finalisation and group-by aggregation into the pulse's
responses.  A "ray" in Mrays/s is one traced segment (one rtTrace of the reference: primary or
bounce).  Pulses are independent, so each GPU keeps three of them in flight (--inflight handles, each with its own
streams): while one pulse's trace kernel finishes its last slow tiles the next handle's trace blocks already fill the
freed CUs, and the scene placement / ordering / finalisation / aggregation of the neighbouring pulses run beside them.
All of that is inside the timed region; ms_per_step is wall time / pulses.  roofline.kernel_ms_avg is the mean
duration of a trace kernel as it ran, i.e. overlapped with its neighbours.

Scaling is strong: the K timed pulses form one coherent processing interval whose K * W^3
(pulse, launch index) pairs are dealt to the N ranks (rts_amd/multigpu.py: whole pulses first; each of the K % N
left-over pulses is shared by a group of ranks in interleaved 4096-index tiles; K = 1 is plain ray sharding).  The per-(receiver,
path) group tables of all pulse parts are exchanged ONCE, at the end of the timed region and inside
it, by an all-gather over RCCL, and merged into the per-pulse responses on every rank.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
PRI = 1.0e-3                   # pulse repetition interval of the synthetic CPI


def pulse_motion(spec, k):
    """target placement of pulse k; the config's interval is n_pulses long (C3: 256 pulses = 51 m of flight), longer runs
    repeat it so that the workload does not drift out of the beam"""
    k = k % max(int(spec.get("n_pulses", 256)), 1)
    out = []
    for m in spec["motion"]:
        v = np.asarray(m["velocity"], np.float64); p0 = np.asarray(m["position"], np.float64)
        out.append(dict(position=tuple(p0 + v * (k * PRI)), velocity=tuple(v)))
    return out


def cpu_baseline(spec, seconds_target=12.0):
    """The CPU restatement (oracle, BVH mode, all host threads) timed on whole pulses of the same
    workload, repeated until ~seconds_target of CPU work.  kind = "port": the reference has no CPU
    path and cannot be built here."""
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    threads = min(os.cpu_count() or 1, 64)
    sc = H.oracle_scene(O, spec, pulse_motion(spec, 0))
    tx = spec["tx"]; W = spec["W"]; total = W ** 3
    n = 20000
    kw = dict(use_bvh=True, threads=threads, debug=False)
    sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=total // n, n_rays=n, **kw)   # builds the BVH
    t0 = time.time()
    r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=total // n, n_rays=n, **kw)
    rate = r["counters"]["segments"] / max(time.time() - t0, 1e-3)
    seg = 0; rays = 0; t0 = time.time(); reps = 0
    per_pulse = max(r["counters"]["segments"] * total / n, 1)
    if per_pulse / rate <= seconds_target:                 # whole pulses, repeated until ~seconds_target of CPU work
        while time.time() - t0 < seconds_target and reps < 256:
            r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=1, n_rays=total, **kw)
            seg += r["counters"]["segments"]; rays += total; reps += 1
        what = "%d whole pulses (%d launch indices each)" % (reps, total)
    else:                                                  # a strided sample of one pulse
        stride = max(int(per_pulse / (rate * seconds_target)), 1)
        m = total // stride
        r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0, ray_stride=stride, n_rays=m, **kw)
        seg = r["counters"]["segments"]; rays = m
        what = "every %d-th of the %d launch indices of one pulse" % (stride, total)
    dt = time.time() - t0
    return dict(value=seg / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample="%s: %d segments in %.1f s, oracle BVH mode, %d threads" % (what, seg, dt, threads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--width", type=int, default=216, help="W (launch indices per pulse = W^3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--link", action="store_true", help="rts_link_handles: the handles' trace kernels run strictly one at a time (clean single-kernel timings, ~20 %% slower)")
    ap.add_argument("--inflight", type=int, default=3, help="pulses in flight per GPU (linked handles); 1 = strictly sequential pulses")
    ap.add_argument("--config", default="c3", choices=["c2", "c3"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the RTS hot path")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()       # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import rts_amd
    from rts_amd import api, scenes, multigpu
    rts_amd.build()
    spec = scenes.config3(W=args.width) if args.config == "c3" else scenes.config2(W=args.width if args.width != 216 else 100)
    W = spec["W"]; total = W ** 3
    tx = spec["tx"]; wl = spec["c"] / spec["carrier"]

    # --inflight handles hold the same scene and take the pulses in turn: the scene placement of the next pulse and the
    # ordering/aggregation of the previous one overlap with the trace kernels, and the tail of one trace kernel (a few slow
    # tiles) is filled by the blocks of the next handle's (--link serialises the trace kernels instead)
    trs = []
    for _ in range(max(args.inflight, 1)):
        t = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank)
        t.set_scene(spec["meshes"]); t.set_receivers(spec["rx"])
        t.reserve(0)                                          # set-up, not warm-up: device buffers exist before the first pulse
        if trs and args.link:
            trs[0].link(t)
        trs.append(t)
    tr = trs[0]

    # complex return cube [rx][pulse][range bin] (derived product; the dense buffer that is all-reduced over RCCL)
    n_bins = 1024; r0 = 2.0 * abs(tx["origin"][0])
    cube_t0 = (r0 - 150.0) / spec["c"]; cube_dt = 300.0 / spec["c"] / n_bins
    cube = torch.zeros((len(spec["rx"]), max(args.steps, args.warmup, 1), n_bins), dtype=torch.complex128, device="cuda")
    for t in trs:                                              # every pulse owns one row of the cube, so the handles can share it
        t.cube_attach(cube.shape[0], cube.shape[1], n_bins, cube_t0, cube_dt, device_ptr=cube.data_ptr())

    def run_cpi(k0, n_pulses):
        """pulses k0 .. k0+n_pulses-1 as one coherent processing interval, sharded over the ranks"""
        parts = []; acc = dict(segments=0, shaded=0, received=0, ms_scene=0.0, ms_trace=0.0, ms_post=0.0, launches=0)
        cube.zero_(); torch.cuda.synchronize()
        t_cpi = time.perf_counter()
        def finish(t, k):
            t.trace_end()
            t.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
            t.cube_accumulate(k, spec["c"], spec["carrier"])
            groups = t.aggregate(spec["c"], spec["carrier"], rts_amd._lib.RTS_BASE_USE_ROWS)
            st = t.stats()                                    # stream already drained by the aggregation's table fetch
            parts.append(dict(pulse=k, groups=groups))
            acc["segments"] += st["segments"]; acc["shaded"] += st["shaded"]; acc["received"] += st["received"]
            acc["ms_scene"] += st["ms_scene"]; acc["ms_trace"] += st["ms_trace"]; acc["ms_post"] += st["ms_compact"] + st["ms_aggregate"]
            acc["launches"] += 1
            if os.environ.get("BENCH_DEBUG"):
                print("pulse %d done at %.3f ms: scene %.3f trace %.3f compact %.3f agg %.3f" % (k, (time.perf_counter() - t_cpi) * 1e3, st["ms_scene"], st["ms_trace"], st["ms_compact"], st["ms_aggregate"]), file=sys.stderr)

        pending = []
        for i, (k, first, count, il) in enumerate(multigpu.refine_plan(multigpu.plan_cpi(total, n_pulses, rank, world), len(trs))):
            t = trs[i % len(trs)]
            t.trace_begin(tx["origin"], tx["span"], tx["dir"], pulse_motion(spec, k0 + k), ray_first=first, ray_count=count, interleave=il)
            pending.append((t, k))
            if len(pending) == len(trs):                      # the oldest pulse in flight is completed while the newer ones run
                finish(*pending.pop(0))
        while pending:
            finish(*pending.pop(0))
        t_a = time.perf_counter()
        allp = multigpu.exchange_parts(parts, dist, torch)    # ONE exchange per CPI (RCCL all-gather), inside the timed region
        t_b = time.perf_counter()
        resp = multigpu.merge_cpi(allp, spec["max_refl"])
        t_c = time.perf_counter()
        if dist is not None:                                  # dense per-receiver return buffers: sum over the ranks
            if args.backend == "nccl":
                dist.all_reduce(torch.view_as_real(cube), op=dist.ReduceOp.SUM)
            else:
                cc = torch.view_as_real(cube).cpu(); dist.all_reduce(cc, op=dist.ReduceOp.SUM); cube.copy_(torch.view_as_complex(cc))
        acc["range_doppler_peak"] = float(torch.fft.fft(cube[:, :n_pulses], dim=1).abs().max().item()) if n_pulses > 0 else 0.0
        acc["tail_ms"] = dict(exchange=(t_b - t_a) * 1e3, merge=(t_c - t_b) * 1e3, cube_reduce_fft=(time.perf_counter() - t_c) * 1e3)
        return acc, resp

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_cpi(0, args.warmup)
    torch.fft.fft(cube[:, :args.steps], dim=1)            # build the slow-time FFT plan for the timed shape outside the timed region
    sync()
    t0 = time.perf_counter()
    acc, resp = run_cpi(args.warmup, args.steps)
    sync()
    dt = time.perf_counter() - t0
    assert len(resp) == args.steps, "every pulse of the interval must come back with its responses"
    seg = acc["segments"]; ms_trace = acc["ms_trace"]; ms_scene = acc["ms_scene"]; ms_post = acc["ms_post"]
    shaded = acc["shaded"]; received = acc["received"]; launches = max(acc["launches"], 1)

    # whole-job aggregates: max time over ranks, sum of segments
    if dist is not None:
        rdev = "cuda" if args.backend == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
        ss = torch.tensor([seg, shaded, received], dtype=torch.float64, device=rdev); dist.all_reduce(ss, op=dist.ReduceOp.SUM)
        seg_all, shaded_all, received_all = [int(x) for x in ss.tolist()]
    else:
        seg_all, shaded_all, received_all = seg, shaded, received

    if rank == 0:
        # traversal counts for the roofline accounting: one untimed pulse of the counting build
        trc = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank, count_traversal=True)
        trc.set_scene(spec["meshes"]); trc.set_receivers(spec["rx"])
        sc = trc.trace(tx["origin"], tx["span"], tx["dir"], pulse_motion(spec, args.warmup))
        trc.close()
        V = sc["node_visits"] / max(sc["segments"], 1); T = sc["tri_tests"] / max(sc["segments"], 1); Hh = sc["shaded"] / max(sc["segments"], 1)
        # SURVEY.md section 8(d), figure (B), with this build's record sizes: 288 B of ray state per segment, 128 B per
        # BVH4 node visit (SURVEY assumed 64-B BVH2 nodes; V is counted in BVH4 nodes), 80 B per triangle test (leaf record),
        # 96 B per shaded hit (3 normals + velocity)
        bytes_per_seg = 288.0 + 128.0 * V + 80.0 * T + 96.0 * Hh
        seg_per_launch = seg / launches                                    # rank 0's launches
        ms_launch = ms_trace / launches
        achieved = bytes_per_seg * seg_per_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        # HBM traffic cannot be read live (PMC needs rocprofv3 passes of their own): taken from the committed profile of
        # THIS workload (profiles/r01_pmc_k_trace.json), null for any other configuration
        traffic = None; traffic_src = None
        pj = os.path.join(ROOT, "profiles", "r01_pmc_k_trace.json")
        if args.config == "c3" and W == 216 and world == 1 and os.path.exists(pj):
            traffic = json.load(open(pj))["hbm_bytes_per_launch"]; traffic_src = "profiles/r01_pmc_k_trace.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        out = {
            "metric": "Mrays/s (primary+bounces) & ms/pulse, 100k-tri scene",
            "value": seg_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: %s, 1 Tx / %d Rx, W=%d (%d launch indices/pulse), maxRefl=%d, target moves every pulse (re-placed on the device per pulse; static target-space BVH4)"
                                   % (spec["name"], len(spec["rx"]), W, total, spec["max_refl"]),
                       "rays_per_pulse": total, "segments_per_pulse": seg_all / args.steps, "received_per_pulse": received_all / args.steps,
                       "primary_Mrays_per_s": total * args.steps / dt / 1e6, "return_cube": "complex128 [%d rx][%d pulses][%d bins], all-reduced once per interval" % (cube.shape[0], args.steps, n_bins), "sharding": "%d-pulse interval over %d ranks: whole pulses, left-over pulses in interleaved 4096-index tiles; one group-table all-gather + one cube all-reduce per interval" % (args.steps, world),
                       "pulses_in_flight": len(trs), "interval_tail_ms_rank0": acc["tail_ms"], "stage_ms_per_launch_rank0": {"scene_placement": ms_scene / launches, "trace": ms_launch, "order+finalise+aggregate": ms_post / launches}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_per_seg * seg_per_launch, "kernel": "k_trace", "bytes_per_segment": bytes_per_seg,
                         "nodes_per_segment": V, "tri_tests_per_segment": T, "shaded_per_segment": Hh,
                         "kernel_ms_avg": ms_launch, "segments_per_launch": seg_per_launch,
                         "hbm_GBps_measured": (traffic / (ms_launch * 1e-3) / 1e9) if (traffic and ms_launch > 0) else None,
                         "note": "achieved counts the ALGORITHMIC bytes of SURVEY 8d (288 B ray state + 128 B per BVH4 node visit + 80 B per triangle test + 96 B per shaded hit); the 20 MB scene is served from L1/L2 (traffic = measured HBM bytes per launch), so frac measures traversal rate against the agreed yardstick and can exceed 1 -- the kernel is latency/issue bound, not bandwidth bound (DESIGN.md section 5)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(spec)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    for t in trs:
        t.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
