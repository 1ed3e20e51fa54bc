#!/usr/bin/env python3
"""bench.py -- RTS hot path on MI355X: Mrays/s and ms/pulse on the BASELINE.json metric config.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[2] -- aircraft-like 100 000-triangle mesh,
1 Tx / 4 Rx, W = 216 (10 077 696 launch indices per pulse), maxRefl = 6; the configuration the
metric ("Mrays/s ... 100k-tri scene") is quoted on.  Synthetic mesh, isotropic antennas, RCS 1.

A step is ONE PULSE end to end: target placement for that pulse (the target moves every
pulse, so the LBVH is rebuilt on the device inside the timed region, as the reference rebuilds
its acceleration structure every pulse), trace of this rank's share of the W^3 launch
indices, ordering + expansion of the received rays, finalisation, group-by aggregation and --
for N > 1 -- the all-gather of the per-(receiver, path) group tables over RCCL and their
merge into the pulse's responses.  A "ray" in Mrays/s is one traced segment (one rtTrace of
the reference: primary or bounce).  Scaling is strong: the pulse's W^3 launch indices are
split into N contiguous ranges.
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
    out = []
    for m in spec["motion"]:
        v = np.asarray(m["velocity"], np.float64); p0 = np.asarray(m["position"], np.float64)
        out.append(dict(position=tuple(p0 + v * (k * PRI)), velocity=tuple(v)))
    return out


def shard(total, rank, world):
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi - lo


def cpu_baseline(spec, seconds_target=15.0):
    """The CPU restatement (oracle, BVH mode, all host threads) timed on a strided sample of
    the same pulse.  kind = "port": the reference has no CPU path and cannot be built here."""
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    threads = min(os.cpu_count() or 1, 64)
    sc = H.oracle_scene(O, spec, pulse_motion(spec, 0))
    tx = spec["tx"]; W = spec["W"]; total = W ** 3
    n = 20000
    t0 = time.time()
    r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0,
                 ray_stride=total // n, n_rays=n, use_bvh=True, threads=threads, debug=False)     # includes the BVH build
    t_first = time.time() - t0
    t0 = time.time()
    r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0,
                 ray_stride=total // n, n_rays=n, use_bvh=True, threads=threads, debug=False)
    dt = time.time() - t0
    rate = r["counters"]["segments"] / dt
    n2 = int(min(total, max(n, rate and n * seconds_target / max(dt, 1e-3))))
    stride = max(total // n2, 1)
    n2 = min(n2, total // stride)
    t0 = time.time()
    r = sc.trace(tx["origin"], tx["span"], tx["dir"], W, spec["max_refl"], 0, spec["smooth"], ray_first=0,
                 ray_stride=stride, n_rays=n2, use_bvh=True, threads=threads, debug=False)
    dt = time.time() - t0
    return dict(value=r["counters"]["segments"] / dt / 1e6, unit="Mrays/s", cores=threads, kind="port",
                sample="%d of %d launch indices (stride %d) of one pulse, %d segments in %.1f s, oracle BVH mode" %
                       (n2, total, stride, r["counters"]["segments"], dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=216, help="W (launch indices per pulse = W^3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="c3", choices=["c2", "c3"])
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the RTS hot path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import rts_amd
    from rts_amd import api, scenes, multigpu
    rts_amd.build()
    spec = scenes.config3(W=args.width) if args.config == "c3" else scenes.config2(W=args.width if args.width != 216 else 100)
    W = spec["W"]; total = W ** 3
    first, count = shard(total, rank, world)
    tx = spec["tx"]; wl = spec["c"] / spec["carrier"]

    tr = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank)
    tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"])

    def step(k):
        st = tr.trace(tx["origin"], tx["span"], tx["dir"], pulse_motion(spec, k), ray_first=first, ray_count=count)
        tr.finalise_uniform(None, wl, 1.0, 1.0, spec["carrier"], spec["c"])
        if world > 1:
            resp, st2 = multigpu.aggregate_sharded(tr, spec["c"], spec["carrier"], dist, torch)
        else:
            groups = tr.aggregate(spec["c"], spec["carrier"], 0)
            resp = api.groups_to_responses(groups)
        return tr.stats(), resp

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync()
    seg = 0; ms_trace = 0.0; ms_scene = 0.0; ms_post = 0.0; shaded = 0; received = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        st, resp = step(args.warmup + k)
        seg += st["segments"]; ms_trace += st["ms_trace"]; ms_scene += st["ms_scene"]; ms_post += st["ms_compact"] + st["ms_aggregate"]
        shaded += st["shaded"]; received += st["received"]
    sync()
    dt = time.perf_counter() - t0

    # whole-job aggregates: max time over ranks, sum of segments
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
        ss = torch.tensor([seg, shaded, received], dtype=torch.float64, device="cuda"); dist.all_reduce(ss, op=dist.ReduceOp.SUM)
        seg_all, shaded_all, received_all = [int(x) for x in ss.tolist()]
    else:
        seg_all, shaded_all, received_all = seg, shaded, received

    if rank == 0:
        # traversal counts for the roofline accounting: one untimed pulse of the counting build
        trc = api.Tracer(W, spec["max_refl"], 0, spec["smooth"], device=local_rank, count_traversal=True)
        trc.set_scene(spec["meshes"]); trc.set_receivers(spec["rx"])
        sc = trc.trace(tx["origin"], tx["span"], tx["dir"], pulse_motion(spec, args.warmup), ray_first=first, ray_count=count)
        trc.close()
        V = sc["node_visits"] / max(sc["segments"], 1); T = sc["tri_tests"] / max(sc["segments"], 1); Hh = sc["shaded"] / max(sc["segments"], 1)
        bytes_per_seg = 288.0 + 64.0 * V + 72.0 * T + 96.0 * Hh           # SURVEY.md section 8(d), figure (B)
        seg_per_launch = seg / max(args.steps, 1)                          # rank 0's launches
        ms_launch = ms_trace / max(args.steps, 1)
        achieved = bytes_per_seg * seg_per_launch / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        out = {
            "metric": "Mrays/s (primary+bounces) & ms/pulse, 100k-tri scene",
            "value": seg_all / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: %s, 1 Tx / %d Rx, W=%d (%d launch indices/pulse), maxRefl=%d, target moves every pulse (LBVH rebuilt per pulse)"
                                   % (spec["name"], len(spec["rx"]), W, total, spec["max_refl"]),
                       "rays_per_pulse": total, "segments_per_pulse": seg_all / args.steps, "received_per_pulse": received_all / args.steps,
                       "primary_Mrays_per_s": total * args.steps / dt / 1e6, "sharding": "contiguous launch-index ranges x%d" % world,
                       "stage_ms_rank0": {"scene+lbvh": ms_scene / args.steps, "trace": ms_launch, "order+finalise+aggregate": ms_post / args.steps}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_trace", "bytes_per_segment": bytes_per_seg,
                         "nodes_per_segment": V, "tri_tests_per_segment": T, "shaded_per_segment": Hh,
                         "kernel_ms_avg": ms_launch, "segments_per_launch": seg_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(spec)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    tr.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
