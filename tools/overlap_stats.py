#!/usr/bin/env python3
"""How the trace kernels of a pipelined bench overlap: reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints, over the
span of the LAST `n` k_trace launches, the time with 0 / 1 / 2 / 3+ trace kernels resident, the mean launch duration and the
launch-to-launch period -- the difference between `ms_per_step` and the kernel's serial time is read off here.
  python tools/overlap_stats.py <dir or csv> [n]"""
import csv, glob, os, sys
path = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
tr = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if r["Kernel_Name"].startswith("void k_trace") or r["Kernel_Name"].startswith("k_trace")])
# the n consecutive launches with the shortest span: the timed region of the bench (the launches around it -- warm-up, the serial
# and dense-control measurements -- are spread out)
best = min(range(0, max(1, len(tr) - n + 1)), key=lambda i: max(e for _, e in tr[i:i + n]) - tr[i][0])
tr = tr[best:best + n]
t0, t1 = tr[0][0], max(e for _, e in tr)
ev = sorted([(s, 1) for s, _ in tr] + [(e, -1) for _, e in tr])
lvl = 0; last = t0; hist = {}
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - last); last = t; lvl += d
span = t1 - t0
others = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if not r["Kernel_Name"].startswith(("void k_trace", "k_trace")) and t0 <= int(r["Start_Timestamp"]) <= t1]
print("%d trace launches over %.3f ms: period %.3f ms, mean duration %.3f ms" % (len(tr), span / 1e6, (tr[-1][0] - tr[0][0]) / 1e6 / (len(tr) - 1), sum(e - s for s, e in tr) / 1e6 / len(tr)))
print("trace kernels resident: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
print("other kernels in the span: %d launches, %.3f ms summed (%.3f ms per trace launch)" % (len(others), sum(e - s for s, e, _ in others) / 1e6, sum(e - s for s, e, _ in others) / 1e6 / len(tr)))
agg = {}
for s, e, k in others:
    k = k.split("(")[0][-60:]; a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-60s %4d  %.3f ms" % (k, c, d / 1e6))
