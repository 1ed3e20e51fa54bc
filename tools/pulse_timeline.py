#!/usr/bin/env python3
"""per-pulse view of a rocprofv3 --kernel-trace of the pipelined bench: for the n launches of the timed window, when the
placement started, when the trace kernel ran, when the post-processing chain started and ended, and on which queues
  python tools/pulse_timeline.py <dir> [n]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), r.get("Queue_Id", "?")) for r in rows])
tr = [e for e in ev if e[2] == "k_trace"]
best = min(range(0, max(1, len(tr) - n + 1)), key=lambda i: tr[i + n - 1][1] - tr[i][0])
t0 = tr[best][0]; t1 = tr[best + n - 1][1]
print("window of %d trace launches: %.3f ms" % (n, (t1 - t0) / 1e6))
names = ("k_place", "k_leaves", "k_tile_keys", "k_trace", "k_sum_counters", "k_recv_order_small", "k_expand", "k_finalise", "k_agg_order_small", "k_agg_finish_small")
for s, e, k, q in ev:
    if t0 - 300000 <= s <= t1 and k in names:
        print("%9.3f -> %9.3f  (%7.3f ms)  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, k))
