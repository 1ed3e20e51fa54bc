#!/usr/bin/env python3
"""every kernel of the last `ms` milliseconds of a rocprofv3 --kernel-trace CSV: start, end, duration, queue, name
   python tools/ktrace_window.py <dir> [ms]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
t1 = max(e for _, e, _, _ in ev); t0 = t1 - int(ms * 1e6)
first = min(s for s, _, _, _ in ev if s >= t0)
for s, e, k, q in ev:
    if s >= t0:
        print("%9.3f -> %9.3f  (%7.3f ms)  q%-3s %s" % ((s - first) / 1e6, (e - first) / 1e6, (e - s) / 1e6, q, k))
