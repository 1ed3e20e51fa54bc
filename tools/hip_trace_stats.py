#!/usr/bin/env python3
"""per-API host durations of a rocprofv3 --hip-trace run (the *_hip_api_trace.csv of one process): count, mean, p50, p99, max in us, and the
same for the launches of the trace kernel alone -- what differs between a fast and a slow process
   tools/hip_trace_stats.py <dir>"""
import collections
import csv
import glob
import os
import sys
import numpy as np
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*hip_api_trace.csv"), recursive=True)):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Function"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    print(os.path.relpath(f, sys.argv[1]))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
        a = np.array(v)
        print("   %-28s n %6d  total %8.1f ms  mean %7.1f  p50 %7.1f  p99 %8.1f  max %9.1f us" % (k, len(a), a.sum() / 1e3, a.mean(), np.percentile(a, 50), np.percentile(a, 99), a.max()))
