#!/usr/bin/env python3
"""lists the trace kernels of a rocprofv3 --kernel-trace CSV: start, duration  (python tools/ktrace_list.py <dir>)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "k_trace" in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%-52s start %9.3f ms  dur %8.3f ms" % (r["Kernel_Name"][:52], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
