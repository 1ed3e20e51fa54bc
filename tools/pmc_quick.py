#!/usr/bin/env python3
"""mean counter values per trace-kernel dispatch (ordinary / cooperative) of a few rocprofv3 --pmc passes:
   tools/pmc_quick.py <dir under gpurun_out/pmc_> <pass> ..."""
import collections
import csv
import glob
import os
import sys
root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in sys.argv[2:]:
    fs = glob.glob(os.path.join(root, "gpurun_out", "pmc_" + sys.argv[1], p, "**", "*_counter_collection.csv"), recursive=True)
    if not fs:
        print(p, "no file"); continue
    d = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        targs = r["Kernel_Name"].split("<", 1)[-1].split(",")
        k = (int(r["Dispatch_Id"]), "coop" if len(targs) > 3 and targs[3].strip() == "true" else "ord")
        e = d.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["_ms"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
    agg = collections.defaultdict(list)
    for (disp, kind), v in list(d.items())[2:]:          # (the handle's first launches: no order yet)
        for c, x in v.items():
            agg[(kind, c)].append(x)
    print(sys.argv[1], p, {"%s %s" % k: round(sum(v) / len(v), 3) for k, v in agg.items()})
