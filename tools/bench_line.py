#!/usr/bin/env python3
"""one-line summary of a bench.py JSON line (the last line of the file)"""
import json
import sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]; c = j["config"]
        print(f.split("/")[-1], round(j["value"]), "Mrays/s", round(j["ms_per_step"], 4), "ms/pulse | serial", round(r["kernel_ms_serial"], 3), "hit", round(r["hit_fraction"], 3),
              "| frac", r.get("frac"), "| host", {k: round(v, 3) for k, v in c["host_ms_per_pulse_rank0"].items()}, "| tail", {k: round(v, 2) for k, v in c["interval_tail_ms_rank0"].items()}, "| setup_s", round(c["scene_setup_s"], 3))
    except Exception as e:          # noqa: BLE001
        print(f, "unreadable:", e)
