#!/usr/bin/env python3
"""summary of tests/adapter/adapter_bench.cpp's JSON line"""
import json
import sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], {k: v for k, v in j.items() if k not in ("builders", "what")})
        for b, v in j["builders"].items():
            print(" ", b, {k: x for k, x in v.items() if k != "intervals"})
            for iv in v["intervals"]:
                print("    %.4f ms/pulse gaps %s setup %.3f laps %s" % (iv["ms_per_pulse"], {k: round(x, 3) for k, x in iv["gap_ms"].items()}, iv["setup_s"], {k: round(x, 4) for k, x in iv["host_lap_ms_per_pulse"].items()}))
    except Exception as e:          # noqa: BLE001
        print(f, "unreadable:", e)
