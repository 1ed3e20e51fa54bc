#!/usr/bin/env python3
"""One GPU's share of a ray-sharded pulse, every part in turn, two ways: the static interleave (RtsPulse.interleave_*) and the
longest-first deal from the tile cost records of the previous interval (rts_tile_records_get / rts_deal_tiles / rts_set_tile_list).
Every part starts from the same history -- the records of whole pulses, as every rank holds them after the interval's exchange.
   python tools/deal_bench.py [c4|c3] [parts=8] [tile=4096] [launches=7]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rts_amd import api, scenes  # noqa: E402
import rts_amd._lib
rts_amd._lib.require_built()

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 7
spec = scenes.config4() if which == "c4" else scenes.config3(rx_radius=50.0)
tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"])
tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"]); tx = spec["tx"]; n = spec["W"] ** 3


def motion(k):
    return spec["motion_fn"](k) if "motion_fn" in spec else [dict(position=tuple(np.add(m["position"], (0.2 * k, 0.02 * k, 0.0))), velocity=m["velocity"]) for m in spec["motion"]]


whole = []
for k in range(6):
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], motion(k), ray_first=0, ray_count=n)
    whole.append(st["ms_trace"])
table = tr.tile_records_get()
print("%s: whole pulse %s ms per launch; %d of %d wave tiles have a record, %d flagged LONG WALKS" % (spec["name"], " ".join("%.3f" % x for x in whole), np.count_nonzero(table), table.shape[0], int((table >> 31).sum())))
part_of, cost = api.deal_tiles(table, n, tile, parts)
c = np.maximum(np.add.reduceat((table & 0x3fffffff).astype(np.uint64), np.arange(0, table.shape[0], tile // 64)), 1)
il_cost = np.array([int(c[r::parts].sum()) for r in range(parts)], np.float64)
print("cost of the parts by the records (units of the mean): interleaved %s | dealt %s" % (" ".join("%.2f" % x for x in il_cost / il_cost.mean()), " ".join("%.3f" % x for x in cost / cost.mean())))
res = {}
for mode in ("interleaved", "dealt"):
    for r in range(parts):
        tr.tile_records_set(table)
        if mode == "dealt":
            tr.set_tile_list(tile, np.flatnonzero(part_of == r).astype(np.uint32)); il = (tile, api.INTERLEAVE_LIST, 0)
        else:
            tr.set_tile_list(0, np.zeros(0, np.uint32)); il = (tile, parts, r)
        ms = []
        for k in range(launches):
            st = tr.trace(tx["origin"], tx["span"], tx["dir"], motion(6 + k), ray_first=0, ray_count=n, interleave=il)
            ms.append(st["ms_trace"])
        res[(mode, r)] = ms
        print("%-11s part %d/%d: %d rays, %d segs, coop tiles %d | ms per launch: %s" % (mode, r, parts, st["rays"], st["segments"], st["coop_tiles"], " ".join("%.3f" % x for x in ms)), flush=True)
for mode in ("interleaved", "dealt"):
    settled = np.array([np.median(res[(mode, r)][2:]) for r in range(parts)])
    print("%-11s settled ms per part: %s | worst %.3f mean %.3f | whole %.3f -> projected efficiency at N = %d: %.2f (worst part), %.2f (mean)" %
          (mode, " ".join("%.2f" % x for x in settled), settled.max(), settled.mean(), np.median(whole[2:]), parts, np.median(whole[2:]) / (parts * settled.max()), np.median(whole[2:]) / (parts * settled.mean())))
