#!/usr/bin/env python3
"""Block / tile timeline of one k_trace launch (debug): RTS_TIMELINE=<file> makes the counting build dump, per block,
its start and end tick (100 MHz wall clock) and, per tile of 256 launch indices, the tile's duration.
   python tools/timeline.py [c3|c3narrow|c2|c4|c5]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "timeline.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["RTS_TIMELINE"] = out
from rts_amd import api, scenes  # noqa: E402
import rts_amd._lib
rts_amd._lib.require_built()        # a timed tool never builds, and never measures a stale library
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
spec = scenes.config2(rx_radius=200.0) if which == "c2" else scenes.config4() if which == "c4" else scenes.config5() if which == "c5" else scenes.config3()
if which == "c3narrow":
    spec["tx"] = dict(spec["tx"], span=(0.004, 0.004, 0.1))
tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], count_traversal=True)
tr.set_scene(spec["meshes"]); tr.set_receivers(spec["rx"]); tx = spec["tx"]
shard = int(os.environ.get("RTS_SHARD", "1"))                     # part RTS_SHARD_PART of RTS_SHARD interleaved parts (one GPU's share of a ray-sharded pulse)
il = (4096, shard, int(os.environ.get("RTS_SHARD_PART", "0"))) if shard > 1 else None
for _ in range(int(os.environ.get("RTS_TIMELINE_LAUNCHES", "3"))):
    st = tr.trace(tx["origin"], tx["span"], tx["dir"], spec["motion"], ray_first=0, ray_count=spec["W"] ** 3, interleave=il)
print("stats of the last launch:", {k: st[k] for k in ("rays", "segments", "coop_tiles", "ms_trace")})
d = np.fromfile(out, np.uint64)
grid, ntiles = int(d[0]), int(d[1]); blk = d[2:2 + 2 * grid].reshape(grid, 2).astype(np.int64); tile = d[2 + 2 * grid:2 + 2 * grid + ntiles].astype(np.float64) / 100.0   # us
tstart = d[2 + 2 * grid + ntiles:2 + 2 * grid + 2 * ntiles].astype(np.int64)
t0 = blk[:, 0].min(); start = (blk[:, 0] - t0) / 100.0; end = (blk[:, 1] - t0) / 100.0
print("launch %.3f ms (events), blocks %d wave tiles %d" % (st["ms_trace"], grid, ntiles))
print("block start us: pct 0/25/50/75/100 =", np.percentile(start, [0, 25, 50, 75, 100]).round(1))
print("block end   us: pct 0/25/50/75/100 =", np.percentile(end, [0, 25, 50, 75, 100]).round(1))
print("block duration us: mean %.1f  pct 50/90/99/100 =" % (end - start).mean(), np.percentile(end - start, [50, 90, 99, 100]).round(1))
print("tile duration us: mean %.1f  pct 50/90/99/99.9/100 =" % tile.mean(), np.percentile(tile, [50, 90, 99, 99.9, 100]).round(1), " sum %.1f ms" % (tile.sum() / 1e3))
late = np.argsort(end)[-8:]
print("last blocks to end:", [(int(b), round(float(start[b]), 1), round(float(end[b]), 1)) for b in late])
busy = np.zeros(int(end.max() / 50) + 2)
for s, e in zip(start, end):
    busy[int(s / 50):int(e / 50) + 1] += 1
print("resident blocks per 50 us bin:", busy.astype(int).tolist())
slow = np.argsort(tile)[-16:][::-1]
print("slowest tiles (tile, start us, duration us, end us):", [(int(t), round(float(tstart[t] - t0) / 100.0, 1), round(float(tile[t]), 1), round(float(tstart[t] - t0) / 100.0 + float(tile[t]), 1)) for t in slow])
srt = np.sort(tile)[::-1]
print("sorted tile durations us: top 1/4/16/64/256/1024 =", [round(float(srt[k]), 1) for k in (0, 3, 15, 63, 255, 1023) if k < len(srt)], " balanced bound %.1f us" % (tile.sum() / (grid * 4)))
tend = (tstart - t0) / 100.0 + tile
print("last tiles to end (tile, start, dur, end):", [(int(t), round(float(tstart[t] - t0) / 100.0, 1), round(float(tile[t]), 1), round(float(tend[t]), 1)) for t in np.argsort(tend)[-8:][::-1]])
hist, edges = np.histogram(tile, bins=[0, 5, 10, 20, 50, 100, 200, 400, 800, 1600, 1e9])
print("tile duration histogram (us):", list(zip(edges[:-1].astype(int).tolist(), hist.tolist())))
np.save(os.path.join(ROOT, "gpurun_out", "tile_us.npy"), tile)
