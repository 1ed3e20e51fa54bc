"""Ray-sharded multi-GPU pulse: one process per GPU, contiguous launch-index ranges per rank.

The reference is single-GPU (no NCCL/MPI anywhere in its 9 files).  Rays are independent
(ray_tracer.cu:227-253 writes only the launch index's own slots), so the only exchange is the
aggregation: every rank reduces its own received rays to a (receiver, path) group table, the
tables are all-gathered (RCCL over xGMI with backend "nccl"; gloo in the CPU tests) and merged
on every rank by librts_amd's host routine rts_merge_groups; the responses the reference would
emit follow from the merged table (rts_groups_to_responses).  The received-list indices that
pathMatch refers to (aggregation.cu:68-69) are made global by offsetting each rank's list with
the received counts of the lower ranks, which is exactly the order of the reference's host scan
(ray_tracer.cpp:1190) because the ranges are contiguous and ascending.
"""
import numpy as np

from . import api
from ._lib import GROUP_DTYPE


def _device_for(dist, torch):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def shard_range(total, rank, world):
    """contiguous launch-index range of `rank`: [first, first + count)"""
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi - lo


def exchange_received_base(n_local, dist, torch):
    """index of this rank's first received ray in the global received list"""
    dev = _device_for(dist, torch)
    world = dist.get_world_size(); rank = dist.get_rank()
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    allc = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, mine)
    counts = [int(c.item()) for c in allc]
    return sum(counts[:rank]), counts


def gather_groups(groups, dist, torch):
    """all-gather variable-length group tables; returns the concatenation in rank order"""
    dev = _device_for(dist, torch)
    world = dist.get_world_size()
    g = np.ascontiguousarray(groups, GROUP_DTYPE)
    n = torch.tensor([len(g)], dtype=torch.int64, device=dev)
    ns = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ns, n)
    ns = [int(x.item()) for x in ns]
    cap = max(max(ns), 1)
    buf = np.zeros(cap, GROUP_DTYPE); buf[:len(g)] = g
    mine = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy()).to(dev)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine)
    parts = [o.cpu().numpy().view(GROUP_DTYPE)[:k] for o, k in zip(outs, ns)]
    return np.concatenate(parts) if parts else np.zeros(0, GROUP_DTYPE)


def merge_and_respond(all_groups, depth):
    merged = api.merge_groups(all_groups, depth)
    return api.groups_to_responses(merged), merged


def aggregate_sharded(tracer, cspeed, carrier, dist, torch):
    """aggregation step of one pulse on every rank; returns (responses, merged group table)"""
    base, _ = exchange_received_base(tracer.received_count(), dist, torch)
    local = tracer.aggregate(cspeed, carrier, base)
    allg = gather_groups(local, dist, torch)
    return merge_and_respond(allg, tracer.depth)


# ------------------------------------------------------------------------------- CPI-level (pulse x ray) sharding
# A coherent processing interval is K independent pulses (ray_tracer.cpp:843) of W^3 launch indices each.
# Whole pulses are dealt out first (K // N per rank: no replicated scene placement, no imbalance); each of the K % N
# left-over pulses is shared by a group of consecutive ranks that split its launch indices into INTERLEAVED
# tiles (rays that hit cluster in launch-index space, so contiguous sub-ranges would leave most of the group
# idle).  K = 1 is plain interleaved ray sharding of one pulse over all ranks.  The per-(receiver, path) group
# tables of all parts are exchanged ONCE per interval -- they are a few hundred bytes, so the exchange is
# latency bound and batching it is what matters on xGMI.
IL_TILE = 4096                # == RTS_PLAN_TILE


def _plan(total_rays, n_pulses, rank, world, mode, min_items):
    """the plan lives in the library (rts_plan_cpi, also used by the C++ adapter); items as (pulse, first, count, interleave)"""
    import ctypes as C
    from . import _lib as L
    n = C.c_uint32(0)
    L.check(L.lib().rts_plan_cpi(total_rays, n_pulses, rank, world, mode, min_items, IL_TILE, None, 0, C.byref(n)))
    arr = (L.RtsPlanItem * max(n.value, 1))()
    L.check(L.lib().rts_plan_cpi(total_rays, n_pulses, rank, world, mode, min_items, IL_TILE, arr, n.value, C.byref(n)))
    return [(int(it.pulse), int(it.ray_first), int(it.ray_count),
             (int(it.interleave_tile), int(it.interleave_parts), int(it.interleave_part)) if it.interleave_parts > 1 else None) for it in arr[:n.value]]


def plan_cpi(total_rays, n_pulses, rank, world, min_items=0):
    """[(pulse, ray_first, ray_count, interleave)] owned by `rank`; interleave = None or (tile, parts, part): whole pulses
    first, left-over pulses shared by groups of ranks in interleaved tiles"""
    return _plan(total_rays, n_pulses, rank, world, 0, min_items)


def plan_rays(total_rays, n_pulses, rank, world, min_items=0):
    """ray sharding: EVERY pulse of the interval is split over all ranks in interleaved tiles ("ray-sharding is
    the one to report"); the work per rank does not depend on how n_pulses divides by the number of ranks"""
    return _plan(total_rays, n_pulses, rank, world, 1, min_items)


def plan_whole(total_rays, n_pulses, rank, world, min_items=0):
    """whole pulses only (rts_plan_cpi, RTS_SHARD_PULSES_WHOLE): contiguous runs, the first n_pulses % world ranks one pulse more; with fewer
    pulses than ranks the left-over rule of plan_cpi.  No launch of a new shape: on a short interval of small pulses that beats splitting the
    left-over pulses (profiles/r05d_as_rank_pulses_*.log)"""
    return _plan(total_rays, n_pulses, rank, world, 2, min_items)


def refine_plan(plan, min_items):
    """split items until the rank owns at least `min_items` of them, so that it can keep that many pulses (or pulse
    parts) in flight: part p of P (tile T) splits into parts p and p + P of 2P -- every other one of its tiles.  (Same
    rule as rts_plan_cpi's min_items; kept for plans assembled by hand.)"""
    plan = list(plan)
    while plan and len(plan) < min_items:
        k, first, count, il = plan.pop(0)                          # oldest first keeps the pulse order of the plan
        tile, parts, part = il if il is not None else (IL_TILE, 1, 0)
        if part_ray_count(count, (tile, 2 * parts, part + parts)) == 0:
            plan.insert(0, (k, first, count, il)); break           # nothing left to split off
        plan += [(k, first, count, (tile, 2 * parts, part)), (k, first, count, (tile, 2 * parts, part + parts))]
    return plan


def part_ray_count(total_rays, interleave):
    if interleave is None:
        return total_rays
    tile, parts, part = interleave
    stride = tile * parts
    full, rem = divmod(total_rays, stride)
    return full * tile + max(0, min(rem - part * tile, tile))


# ------------------------------------------------------------------------------- ray sharding dealt by last-seen cost
# The static interleave gives every rank the same NUMBER of tiles; what a tile costs differs by four orders of magnitude (a dead
# tile 5 us of one wave, a grazing tile of BASELINE configs[3] milliseconds), so the ranks' parts of a pulse differ by 2 x.  Once per
# interval the ranks exchange what each tile cost the rank that traced it (one uint32 per 64 launch indices: 6 MB for 100 M indices),
# every rank adopts the merged table as its history -- tile order AND cooperative-kernel head tiles for tiles it never traced --
# and computes the same longest-first deal from it (rts_deal_tiles); the next interval's launches use interleave = (tile,
# INTERLEAVE_LIST, 0) after Tracer.set_tile_list.  The data path has no collective: results are keyed by global buffer rows as before.
def exchange_tile_records(local_tables, dist, torch):
    """local_tables: Tracer.tile_records_get() of every tracer of this rank (they traced the same part, pulse after pulse: the
    element-wise maximum stands for the rank) -> the table of all ranks (their parts are disjoint: a sum)"""
    mine = np.maximum.reduce([np.ascontiguousarray(t, np.uint32) for t in local_tables]) if len(local_tables) > 1 else np.ascontiguousarray(local_tables[0], np.uint32)
    if dist is None or dist.get_world_size() == 1:
        return mine
    t = torch.from_numpy(mine.astype(np.int64)).to(_device_for(dist, torch))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint32)


def dealt_tiles(records, total_rays, rank, world, tile=None):
    """(tile, ascending plan-tile ids of `rank`, cost of every rank's part) from a merged record table"""
    tile = tile or IL_TILE
    part_of, cost = api.deal_tiles(records, total_rays, tile, world)
    return tile, np.flatnonzero(part_of == rank).astype(np.uint32), cost


def list_ray_count(total_rays, tile, ids):
    return int(sum(min(tile, total_rays - int(i) * tile) for i in ids))


def exchange_parts(parts, dist, torch):
    """parts: [dict(pulse, groups)] of this rank -> the same list for ALL ranks.  Group tables must be keyed by
    global buffer rows (rts_aggregate(..., RTS_BASE_USE_ROWS)) so that they merge with a plain min.
    One all-gather of sizes + one all-gather of payload (meta int64 rows followed by the group records)."""
    if dist is None:                                         # (one rank: responses formed while the pulses were still being traced -- bench.py's collect -- travel with their tables)
        return [dict(pulse=int(p["pulse"]), groups=np.ascontiguousarray(p["groups"], GROUP_DTYPE), responses=p.get("responses")) for p in parts]
    meta = np.array([[p["pulse"], len(p["groups"])] for p in parts], np.int64).reshape(-1, 2)
    groups = np.concatenate([np.ascontiguousarray(p["groups"], GROUP_DTYPE) for p in parts]) if parts else np.zeros(0, GROUP_DTYPE)
    payload = np.concatenate([meta.view(np.uint8).reshape(-1), groups.view(np.uint8).reshape(-1)])
    if dist is None:
        blobs, sizes = [payload], [(len(meta), len(groups))]
    else:
        dev = _device_for(dist, torch); world = dist.get_world_size()
        sz = torch.tensor([len(meta), len(groups)], dtype=torch.int64, device=dev)
        szs = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(szs, sz)
        sizes = [(int(x[0]), int(x[1])) for x in szs]
        cap = max(max(m * 16 + g * GROUP_DTYPE.itemsize for m, g in sizes), 8)
        buf = np.zeros(cap, np.uint8); buf[:len(payload)] = payload
        mine = torch.from_numpy(buf).to(dev)
        outs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(outs, mine)
        blobs = [o.cpu().numpy() for o in outs]
    allp = []
    for blob, (m, g) in zip(blobs, sizes):
        mt = blob[:m * 16].view(np.int64).reshape(m, 2)
        gr = blob[m * 16:m * 16 + g * GROUP_DTYPE.itemsize].view(GROUP_DTYPE)
        off = 0
        for row in mt:
            allp.append(dict(pulse=int(row[0]), groups=gr[off:off + int(row[1])].copy()))
            off += int(row[1])
    return allp


def merge_cpi(all_parts, depth):
    """all parts of all ranks -> {pulse: (responses, merged groups)}; response.ray = global buffer row of the
    representative ray (same order as the reference's received-list index)"""
    by_pulse = {}; ready = {}
    for p in all_parts:
        by_pulse.setdefault(p["pulse"], []).append(p["groups"])
        if p.get("responses") is not None:
            ready[p["pulse"]] = p["responses"]
    out = {}
    for k, tabs in by_pulse.items():
        if len(tabs) == 1:                                   # a pulse traced whole by one rank: its table is already merged
            out[k] = (ready[k] if k in ready else api.groups_to_responses(tabs[0]), tabs[0])
        else:
            out[k] = merge_and_respond(np.concatenate(tabs), depth)
    return out
