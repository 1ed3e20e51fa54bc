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
