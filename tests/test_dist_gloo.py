"""The N > 1 path on the CPU: two gloo ranks shard a pulse's launch indices, each rank reduces ITS
received rays to a (receiver, path) group table, the tables travel through
rts_amd.multigpu (all-gather) and are merged by librts_amd's host routines.  The merged
responses must equal the literal single-process aggregation of the whole pulse.

On the CPU there is no device to trace with, so each rank's received set is produced by the
oracle for the rank's launch-index range (the oracle stands in for the device here, as the
checker's input generator); what is under test is the product's sharding arithmetic, the
exchange and the merge."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
C0 = 299792458.0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rts_amd import scenes, multigpu
    import helpers as H
    from test_host_logic import numpy_group_table
    spec = scenes.config_multi(W=16)
    total = spec["W"] ** 3
    first, count = multigpu.shard_range(total, rank, world)
    o = H.oracle_trace(O, spec, ray_first=first, ray_stride=1, n_rays=count)
    wl = spec["c"] / spec["carrier"]
    rx, rxi, slots = O.filter_finalise(o["results"], o["path"], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    base, counts = multigpu.exchange_received_base(len(rx), dist, torch)
    local = numpy_group_table(rx, rxi, spec["c"], spec["carrier"], base=base)
    allg = multigpu.gather_groups(local, dist, torch)
    resp, merged = multigpu.merge_and_respond(allg, spec["max_refl"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), resp=resp, merged=merged, counts=np.array(counts), base=base, first=first, count=count)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_sharded_pulse(tmp_path, world, oracle):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    from rts_amd import scenes
    import helpers as H
    spec = scenes.config_multi(W=16)
    o = H.oracle_trace(oracle, spec)
    wl = spec["c"] / spec["carrier"]
    rx, rxi, slots = oracle.filter_finalise(o["results"], o["path"], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], spec["W"] ** 3)
    uniq = oracle.unique_paths(lit["pathMatch"])
    outs = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert sum(int(x["count"]) for x in outs) == spec["W"] ** 3
    assert [int(x["first"]) for x in outs] == sorted(int(x["first"]) for x in outs)
    assert int(outs[0]["counts"].sum()) == len(rx)
    for x in outs:                                            # every rank ends with the same, complete answer
        resp = x["resp"]
        assert np.array_equal(resp["ray"].astype(np.int64), uniq.astype(np.int64))
        np.testing.assert_allclose(resp["power"], lit["results"]["power"][uniq], rtol=1e-12)
        np.testing.assert_allclose(resp["delay"], lit["delay"][uniq], rtol=1e-12)
        np.testing.assert_allclose(resp["doppler"], lit["results"]["doppler"][uniq], rtol=1e-12, atol=1e-12)
        assert int(x["merged"]["n"].sum()) == len(rx)


def _tile_indices(total, interleave):
    if interleave is None:
        return np.arange(total, dtype=np.int64)
    tile, parts, part = interleave
    idx = np.arange(total, dtype=np.int64)
    return idx[(idx // tile) % parts == part]


def _cpi_worker(rank, world, port, out_dir, n_pulses, shard="pulses"):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rts_amd import scenes, multigpu
    import helpers as H
    from test_host_logic import numpy_group_table
    multigpu.IL_TILE = 64                                       # small tiles so that the 1728-index test pulse really interleaves
    spec = scenes.config_multi(W=12)
    total = spec["W"] ** 3
    wl = spec["c"] / spec["carrier"]
    parts = []
    for (k, first, count, il) in (multigpu.plan_rays if shard == "rays" else multigpu.plan_cpi)(total, n_pulses, rank, world):
        mo = [dict(position=tuple(np.add(m["position"], (0.3 * k, 0.0, 0.1 * k))), velocity=m["velocity"]) for m in spec["motion"]]
        o = H.oracle_trace(O, spec, motion=mo)                  # whole pulse, then keep this part's launch indices (stand-in for the device)
        mine = _tile_indices(total, il)
        assert len(mine) == multigpu.part_ray_count(total, il)
        rx, rxi, slots = O.filter_finalise(o["results"][mine], o["path"][mine], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
        g = numpy_group_table(rx, rxi, spec["c"], spec["carrier"], base=0)
        rows = mine[slots.astype(np.int64)]                     # global buffer rows of this part's received rays
        g["min_ray"] = rows[g["min_ray"].astype(np.int64)]      # row-keyed, as rts_aggregate(..., RTS_BASE_USE_ROWS) produces
        parts.append(dict(pulse=k, groups=g))
    allp = multigpu.exchange_parts(parts, dist, torch)
    merged = multigpu.merge_cpi(allp, spec["max_refl"])
    np.savez(os.path.join(out_dir, "cpi_rank%d.npz" % rank), pulses=np.array(sorted(merged)), **{"resp%d" % k: merged[k][0] for k in merged})
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pulses,shard", [(2, 3, "pulses"), (3, 2, "pulses"), (2, 1, "pulses"), (3, 4, "pulses"), (2, 2, "rays"), (3, 3, "rays")])
def test_cpi_sharding(tmp_path, world, n_pulses, shard, oracle):
    """pulse x ray sharding of a CPI: whole pulses per rank, left-over pulses in interleaved tiles within rank groups -- or
    (bench.py --shard rays) EVERY pulse split over all ranks in interleaved tiles --, ONE exchange; per-pulse responses
    identical to the literal single-process pipeline"""
    mp.spawn(_cpi_worker, args=(world, _free_port(), str(tmp_path), n_pulses, shard), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    from rts_amd import scenes
    import helpers as H
    spec = scenes.config_multi(W=12)
    wl = spec["c"] / spec["carrier"]
    outs = [np.load(os.path.join(str(tmp_path), "cpi_rank%d.npz" % r)) for r in range(world)]
    for k in range(n_pulses):
        mo = [dict(position=tuple(np.add(m["position"], (0.3 * k, 0.0, 0.1 * k))), velocity=m["velocity"]) for m in spec["motion"]]
        o = H.oracle_trace(oracle, spec, motion=mo)
        rx, rxi, slots = oracle.filter_finalise(o["results"], o["path"], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
        lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], spec["W"] ** 3)
        uniq = oracle.unique_paths(lit["pathMatch"])
        for x in outs:
            assert list(x["pulses"]) == list(range(n_pulses))
            resp = x["resp%d" % k]
            assert np.array_equal(resp["ray"].astype(np.int64), slots[uniq].astype(np.int64))     # representative ray, as a buffer row
            np.testing.assert_allclose(resp["power"], lit["results"]["power"][uniq], rtol=1e-12)
            np.testing.assert_allclose(resp["delay"], lit["delay"][uniq], rtol=1e-12)


def _deal_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from rts_amd import scenes, multigpu
    import helpers as H
    from test_host_logic import numpy_group_table
    spec = scenes.config_multi(W=12)
    total = spec["W"] ** 3; tile = 64; n_rec = (total + 63) // 64
    wl = spec["c"] / spec["carrier"]
    o = H.oracle_trace(O, spec)                                  # whole pulse; a rank keeps ITS launch indices (stand-in for the device)
    # interval 0: the static interleave.  What a tile "cost" this rank: its shaded segments (a stand-in for the device's clock)
    mine0 = _tile_indices(total, (tile, world, rank))
    seg = np.zeros(n_rec, np.uint32)
    np.add.at(seg, mine0 // 64, (1 + o["results"]["reflDepth"][mine0]).astype(np.uint32))
    table = multigpu.exchange_tile_records([seg], dist, torch)
    t, ids, cost = multigpu.dealt_tiles(table, total, rank, world, tile)
    assert multigpu.list_ray_count(total, t, ids) > 0
    # interval 1: the dealt tiles
    mine = np.concatenate([np.arange(int(i) * t, min((int(i) + 1) * t, total), dtype=np.int64) for i in ids])
    rx, rxi, slots = O.filter_finalise(o["results"][mine], o["path"][mine], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    g = numpy_group_table(rx, rxi, spec["c"], spec["carrier"], base=0)
    g["min_ray"] = mine[slots.astype(np.int64)][g["min_ray"].astype(np.int64)]
    allp = multigpu.exchange_parts([dict(pulse=0, groups=g)], dist, torch)
    merged = multigpu.merge_cpi(allp, spec["max_refl"])
    np.savez(os.path.join(out_dir, "deal_rank%d.npz" % rank), table=table, ids=ids, cost=cost, resp=merged[0][0], seg=seg)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ray_sharding_dealt_by_cost(tmp_path, world, oracle):
    """one exchange of the tile cost table per interval (all-reduce), the same longest-first deal computed on every rank, the
    next interval's pulse traced as dealt tile lists: the ranks' lists partition the lattice, their costs are level, and the
    merged responses are the literal single-process pipeline's"""
    mp.spawn(_deal_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    from rts_amd import scenes
    import helpers as H
    spec = scenes.config_multi(W=12); total = spec["W"] ** 3
    wl = spec["c"] / spec["carrier"]
    outs = [np.load(os.path.join(str(tmp_path), "deal_rank%d.npz" % r)) for r in range(world)]
    for x in outs[1:]:
        assert np.array_equal(x["table"], outs[0]["table"]) and np.array_equal(x["cost"], outs[0]["cost"])
    assert np.array_equal(outs[0]["table"], np.sum([x["seg"] for x in outs], axis=0))          # disjoint parts: the sum IS the table
    ids = np.concatenate([x["ids"] for x in outs])
    assert np.array_equal(np.sort(ids), np.arange((total + 63) // 64))                         # a partition of the plan tiles
    cost = outs[0]["cost"].astype(np.int64)
    assert cost.max() - cost.min() <= int(outs[0]["table"].max())
    o = H.oracle_trace(oracle, spec)
    rx, rxi, slots = oracle.filter_finalise(o["results"], o["path"], [1.0] * 3, wl, 1.0, 1.0, spec["carrier"], spec["c"])
    lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], total)
    uniq = oracle.unique_paths(lit["pathMatch"])
    for x in outs:
        assert np.array_equal(x["resp"]["ray"].astype(np.int64), slots[uniq].astype(np.int64))
        np.testing.assert_allclose(x["resp"]["power"], lit["results"]["power"][uniq], rtol=1e-12)
        np.testing.assert_allclose(x["resp"]["delay"], lit["delay"][uniq], rtol=1e-12)


def test_plan_cpi_covers_exactly():
    from rts_amd import multigpu
    for total in (1000, 216 ** 3):
        for K, N in [(1, 8), (8, 8), (5, 3), (20, 8), (3, 1), (1, 1), (5, 8), (9, 4)]:
            cover = {}
            for r in range(N):
                for (k, first, count, il) in multigpu.plan_cpi(total, K, r, N):
                    assert first == 0 and count == total
                    cover.setdefault(k, []).append(il)
            assert sorted(cover) == list(range(K))
            for k, ils in cover.items():
                if ils == [None]:
                    continue
                parts = ils[0][1]
                assert sorted(il[2] for il in ils) == list(range(parts)) and all(il[1] == parts for il in ils)
                assert sum(multigpu.part_ray_count(total, il) for il in ils) == total


def test_shard_ranges_cover_exactly():
    from rts_amd import multigpu
    for total in (1, 7, 10648, 216 ** 3):
        for world in (1, 2, 3, 8):
            r = [multigpu.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and sum(c for _, c in r) == total
            assert all(r[k][0] + r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(c for _, c in r) - min(c for _, c in r) <= 1


def test_refine_plan_partitions_the_same_indices():
    """refine_plan (pulse parts kept in flight on linked handles) only re-partitions what the rank already owned"""
    from rts_amd import multigpu

    def indices(total, il):
        idx = np.arange(total)
        if il is None:
            return idx
        tile, parts, part = il
        return idx[(idx // tile) % parts == part]

    for total in (1000, 40 ** 3, 100000):
        for K, N, want in [(1, 1, 2), (1, 8, 2), (8, 8, 2), (5, 8, 2), (3, 2, 2), (1, 2, 4), (16, 8, 2)]:
            for r in range(N):
                plan = multigpu.plan_cpi(total, K, r, N)
                fine = multigpu.refine_plan(plan, want)
                assert len(fine) >= min(want, len(plan)) and [p[0] for p in fine] == sorted(p[0] for p in fine)
                for k in {p[0] for p in plan}:
                    a = np.sort(np.concatenate([indices(total, il) for (kk, _, _, il) in plan if kk == k]))
                    b = np.sort(np.concatenate([indices(total, il) for (kk, _, _, il) in fine if kk == k]))
                    assert np.array_equal(a, b)
                for (k, first, count, il) in fine:
                    assert multigpu.part_ray_count(count, il) == len(indices(total, il))
