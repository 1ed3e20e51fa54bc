"""CPU tests of the product's host side: the C-ABI library loads and exports every declared
symbol, the host scene helpers reproduce the reference's builders bit-for-bit (checked against
the oracle), the group-table algebra reproduces the reference's aggregation semantics, and the
compute entry points refuse to run without a GPU (there is no CPU fallback)."""
import ctypes as C
import math
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C0 = 299792458.0


def test_library_exports_every_declared_symbol(rts):
    from rts_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "rts_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(rts_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "librts_amd.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared
    # the C++ symbol SOARS links against: rs::kernel_wrapper(PerRayData*, int*, unsigned, unsigned, unsigned, unsigned,
    # double, double, double*, double*, double*, double*, double*, int*)   (aggregation.cuh:19-22)
    assert hasattr(L, "_ZN2rs14kernel_wrapperEP10PerRayDataPijjjjddPdS3_S3_S3_S3_S2_")


def test_no_cpu_fallback(rts):
    """without a GPU the compute path must fail loudly, not silently compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    from rts_amd import _lib
    with pytest.raises(_lib.RtsError) as e:
        rts.Tracer(4, 1)
    assert e.value.code == _lib.RTS_ERR_NO_DEVICE
    z = np.zeros(1, _lib.PRD_DTYPE)
    with pytest.raises(_lib.RtsError):
        rts.kernel_wrapper(z, np.zeros((1, 1), np.int32), C0, 1e9, 10)


def test_prd_header_layout_matches_reference_struct():
    """include/rts_prd.h compiles as plain C++ and has the 144-byte layout"""
    import subprocess, tempfile
    src = '#include "rts_prd.h"\n#include <cstdio>\nint main(){ printf("%zu %zu", sizeof(PerRayData), alignof(PerRayData)); return 0; }\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode()
    assert out == "144 16"


@pytest.mark.parametrize("ypr", [(0.0, 0.0, 0.0), (0.3, -0.2, 1.1), (math.pi / 2, math.pi, -2.5)])
def test_rect_mesh_matches_oracle(rts, oracle, ypr):
    a = rts.rect_mesh(2.0, 3.5, 0.25, *ypr); b = oracle.rect_mesh(2.0, 3.5, 0.25, *ypr)
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4])
def test_sphere_mesh_matches_oracle(rts, oracle, n):
    """the sort-based de-duplication must give the std::set vertex/triangle ORDER of ray_tracer.cpp:397-418"""
    a = rts.sphere_mesh(n, 5.0, 0.4, 0.1, -0.7); b = oracle.sphere_mesh(n, 5.0, 0.4, 0.1, -0.7)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))


def test_file_mesh_matches_oracle(rts, oracle, tmp_path):
    rng = np.random.default_rng(11)
    vf, nf = tmp_path / "v.txt", tmp_path / "n.txt"
    for f in (vf, nf):
        with open(f, "w") as fh:
            for row in rng.normal(size=(33, 9)) * 10:
                fh.write("%.17g %.17g %.17g, %.17g %.17g %.17g, %.17g %.17g %.17g,\n" % tuple(row))
    a = rts.file_mesh(str(vf), str(nf), 0.2, 0.3, 0.4); b = oracle.file_mesh(str(vf), str(nf), 0.2, 0.3, 0.4)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8))
    # the same lines WITHOUT the final comma (every line / every other line): the reference's fscanf has matched its nine conversions before it
    # looks for that comma and returns 9 either way (ray_tracer.cpp:461) -- such files load, and load the same
    for k, keep in enumerate((lambda i: False, lambda i: i % 2 == 0)):
        vf2, nf2 = tmp_path / ("v%d.txt" % k), tmp_path / ("n%d.txt" % k)
        for src, dst in ((vf, vf2), (nf, nf2)):
            lines = open(src).read().splitlines()
            open(dst, "w").write("".join((ln if keep(i) else ln.rstrip().rstrip(",")) + "\n" for i, ln in enumerate(lines)))
        a2 = rts.file_mesh(str(vf2), str(nf2), 0.2, 0.3, 0.4); b2 = oracle.file_mesh(str(vf2), str(nf2), 0.2, 0.3, 0.4)
        for x, y, z in zip(a2, b2, a):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)) and np.array_equal(x.view(np.uint8), z.view(np.uint8))
    from rts_amd import _lib
    with pytest.raises(_lib.RtsError) as e:            # the reference exit()s (ray_tracer.cpp:455-458); the library returns a status
        rts.file_mesh(str(tmp_path / "nope"), str(nf))
    assert e.value.code == _lib.RTS_ERR_IO


@pytest.mark.parametrize("where", ["vertices", "normals"])
@pytest.mark.parametrize("damage", ["missing comma", "a word", "seven numbers", "short file"])
def test_file_mesh_reports_malformed_lines(rts, tmp_path, where, damage):
    """a line that does not yield its nine numbers is RTS_ERR_IO naming the file and the triangle -- the reference only tests
    fscanf() == EOF (ray_tracer.cpp:459-476) and carries on with zeros (NaN normals downstream); VERDICT r3 weak #11"""
    from rts_amd import _lib
    rng = np.random.default_rng(5)
    rows = ["%.17g %.17g %.17g, %.17g %.17g %.17g, %.17g %.17g %.17g,\n" % tuple(r) for r in rng.normal(size=(12, 9))]
    bad = list(rows)
    if damage == "missing comma":
        bad[7] = bad[7].replace(",", "", 1)
    elif damage == "a word":
        bad[7] = bad[7].replace(bad[7].split()[4], "nan-ish?", 1)
    elif damage == "seven numbers":
        bad[7] = " ".join(bad[7].split()[:7]) + "\n"
    else:                                                   # 12 lines counted in the vertex file, fewer to read here
        bad = bad[:7] if where == "normals" else bad[:7] + ["\n"] * 5
    vf, nf = tmp_path / "v.txt", tmp_path / "n.txt"
    open(vf, "w").writelines(bad if where == "vertices" else rows); open(nf, "w").writelines(bad if where == "normals" else rows)
    with pytest.raises(_lib.RtsError) as e:
        rts.file_mesh(str(vf), str(nf))
    assert e.value.code == _lib.RTS_ERR_IO and "triangle 8 of 12" in str(e.value) and ("v.txt" if where == "vertices" else "n.txt") in str(e.value), str(e.value)
    open(vf, "w").writelines(rows); open(nf, "w").writelines(rows)          # the undamaged pair loads
    v, t, n = rts.file_mesh(str(vf), str(nf))
    assert t.shape == (12, 3) and np.isfinite(v).all() and np.isfinite(n).all()


def test_plan_whole_pulses(rts):
    """rts_plan_cpi, RTS_SHARD_PULSES_WHOLE (round 5; bench.py's strong-scaling default): whole pulses only, contiguous runs, every pulse exactly once, the
    workers' counts at most one apart; with fewer pulses than workers the left-over rule of RTS_SHARD_PULSES (groups of workers share a pulse)"""
    from rts_amd import multigpu as M
    total = 97 ** 3
    for K, N in [(20, 8), (21, 8), (7, 3), (16, 8), (8, 8), (1000, 7), (5, 1)]:
        plans = [M.plan_whole(total, K, r, N) for r in range(N)]
        assert all(il is None and first == 0 and count == total for p in plans for (_, first, count, il) in p)
        flat = [k for p in plans for (k, _, _, _) in p]
        assert flat == list(range(K)), (K, N)                               # contiguous runs in rank order
        assert max(len(p) for p in plans) - min(len(p) for p in plans) <= 1
    for K, N in [(3, 8), (1, 4), (5, 8)]:
        assert [M.plan_whole(total, K, r, N) for r in range(N)] == [M.plan_cpi(total, K, r, N) for r in range(N)]


def test_deal_tiles_partial_table(rts):
    """64 plan tiles, three workers, records for three tiles only (500 / 100 / 50): the 61 tiles without a record are spread by
    COUNT -- they used to enter the longest-first heap with cost 1 and all went to the two lightest workers (0 / ~6 / ~55)"""
    from rts_amd import api
    rec = np.zeros(64, np.uint32); rec[[5, 20, 41]] = [500, 100, 50]
    part, cost = api.deal_tiles(rec, 64 * 64, 64, 3)
    assert sorted(int(part[i]) for i in (5, 20, 41)) == [0, 1, 2]
    cnt = np.bincount(part, minlength=3)
    assert cnt.max() - cnt.min() <= 1 and cnt.sum() == 64, cnt
    assert sorted(int(x) for x in cost) == sorted([500 + int(cnt[part[5]]) - 1, 100 + int(cnt[part[20]]) - 1, 50 + int(cnt[part[41]]) - 1])


def test_deal_tiles_spreads_long_walks_by_count(rts):
    """round 5: tiles flagged LONG / LONGISH WALKS (bits 31 / 30 of a record: the cooperative kernel's candidates) are dealt first and by COUNT -- a part's time
    follows how many of them it holds more closely than their recorded cost -- the rest longest-first by cost on top of that; every tile is dealt exactly once"""
    from rts_amd import api
    rng = np.random.default_rng(5)
    n = 4096
    rec = rng.integers(1, 200, n).astype(np.uint32)
    flagged = rng.choice(n, 96, replace=False)
    rec[flagged[:48]] = (rng.integers(2000, 90000, 48).astype(np.uint32)) | np.uint32(0x80000000)       # LONG WALKS, costs over two decades
    rec[flagged[48:]] = (rng.integers(500, 3000, 48).astype(np.uint32)) | np.uint32(0x40000000)         # LONGISH
    part, cost = api.deal_tiles(rec, n * 64, 64, 8)
    held = np.bincount(part[flagged], minlength=8)
    assert held.max() - held.min() <= 1 and held.sum() == 96, held                                       # by count (dealt by cost alone: 3 .. 20 per part)
    assert np.bincount(part, minlength=8).sum() == n
    tot = np.array([int((rec[part == r] & 0x3fffffff).sum()) for r in range(8)])
    assert np.array_equal(np.sort(tot), np.sort(np.asarray(cost, np.int64)))
    plain = np.array([int((rec[(part == r) & ((rec >> 30) == 0)] & 0x3fffffff).sum()) for r in range(8)])
    assert tot.max() - tot.min() <= max(200, (rec[flagged] & 0x3fffffff).max()), (tot, plain)            # the unflagged tiles fill the parts up as far as one flagged tile's cost allows


def check_bvh4(nodes, leaf_prim, root, tris_verts, all_reachable=True):
    """invariants of one mesh's BVH4 as the library stores it (nodes [n][32] f32 view of the 128-byte records): every node reachable
    exactly once from the root, child boxes nested in the parent's, unused slots unreachable points, every triangle in at least
    one leaf slot, an unsplit triangle strictly inside its (padded) box, the boxes of a split triangle's references together
    covering its vertices and its centroid"""
    nprim = tris_verts.shape[0]
    refs = np.bincount(leaf_prim, minlength=nprim)
    assert refs.min() >= 1
    child = nodes[:, 24:28].copy().view(np.int32)
    lo = np.stack([nodes[:, 0:4], nodes[:, 4:8], nodes[:, 8:12]], axis=2); hi = np.stack([nodes[:, 12:16], nodes[:, 16:20], nodes[:, 20:24]], axis=2)
    seen_nodes = np.zeros(len(nodes), bool); seen_leaves = np.zeros(len(leaf_prim), bool); covered = np.zeros((nprim, 4), bool)
    stack = [(int(root), np.full(3, -np.inf), np.full(3, np.inf))]
    while stack:
        i, plo, phi = stack.pop()
        assert not seen_nodes[i]; seen_nodes[i] = True
        used = 0
        for k in range(4):
            c = child[i, k]
            if c == 0x7fffffff:
                assert (lo[i, k] == hi[i, k]).all() and (lo[i, k] > 1e38).all()
                continue
            used += 1
            assert (lo[i, k] >= plo).all() and (hi[i, k] <= phi).all()
            if c < 0:
                leaf = ~c; assert not seen_leaves[leaf]; seen_leaves[leaf] = True
                p = leaf_prim[leaf]
                pts = np.concatenate([tris_verts[p], tris_verts[p].mean(axis=0, keepdims=True)])
                inside = ((pts > lo[i, k].astype(np.float64)) & (pts < hi[i, k].astype(np.float64))).all(axis=1)
                covered[p] |= inside
                if refs[p] == 1:
                    assert inside.all()
            else:
                stack.append((int(c), lo[i, k], hi[i, k]))
        assert used >= 1
    assert seen_leaves.all() and covered.all() and (seen_nodes.all() or not all_reachable)
    return refs


def test_host_sah_builder_invariants(rts):
    """rts_sah.cpp without a device (rts_build_hierarchy_host): the airframe's ellipsoids with their pole fans (split references),
    an icosphere, a single triangle, a mesh with a non-finite vertex (that triangle gets no leaf), no triangles at all"""
    from rts_amd import scenes
    v, t, _ = scenes.aircraft_mesh(detail=0.05)
    nodes, leaf, root = rts.build_hierarchy_host(v, t)
    refs = check_bvh4(nodes, leaf, root, v[t])
    assert refs.max() > 1 and len(leaf) > len(t)                      # the sliver fans are cut into several references
    nodes0, leaf0, root0 = rts.build_hierarchy_host(v, t, split_budget=0.0)
    assert len(leaf0) == len(t) and check_bvh4(nodes0, leaf0, root0, v[t]).max() == 1
    sv, st, _ = rts.sphere_mesh(3, 2.0)
    check_bvh4(*rts.build_hierarchy_host(sv, st), sv[st])
    one_v = np.array([[3.0, -2, -2], [3.0, 2, -2], [3.2, 0, 2.5]]); one_t = np.array([[0, 1, 2]], np.uint32)
    n1, l1, r1 = rts.build_hierarchy_host(one_v, one_t)
    assert len(n1) == 1 and set(l1.tolist()) == {0} and r1 == 0 and check_bvh4(n1, l1, r1, one_v[one_t]).max() <= 8
    bv = sv.copy(); bv[st[5, 1]] = np.nan
    nb, lb, rb = rts.build_hierarchy_host(bv, st)
    hit = np.isnan(bv[st]).any(axis=(1, 2))
    assert hit.sum() >= 1 and not np.isin(np.nonzero(hit)[0], lb).any() and set(np.nonzero(~hit)[0]) == set(lb.tolist())
    ne, le, re_ = rts.build_hierarchy_host(np.zeros((0, 3)), np.zeros((0, 3), np.uint32))
    assert len(le) == 0 and re_ < 0


def test_vertex_rotation_and_rx_sphere_match_oracle(rts, oracle):
    rng = np.random.default_rng(2)
    v = rng.normal(size=(50, 3))
    for ypr in [(0.1, 0.2, 0.3), (-2.0, 1.0, 0.5)]:
        assert np.array_equal(rts.vertex_rotation(v, *ypr), oracle.vertex_rotation(v, *ypr))
        R = rts.rotation_matrix(*ypr).reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)          # float trig: orthonormal to f32 accuracy only
    for pos, az, el in [((-1000.0, 5.0, 2.0), 0.3, -0.1), ((10.0, 20.0, 30.0), 2.5, 0.7), ((0.0, 0.0, 0.0), -3.0, 1.2)]:
        a = rts.rx_sphere(pos, az, el, 25.0, 1.0, 0.5); b = oracle.rx_sphere(pos, az, el, 25.0, 1.0, 0.5)
        for k in ("radius", "minTheta", "maxTheta", "minPhi", "maxPhi"):
            assert a[k] == b[k]
        assert np.array_equal(a["centre"], b["centre"])


# ------------------------------------------------------------------------------- group-table algebra
def numpy_group_table(rx_results, rx_paths, cspeed, carrier, base=0):
    """test-side reference group-by: (rx, path) -> partial sums (what rts_aggregate produces on the GPU)"""
    from rts_amd._lib import GROUP_DTYPE, RTS_MAX_DEPTH
    groups = {}
    for i in range(len(rx_results)):
        key = (int(rx_results["received"][i]),) + tuple(int(x) for x in rx_paths[i])
        delay = rx_results["rayLength"][i] / cspeed
        phase = -math.fmod(delay * 2 * math.pi * carrier, 2 * math.pi)
        g = groups.setdefault(key, [0.0, 0.0, 0.0, 0.0, 0.0, base + i])
        g[0] += 1; g[1] += math.sqrt(rx_results["power"][i]); g[2] += delay; g[3] += phase; g[4] += rx_results["doppler"][i]
    out = np.zeros(len(groups), GROUP_DTYPE)
    for j, (key, g) in enumerate(sorted(groups.items())):
        out[j]["rx"] = key[0]; out[j]["path"][:] = -1; out[j]["path"][:len(key) - 1] = key[1:]
        out[j]["direct"] = 1 if all(k < 0 for k in key[1:]) else 0
        out[j]["n"], out[j]["sum_sqrt_power"], out[j]["sum_delay"], out[j]["sum_phase"], out[j]["sum_doppler"] = g[:5]
        out[j]["min_ray"] = g[5]
    return out


def random_received_set(oracle, rng, R, D, n_rx, n_targ, p_direct=0.2):
    a = np.zeros(R, oracle.PRD_DTYPE)
    a["received"] = rng.integers(0, n_rx, R); a["refrIndex"] = 1.0
    a["power"] = rng.uniform(1e-12, 1e-9, R); a["rayLength"] = rng.uniform(1000, 3000, R); a["doppler"] = rng.normal(size=R) * 100
    paths = np.full((R, D), -1, np.int32)
    depth = rng.integers(1, D + 1, R)
    direct = rng.random(R) < p_direct
    depth[direct] = 0
    for i in range(R):
        paths[i, :depth[i]] = rng.integers(0, n_targ, depth[i])
    a["reflDepth"] = depth
    return a, paths


@pytest.mark.parametrize("seed,R,D,n_rx,n_targ", [(1, 60, 3, 2, 2), (2, 400, 4, 4, 3), (3, 25, 1, 1, 1), (4, 300, 6, 3, 1)])
def test_groups_to_responses_equals_reference_aggregation(rts, oracle, seed, R, D, n_rx, n_targ):
    """responses derived from the (receiver, path) group table == unique-path responses of the literal
    O(R^2) restatement of myKernel1/2 (aggregation.cu:32-97, ray_tracer.cpp:1290-1321), including the
    direct-ray rule and its collapse (quirk 9)"""
    rng = np.random.default_rng(seed)
    a, paths = random_received_set(oracle, rng, R, D, n_rx, n_targ)
    fc = 10e9
    lit = oracle.aggregate_literal(a, paths, C0, fc, 10 ** 6)
    uniq = oracle.unique_paths(lit["pathMatch"])
    resp = rts.groups_to_responses(numpy_group_table(a, paths, C0, fc))
    assert np.array_equal(resp["ray"].astype(np.int64), uniq.astype(np.int64))
    assert np.array_equal(resp["rx"], lit["results"]["received"][uniq])
    np.testing.assert_allclose(resp["power"], lit["results"]["power"][uniq], rtol=1e-12)
    np.testing.assert_allclose(resp["doppler"], lit["results"]["doppler"][uniq], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(resp["delay"], lit["delay"][uniq], rtol=1e-12)
    np.testing.assert_allclose(resp["phase"], lit["phase"][uniq], rtol=1e-9, atol=1e-12)
    assert np.array_equal(resp["n"], lit["npath"][uniq].astype(np.uint32))


def test_merge_groups_is_shard_invariant(rts, oracle):
    """splitting the received list into contiguous shards, grouping each and merging == grouping the whole"""
    rng = np.random.default_rng(9)
    a, paths = random_received_set(oracle, rng, 500, 4, 3, 2)
    fc = 10e9
    whole = numpy_group_table(a, paths, C0, fc)
    for cuts in ([0, 250, 500], [0, 10, 11, 300, 500], [0, 500], [0, 0, 123, 500]):
        parts = [numpy_group_table(a[lo:hi], paths[lo:hi], C0, fc, base=lo) for lo, hi in zip(cuts[:-1], cuts[1:])]
        merged = rts.merge_groups(np.concatenate(parts), 4)
        assert len(merged) == len(whole)
        assert np.array_equal(merged["rx"], whole["rx"]) and np.array_equal(merged["path"], whole["path"])
        assert np.array_equal(merged["min_ray"], whole["min_ray"]) and np.array_equal(merged["n"], whole["n"])
        for f in ("sum_sqrt_power", "sum_delay", "sum_phase", "sum_doppler"):
            np.testing.assert_allclose(merged[f], whole[f], rtol=1e-12, atol=1e-15)
        ra = rts.groups_to_responses(merged); rb = rts.groups_to_responses(whole)
        assert np.array_equal(ra["ray"], rb["ray"])
        np.testing.assert_allclose(ra["power"], rb["power"], rtol=1e-12)


def test_merge_cpi_takes_responses_formed_earlier(rts, oracle):
    """one rank (bench.py): a pulse's responses are formed when the pulse is collected and travel with its table through exchange_parts (no
    collective); merge_cpi then uses them -- the same responses it would have formed -- and still forms them itself where none came along or
    where a pulse's table has to be merged from several parts"""
    from rts_amd import multigpu
    rng = np.random.default_rng(21)
    fc = 10e9
    tabs = []
    for k in range(3):
        a, paths = random_received_set(oracle, rng, 120 + 40 * k, 3, 2, 2)
        tabs.append(numpy_group_table(a, paths, C0, fc))
    parts = [dict(pulse=0, groups=tabs[0], responses=rts.groups_to_responses(tabs[0])), dict(pulse=1, groups=tabs[1]),
             dict(pulse=2, groups=tabs[2][:50], responses=None), dict(pulse=2, groups=tabs[2][50:])]
    allp = multigpu.exchange_parts(parts, None, None)
    assert allp[0]["responses"] is parts[0]["responses"] and allp[1]["responses"] is None
    out = multigpu.merge_cpi(allp, 3)
    assert out[0][0] is parts[0]["responses"]
    for k in (0, 1):
        want = rts.groups_to_responses(tabs[k])
        assert out[k][0].tobytes() == want.tobytes() and out[k][1].tobytes() == np.ascontiguousarray(tabs[k]).tobytes()
    merged = rts.merge_groups(tabs[2], 3)
    assert out[2][0].tobytes() == rts.groups_to_responses(merged).tobytes()


def test_empty_tables(rts):
    from rts_amd._lib import GROUP_DTYPE
    assert len(rts.merge_groups(np.zeros(0, GROUP_DTYPE), 3)) == 0
    assert len(rts.groups_to_responses(np.zeros(0, GROUP_DTYPE))) == 0


def test_scene_generators():
    from rts_amd import scenes
    v, t, n = scenes.aircraft_mesh()
    assert t.shape == (100000, 3) and t.max() == v.shape[0] - 1 and n.shape == v.shape
    a = v[t[:, 1]] - v[t[:, 0]]; b = v[t[:, 2]] - v[t[:, 0]]
    assert (np.linalg.norm(np.cross(a, b), axis=1) > 0).all()            # no degenerate triangles
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0)
    s = scenes.config3()
    assert s["W"] == 216 and s["max_refl"] == 6 and len(s["rx"]) == 4 and s["meshes"][0]["tris"].shape[0] == 100000
    s2 = scenes.config2()
    assert s2["meshes"][0]["tris"].shape[0] == 20480 and s2["W"] == 100 and s2["max_refl"] == 4
    s1 = scenes.config1()
    assert s1["meshes"][0]["tris"].shape[0] == 2 and s1["W"] == 22 and s1["max_refl"] == 1


def build_adapter_binary(out_path):
    """compiles tests/adapter/adapter_main.cpp (include/rts_adapter.hpp over the mock SOARS World) with g++"""
    import subprocess
    src = os.path.join(ROOT, "tests", "adapter", "adapter_main.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "adapter"),
                           src, "-L", os.path.join(ROOT, "rts_amd"), "-lrts_amd", "-Wl,-rpath," + os.path.join(ROOT, "rts_amd"), "-o", out_path])
    return out_path


def test_adapter_bench_compiles(rts, tmp_path):
    """tests/adapter/adapter_bench.cpp (the C++ boundary timed end to end) builds against the header and the library"""
    import subprocess
    out = str(tmp_path / "adapter_bench")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "adapter"),
                           os.path.join(ROOT, "tests", "adapter", "adapter_bench.cpp"), "-L", os.path.join(ROOT, "rts_amd"), "-lrts_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "rts_amd"), "-o", out])
    assert os.path.exists(out)


def test_adapter_header_compiles_and_links(rts, tmp_path):
    """the rs::RTS replacement (header-only, templated on the simulator's types) builds with a plain host
    compiler against the C-ABI and resolves rs::kernel_wrapper from librts_amd.so"""
    exe = build_adapter_binary(str(tmp_path / "adapter_main"))
    assert os.path.exists(exe)


def test_soars_traits_branch_compiles_and_links(rts, tmp_path):
    """the RTS_ADAPTER_WITH_SOARS branch of rts_adapter.hpp (rts_amd::SoarsTraits, rs::RTS with the reference's signature,
    ray_tracer.cpp:512) parses, type-checks and links against header-only stand-ins that carry the SOARS class and method
    NAMES of ray_tracer.cpp:50-60 (tests/adapter/soars_stub/: a mock, SOARS itself is not in the reference repository)"""
    import subprocess
    exe = str(tmp_path / "soars_glue")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "adapter", "soars_stub"),
                           os.path.join(ROOT, "tests", "adapter", "soars_glue.cpp"), "-L", os.path.join(ROOT, "rts_amd"), "-lrts_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "rts_amd"), "-o", exe])
    assert subprocess.run([exe], timeout=60).returncode == 0


def test_deal_tiles_longest_first(rts):
    """rts_deal_tiles (host code; ray sharding dealt by last-seen cost): every plan tile goes to exactly one worker, the workers'
    costs differ by at most the dearest tile, tiles without a record are spread by count, the deal is a function of the table
    alone (every rank computes the same map), flags in bits 30-31 do not count as cost, bad arguments are refused"""
    from rts_amd import api
    rng = np.random.default_rng(11)
    total = 216 ** 3; n = (total + 63) // 64
    rec = np.zeros(n, np.uint32)
    hot = rng.choice(n, 30000, replace=False)
    rec[hot] = rng.integers(1, 900000, hot.shape[0]).astype(np.uint32)
    rec[hot[:300]] |= np.uint32(0x80000000); rec[hot[300:500]] |= np.uint32(0x40000000)
    for tile, parts in [(4096, 8), (64, 3), (1024, 5), (4096, 1)]:
        part, cost = api.deal_tiles(rec, total, tile, parts)
        n_plan = (total + tile - 1) // tile
        assert part.shape[0] == n_plan and part.max() < parts
        c = np.maximum(np.add.reduceat((rec & 0x3fffffff).astype(np.uint64), np.arange(0, n, tile // 64)), 1)
        loads = np.array([int(c[part == r].sum()) for r in range(parts)], np.uint64)
        assert np.array_equal(loads, cost)
        assert int(loads.max() - loads.min()) <= int(c.max())
        raw = np.add.reduceat((rec & 0x3fffffff).astype(np.uint64), np.arange(0, n, tile // 64))
        tot = np.bincount(part, minlength=parts)                            # tiles nobody traced are dealt by COUNT, behind the recorded ones: every worker ends with
        assert tot.max() - tot.min() <= max(1, int(np.bincount(part[raw > 0], minlength=parts).max() - np.bincount(part[raw > 0], minlength=parts).min())), (tile, parts, tot)      # as many tiles as the records' deal allows
        part2, _ = api.deal_tiles(rec.copy(), total, tile, parts)
        assert np.array_equal(part, part2)
    # no records at all: round-robin by count
    part, cost = api.deal_tiles(np.zeros(n, np.uint32), total, 4096, 8)
    assert np.bincount(part).max() - np.bincount(part).min() <= 1
    for bad in [dict(tile=100), dict(parts=0), dict(total=total + 64)]:
        with pytest.raises(Exception):
            api.deal_tiles(rec, bad.get("total", total), bad.get("tile", 4096), bad.get("parts", 8))


def test_plan_cpi_in_the_library(rts):
    """rts_plan_cpi (the plan the C++ adapter and bench.py share): ray mode gives every rank its interleaved part of every
    pulse; min_items refines to that many items without changing what the rank owns; bad arguments are refused"""
    from rts_amd import multigpu, _lib
    total = 216 ** 3
    for world in (1, 2, 5, 8):
        for rank in range(world):
            p = multigpu.plan_rays(total, 3, rank, world)
            assert [k for k, _, _, _ in p] == [0, 1, 2]
            assert all(il == ((multigpu.IL_TILE, world, rank) if world > 1 else None) for _, _, _, il in p)
        cover = sum(multigpu.part_ray_count(total, il) for r in range(world) for (_, _, _, il) in multigpu.plan_rays(total, 1, r, world))
        assert cover == total
    for K, N, want in ((20, 8, 3), (1, 4, 3), (7, 3, 4)):
        for r in range(N):
            coarse = multigpu.plan_cpi(total, K, r, N); fine = multigpu.plan_cpi(total, K, r, N, min_items=want)
            assert fine == multigpu.refine_plan(coarse, want)
            assert sum(multigpu.part_ray_count(c, il) for _, _, c, il in fine) == sum(multigpu.part_ray_count(c, il) for _, _, c, il in coarse)
    with pytest.raises(_lib.RtsError):
        multigpu.plan_cpi(total, 4, 3, 3)                            # rank >= world


def test_product_trace_kernels_use_no_scratch():
    """the product instantiations of the trace kernel (COUNT = false) must not spill vector registers in any loop: 128 VGPRs at
    four waves per SIMD is the budget the kernel is written for, and a spilled draw of the tile queue once cost the counting
    builds whole tiles' worth of counters (rts_trace.hip, RTS_DRAW).  hipcc's own resource remarks and the ISA, device code
    only (no GPU needed).  Allowed: up to three values parked in scratch in the prologue (stores OUTSIDE every loop, once per persistent wave; the
    allocator has no register for them across the tile loop) and reloaded in the epilogue or -- loop invariants -- per draw / per tile, never deeper.
    Also checked here: the record fetch of a traversal step -- its global_load_dwordx4 group and the
    s_waitcnt vmcnt(0) that covers it -- is ONE inline-asm block, so no compiler-placed instruction can touch the destination
    registers while the loads are in flight.
    The asynchronous-bounce instantiations (k_trace<.., ASYNC = true>, an experiment that is off by default: RTS_ASYNC_IDLE0)
    carry their walk state through the shading code and do spill -- per tile or per advance phase (loop depth <= 3), never per walk
    step, which is what is checked for them."""
    import re, shutil, subprocess, tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "rts_amd", "csrc", "rts_trace.hip")
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
                            "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-S", src, "-o", os.path.join(td, "t.s")], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        isa = open(os.path.join(td, "t.s")).read()
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not name.startswith("_Z7k_traceILb0E"):
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)); vspill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)); occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        flags = [q == "1" for q in re.findall(r"Lb([01])", name.split("EEv")[0])]      # k_trace<COUNT, KEEP_ALL, REFR, COOP, ASYNC, AFFINE, VERS>
        assert len(flags) == 7, name
        async_k = flags[4]
        if async_k:
            assert scratch <= 64 and vspill <= 16, (name, scratch, vspill)
        else:
            assert scratch <= 16 and vspill <= 3, (name, scratch, vspill)
        coop = flags[3]; refr = flags[2]
        if coop:
            assert lds * 3 <= 160 * 1024 and occ >= (2 if refr else 3), (name, lds, occ)      # the cooperative kernel: three blocks per CU -- and three waves per SIMD: at 169 registers (one above
                                                                                              # 168 = 512 / 3 in granules of 8) it ran at TWO for most of round 4, unnoticed; __launch_bounds__ now says 3
        else:
            assert lds * 4 <= 160 * 1024 and occ >= (2 if refr else 4), (name, lds, occ)      # four blocks of four waves per CU
        body = isa[isa.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")].splitlines()
        in_loop = False; in_asm = False; asm_loads = 0; fetch_blocks = 0; depth = 0
        for line in body:
            t = line.strip()
            if t.startswith(".LBB") or t.startswith("; %bb"):
                in_loop = "in Loop" in t
                m = re.search(r"Depth=(\d+)", t); depth = int(m.group(1)) if (in_loop and m) else 0
            if t.startswith(";;#ASMSTART"):
                in_asm = True; asm_loads = 0
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm:
                if t.startswith("global_load_dwordx4"):
                    asm_loads += 1
                if t.startswith("s_waitcnt vmcnt(0)") and asm_loads:
                    assert asm_loads == 7, (name, asm_loads); fetch_blocks += 1; asm_loads = 0
            if async_k:                                            # (an experiment, off by default) no STORE to scratch below the tile level; r04: the base of the stack's spill
                assert "scratch_" not in t or depth <= 3 or t.startswith("scratch_load"), (name, depth, t)      # slab is reloaded in the (rare) deep-stack branch of its walk step; r05: a store per advance phase (depth 3) since the pre-filter grew -- never in the walk loop (depth 4)
                continue
            if "scratch_" in t:                                   # a value the tile loop has no register for, parked in scratch in the prologue: STORES only outside every loop (once per persistent
                assert not in_loop or (t.startswith("scratch_load") and depth <= 2), (name, depth, t)      # wave); RELOADS of such a loop-invariant value per draw or per tile (depth <= 2), never per segment or walk step (r05:
                                                                                                            # the dead-tile batches' screen took the last register the draw counter's address had)
        assert fetch_blocks >= 1 and asm_loads == 0, (name, fetch_blocks, asm_loads)     # every asm load group ends in its own wait
    assert seen == 16      # 4 ordinary (role fetch) + 4 ordinary (octant versions) + 4 cooperative + 1 cooperative (octant versions) + 2 asynchronous + 1 XCD-affine product instantiations


def test_fuzz_generator_versions_are_frozen(rts):
    """the regression seeds of test_differential_fuzz_seeds name scenes through tools/fuzz_equal.random_scene(seed, version):
    the old generator versions must keep producing the scenes the bugs were found in"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_equal as F
    sp, place, aim = F.random_scene(12661, 1)
    assert (place, aim, sp["W"], sp["max_refl"], len(sp["rx"])) == ("ecef", "at", 31, 4, 1)
    assert abs(sp["rx"][0]["radius"] - 0.8559495751013925) < 1e-15 and abs(sp["tx"]["origin"][0] + 868346.5622011841) < 1e-6
    sp, place, aim = F.random_scene(50301, 2)
    assert (place, aim, sp["W"], sp["max_refl"], len(sp["rx"])) == ("far", "at", 37, 2, 5)
    assert sp["tx"]["span"][0] == 1.400077894938366                 # the beam start whose cosine glibc's sincos and cos disagree on
