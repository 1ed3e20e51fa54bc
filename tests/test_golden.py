"""The oracle reproduces the committed golden vectors bit-for-bit (CPU).  The vectors were produced
by tests/golden/make_golden.py; this test freezes the oracle so that a later edit cannot silently
move the target the GPU path is compared against."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402
import helpers as H  # noqa: E402

NAMES = ["c1_plate", "c2_sphere_w16", "c3_aircraft_w14", "multi_w16"]


def check_against_golden(g, res, path, hit_prim, hit_t, rcs_angle=None):
    assert np.array_equal(res["received"].astype(np.int16), g["received"])
    assert np.array_equal(res["reflDepth"].astype(np.uint8), g["reflDepth"])
    assert np.array_equal(hit_prim, g["hit_prim"])
    assert np.array_equal(hit_t.view(np.uint32), g["hit_t"].view(np.uint32))
    assert np.array_equal(path.astype(np.int8), g["path"])
    slots = g["rx_slots"].astype(np.int64)
    assert np.array_equal(np.nonzero(res["received"] >= 0)[0], slots)
    from oracle.oracle import PRD_DTYPE
    want = np.ascontiguousarray(g["rx_records"]).reshape(-1).view(PRD_DTYPE)        # field-wise: struct padding is not data
    H.assert_prd_equal(np.ascontiguousarray(res[slots]), want, "received records vs golden")
    if rcs_angle is not None:
        np.testing.assert_allclose(rcs_angle[slots], g["rx_rcs_angle"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_golden(oracle, name):
    spec = make_golden.golden_specs()[name]
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    o = H.oracle_trace(oracle, spec)
    check_against_golden(g, o["results"], o["path"], o["hit_prim"], o["hit_t"], o["rcs_angle"])
    assert o["counters"]["segments"] == g["counters"][0] and o["counters"]["shaded"] == g["counters"][1]
    wl = spec["c"] / spec["carrier"]
    rx, rxi, slots = oracle.filter_finalise(o["results"], o["path"], [1.0] * len(spec["meshes"]), wl, 1.0, 1.0, spec["carrier"], spec["c"])
    assert np.array_equal(rx["power"], g["fin_power"]) and np.array_equal(rx["doppler"], g["fin_doppler"])
    lit = oracle.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], spec["W"] ** 3)
    assert np.array_equal(lit["pathMatch"], g["agg_pathMatch"]) and np.array_equal(oracle.unique_paths(lit["pathMatch"]), g["unique"])
    assert np.array_equal(lit["results"]["power"], g["agg_power"]) and np.array_equal(lit["delay"], g["agg_delay"])
