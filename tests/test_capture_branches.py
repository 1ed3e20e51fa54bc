"""Receiver-capture and Earth branches of the miss program (ray_tracer.cu:260-478) that the BASELINE scenes never reach,
and the same scenes at the coordinates the reference is written for (Earth-centred, |x| ~ 6.4e6 m).

CPU part (no marker): the oracle's branch-coverage counters prove that each scene really drives the restatement through
the branch it claims to cover, and the double capture of quirk 4 is checked against its closed form.
GPU part (-m gpu): the HIP path, through the C-ABI, is bit-identical to the oracle (brute-force closest hit) on them.
"""
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402
from rts_amd import scenes as S  # noqa: E402

ECEF_CASES = {"equator": dict(), "pole": dict(lat=math.pi / 2), "oblique": dict(lat=0.6, lon=-2.0)}


def traced_with_coverage(O, spec, **kw):
    O.coverage()
    o = H.oracle_trace(O, spec, **kw)
    return o, O.coverage()


# ------------------------------------------------------------------------------------------------ CPU: coverage is real
def test_miss_branch_scene_covers_what_it_claims(oracle):
    spec = S.config_miss_branches()
    o, cov = traced_with_coverage(oracle, spec)
    r = o["results"]
    per_rx = np.bincount(r["received"][r["received"] >= 0], minlength=len(spec["rx"]))
    assert cov["win_maxphi"] > 100 and cov["win_minphi"] > 100          # windows crossing +pi/2 and -pi/2 (:354-368)
    assert cov["capture_region2"] > 1000                                # captured by the SECOND (theta, phi) region only (:375)
    assert cov["both_roots"] > 100                                      # both roots inside the window, nearest wins (:378-381)
    assert cov["second_root_only"] > 1000
    assert cov["recapture"] > 1000 and cov["recapture_reflected"] >= 50  # quirk 4 on direct AND on reflected rays
    assert per_rx[0] > 10 and per_rx[1] > 10 and per_rx[4] > 100 and per_rx[5] > 1000
    assert cov["earth_tested"] > 1000 and cov["earth_both"] == 0         # origin-centred scene: inside the Earth sphere, t1 only
    assert cov["earth_root1"] == cov["earth_tested"]


def test_double_capture_closed_form(oracle):
    """quirk 4 (ray_tracer.cu:272, 393-426): no `break` over receivers.  A reflected ray crossing the windows of rx2 and
    rx3 has power multiplied by 1/((4 pi)^2 d^2) TWICE and both segment lengths added; the last receiver wins."""
    spec = S.config_miss_branches()
    both, cov = traced_with_coverage(oracle, spec)
    single_spec = dict(spec, rx=[spec["rx"][3]])                         # rx3 alone
    single = H.oracle_trace(oracle, single_spec)
    b, s = both["results"], single["results"]
    sel = (b["received"] == 3) & (s["received"] == 0) & (b["reflDepth"] > 0) & (b["power"] != s["power"])
    # rx2 and rx3 are the only receivers crossed by rays that end up at rx3 after a reflection
    assert sel.sum() == cov["recapture_reflected"] > 0
    t2 = b["rayLength"][sel] - s["rayLength"][sel]                       # length added by the first capture
    assert np.all(t2 > 0)
    # bounce directions are f32 unit vectors widened to f64: |dir| = 1 +- 1e-7, so |end - prev| = t (1 +- 1e-7)
    np.testing.assert_allclose(b["power"][sel] / s["power"][sel], 1.0 / ((4 * math.pi) ** 2 * t2 ** 2), rtol=1e-6)
    # a direct ray captured twice: power is OVERWRITTEN (:413), only the lengths add up
    d = (b["received"] == 3) & (s["received"] == 0) & (b["reflDepth"] == 0) & (b["rayLength"] != s["rayLength"])
    assert d.sum() > 1000
    assert np.all(b["rayLength"][d] > s["rayLength"][d])


@pytest.mark.parametrize("up", [True, False])
def test_pole_scene_takes_the_phi_correction(oracle, up):
    o, cov = traced_with_coverage(oracle, S.config_pole(up))
    assert cov["phi_low"] > 0 and cov["phi_high"] > 0                    # :332-340, reachable only where atan2f returns f32(pi/2) > pi/2
    assert cov["capture_region1"] > 0 and cov["capture_region2"] > 0
    assert (o["results"]["received"] >= 0).sum() > 0


@pytest.mark.parametrize("case", sorted(ECEF_CASES))
def test_ecef_scene_makes_the_earth_test_live(oracle, case):
    spec = S.translate(S.config_miss_branches(), S.ecef_offset(**ECEF_CASES[case]))
    o, cov = traced_with_coverage(oracle, spec)
    assert cov["earth_both"] > 100                                       # rays pointing down: t0 AND t1 added (quirk 6, :462-471)
    assert cov["earth_miss"] > 100                                       # rays pointing up: no root
    assert cov["earth_root0"] == cov["earth_root1"] == cov["earth_both"]
    assert cov["recapture_reflected"] > 0 and cov["capture_region2"] > 1000
    assert (o["results"]["received"] >= 0).sum() > 4000


# ------------------------------------------------------------------------------------------------ GPU parity
def _full_parity(api, O, spec, motion=None, threads=1):
    n = spec["W"] ** 3
    tr, st = H.gpu_trace(api, spec, motion=motion)
    g = tr.all_rays(n)
    o = H.oracle_trace(O, spec, motion=motion, threads=threads)          # brute-force closest hit
    H.compare_full(o, g, n)
    assert st["segments"] == o["counters"]["segments"] and st["shaded"] == o["counters"]["shaded"]
    rec = tr.received()
    idx = np.nonzero(o["results"]["received"] >= 0)[0]
    assert np.array_equal(rec["slots"], idx.astype(np.uint64))
    H.assert_prd_equal(o["results"][idx], rec["results"], "received records")
    assert np.array_equal(o["path"][idx], rec["path"])
    tr.close()
    return o, st


@pytest.mark.gpu
def test_gpu_miss_branches(rts, oracle):
    o, st = _full_parity(rts, oracle, S.config_miss_branches())
    assert st["received"] > 4000


@pytest.mark.gpu
@pytest.mark.parametrize("up", [True, False])
def test_gpu_pole_phi_correction(rts, oracle, up):
    o, st = _full_parity(rts, oracle, S.config_pole(up))
    assert st["received"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(ECEF_CASES))
def test_gpu_miss_branches_ecef(rts, oracle, case):
    """every branch above again with transmitter, receivers and targets at |x| = 6.388e6 m: the receiver quadratic's C
    term cancels catastrophically there (terms of 4e13 m^2), the Earth test is live for both roots"""
    spec = S.translate(S.config_miss_branches(), S.ecef_offset(**ECEF_CASES[case]))
    o, st = _full_parity(rts, oracle, spec)
    assert st["received"] > 4000


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["pole", "oblique"])
def test_gpu_c3_ecef_rotated_brute_force(rts, oracle, case):
    """C3 geometry (reduced: 10 068 triangles, W = 40) translated to Earth-centred coordinates 10 km above the Earth
    sphere, target rotated by a float-trig matrix and moving; against the oracle's BRUTE-FORCE closest hit"""
    spec = S.translate(S.config3(W=40, detail=0.1, rx_radius=300.0), S.ecef_offset(**ECEF_CASES[case]))
    motion = [dict(spec["motion"][0], rotation=rts.rotation_matrix(0.7, -0.3, 1.1))]
    oracle.coverage()
    o, st = _full_parity(rts, oracle, spec, motion=motion, threads=8)
    cov = oracle.coverage()
    assert cov["earth_both"] > 100 and cov["earth_miss"] > 100
    assert st["shaded"] > 5000 and st["received"] > 50000
