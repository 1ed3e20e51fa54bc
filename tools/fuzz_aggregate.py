#!/usr/bin/env python3
"""Fuzz of the aggregation boundary: rs::kernel_wrapper (aggregation.cuh:19-22) on random received sets against the literal
O(R^2 D) restatement of myKernel1 / myKernel2 (oracle.aggregate_literal): `pathMatch` must be equal element for element, the
aggregated power / delay within 1e-11 / 1e-12 relative, Doppler and phase within the tolerances of the parity tests, and the
untouched fields must come back unchanged.  Random R (1..4000), depth D (1..16), receivers (1..400), targets (1..254), share
of direct rays (0..1), heavy ties (few distinct paths) and all-distinct paths; key widths from 2 to > 128 bits.
   python tools/fuzz_aggregate.py [n_cases] [seed0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rts_amd import api  # noqa: E402
from oracle import oracle as O  # noqa: E402

C0 = 299792458.0


def case(seed):
    rng = np.random.default_rng(seed)
    R = int(10 ** rng.uniform(0, 3.6)); D = int(rng.integers(1, 17)); n_rx = int(10 ** rng.uniform(0, 2.6)); n_targ = int(10 ** rng.uniform(0, 2.4))
    n_targ = min(n_targ, 254)
    p_direct = float(rng.choice([0.0, 0.05, 0.3, 1.0], p=[0.2, 0.4, 0.3, 0.1]))
    a = np.zeros(R, O.PRD_DTYPE)
    a["received"] = rng.integers(0, n_rx, R); a["refrIndex"] = 1.0
    a["power"] = 10 ** rng.uniform(-14, -6, R); a["rayLength"] = rng.uniform(10, 5.0e4, R); a["doppler"] = rng.normal(size=R) * 10 ** rng.uniform(0, 4)
    paths = np.full((R, D), -1, np.int32)
    depth = rng.integers(1, D + 1, R)
    depth[rng.random(R) < p_direct] = 0
    few = rng.random() < 0.5                                            # heavy ties: rays share a handful of paths
    pool = [rng.integers(0, n_targ, D) for _ in range(int(rng.integers(1, 6)))]
    for i in range(R):
        src = pool[int(rng.integers(0, len(pool)))] if few else rng.integers(0, n_targ, D)
        paths[i, :depth[i]] = src[:depth[i]]
    a["reflDepth"] = depth
    bits = D * int(np.ceil(np.log2(n_targ + 1))) + int(np.ceil(np.log2(max(n_rx, 2))))
    return a, paths, float(10 ** rng.uniform(8, 11)), dict(R=R, D=D, n_rx=n_rx, n_targ=n_targ, p_direct=p_direct, few=few, key_bits=bits)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    wide = 0; rays = 0
    for seed in range(seed0, seed0 + n):
        a, paths, fc, info = case(seed)
        try:
            lit = O.aggregate_literal(a, paths, C0, fc, 10 ** 6)
            got = api.kernel_wrapper(a, paths, C0, fc, 10 ** 6)
            assert np.array_equal(got["pathMatch"], lit["pathMatch"]), "pathMatch"
            np.testing.assert_allclose(got["results"]["power"], lit["results"]["power"], rtol=1e-11)
            np.testing.assert_allclose(got["results"]["doppler"], lit["results"]["doppler"], rtol=1e-10, atol=1e-9 * max(1.0, float(np.abs(a["doppler"]).max())))
            np.testing.assert_allclose(got["delay"], lit["delay"], rtol=1e-12)
            np.testing.assert_allclose(got["phase"], lit["phase"], rtol=1e-9, atol=1e-9)
            for f in ("rayLength", "received", "reflDepth", "firstHitPoint", "prevHitPoint"):
                assert np.array_equal(got["results"][f], a[f]), f
        except Exception as e:
            print("FAILED seed %d %s: %r" % (seed, info, e), flush=True)
            raise
        wide += 1 if info["key_bits"] > 64 else 0; rays += info["R"]
        if (seed - seed0 + 1) % 50 == 0:
            print("%d cases ok (%d with keys wider than 64 bits, %d rays)" % (seed - seed0 + 1, wide, rays), flush=True)
    print("ALL OK: %d cases, %d with keys wider than 64 bits, %d rays" % (n, wide, rays))


if __name__ == "__main__":
    main()
