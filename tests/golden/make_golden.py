#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference ships no fixtures and cannot be run here, so these vectors are outputs of the
oracle (oracle/rts_oracle.cpp) on this repository's synthetic scenes; they freeze the oracle's
behaviour (CPU test: oracle == golden) and give the GPU tests a checker that needs no oracle
run (GPU == golden).  Data only: inputs are regenerated from rts_amd.scenes by name.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O          # noqa: E402
from rts_amd import scenes              # noqa: E402
import helpers as H                     # noqa: E402


def golden_specs():
    return {
        "c1_plate": scenes.config1(),
        "c2_sphere_w16": scenes.config2(subdiv=3, W=16, rx_radius=400.0),
        "c3_aircraft_w14": scenes.config3(W=14, detail=0.03, rx_radius=400.0),
        "multi_w16": scenes.config_multi(W=16),
    }


def dump(name, spec):
    o = H.oracle_trace(O, spec)
    res = o["results"]
    wl = spec["c"] / spec["carrier"]
    rx, rxi, slots = O.filter_finalise(res, o["path"], [1.0] * len(spec["meshes"]), wl, 1.0, 1.0, spec["carrier"], spec["c"])
    lit = O.aggregate_literal(rx, rxi, spec["c"], spec["carrier"], spec["W"] ** 3)
    uniq = O.unique_paths(lit["pathMatch"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        received=res["received"].astype(np.int16), reflDepth=res["reflDepth"].astype(np.uint8),
                        hit_prim=o["hit_prim"], hit_t=o["hit_t"], path=o["path"].astype(np.int8),
                        rx_slots=slots.astype(np.uint32), rx_records=res[slots.astype(np.int64)].view(np.uint8).reshape(-1, 144),
                        rx_rcs_angle=o["rcs_angle"][slots.astype(np.int64)],
                        fin_power=rx["power"], fin_doppler=rx["doppler"],
                        agg_power=lit["results"]["power"], agg_doppler=lit["results"]["doppler"], agg_delay=lit["delay"],
                        agg_phase=lit["phase"], agg_pathMatch=lit["pathMatch"], unique=uniq,
                        counters=np.array([o["counters"]["segments"], o["counters"]["shaded"]], np.int64))
    print(name, "rays", len(res), "received", len(slots), "responses", len(uniq), "segments", o["counters"]["segments"])


if __name__ == "__main__":
    for name, spec in golden_specs().items():
        dump(name, spec)
