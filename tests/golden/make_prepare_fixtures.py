#!/usr/bin/env python3
"""Generates tests/golden/prepare_fixtures.npz by IMPORTING the reference's own Python modules
(5-sim-genome/src/prepare/system_definition.py, 5-sim-genome/src/refine/refinement.py) in this container and
recording their outputs for small fixed inputs.  Only inputs and outputs are stored; run here (the reference
tree does not exist on the GPU box):  python tests/golden/make_prepare_fixtures.py"""
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/5-sim-genome/src")
from prepare.system_definition import make_system_definition  # noqa: E402
from refine.refinement import refine_path_spline  # noqa: E402

GENOME = [  # chain start end A B tags
    ("chr1", 0, 100000, 1.0, 0.0, "A"), ("chr1", 100000, 200000, 0.0, 1.0, "B"), ("chr1", 200000, 300000, 0.5, 0.5, "u,cen"),
    ("chr1", 300000, 400000, 0.5, 0.5, "cen"), ("chr1", 400000, 500000, 0.7, 0.3, "A,anor"), ("chr1", 500000, 600000, 0.0, 1.0, "B,bnor"),
    ("chr2", 0, 100000, 0.25, 0.75, "B"), ("chr2", 100000, 200000, 0.5, 0.5, "u"), ("chr2", 200000, 300000, 1.0, 0.0, "anor"),
    ("chr2", 300000, 400000, 1.0, 0.0, "A,L1"), ("chrX", 0, 100000, 0.0, 1.0, "B"), ("chrX", 100000, 200000, 0.0, 1.0, "B"),
]
CONFIG = {"nucleolus_sidebeads": 2, "nucleolus_a_factor": 5, "nucleolus_b_factor": 5}


def main():
    genome = pd.DataFrame(GENOME, columns=["chain", "start", "end", "A", "B", "tags"])
    sd = make_system_definition(genome, CONFIG)
    out = {
        "types": np.array([p.type for p in sd.particles]), "ab": np.array([[p.A, p.B] for p in sd.particles]),
        "chains": np.array([[c.start, c.end, c.cen_start, c.cen_end] for c in sd.chromatin_chains]),
        "nucleolus_spans": np.array([[s.start, s.end] for s in sd.nucleolus_spans]).reshape(-1, 2),
        "nucleolus_bonds": np.array(sd.nucleolus_bonds).reshape(-1, 2),
    }
    rng = np.random.default_rng(3)
    for k, (m, n) in enumerate([(8, 80), (5, 23), (31, 3100)]):
        path = np.cumsum(rng.normal(size=(m, 3)), axis=0)
        out[f"path{k}"] = path
        out[f"fine{k}"] = refine_path_spline(path, n)
    np.savez_compressed(os.path.join(HERE, "prepare_fixtures.npz"), **out)
    json.dump({"genome": GENOME, "config": CONFIG, "chain_names": [c.name for c in sd.chromatin_chains],
               "nucleolus_names": [s.name for s in sd.nucleolus_spans]}, open(os.path.join(HERE, "prepare_fixtures.json"), "w"))
    print("ok", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
