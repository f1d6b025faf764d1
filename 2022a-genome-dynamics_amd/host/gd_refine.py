#!/usr/bin/env python3
"""gd_refine -- the 100 kb initial conformation from the packed coarse-grained one.

The reference's `scripts/refine <trajectory>` (5-sim-genome/src/refine/run.py:9-46): the last frame of the `packing`
phase is upsampled chain by chain (cubic spline at the bin midpoints, refinement.py:9-19; x init_coarse_graining,
truncated to the chain's fine length), nucleolar beads start on top of their NOR, and the result replaces
/snapshots/relaxation/0/positions (float64).  HDF5 access goes through gd_h5tool (no h5py in this image).
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gd_prepare import _module, h5tool      # noqa: E402


def _dataset(tmp, trajfile, path, cols, dtype=float):
    out = os.path.join(tmp, "d.f64")
    h5tool("dataset", trajfile, path, out, capture=True)
    return np.fromfile(out, dtype="<f8").reshape(-1, cols).astype(dtype)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="gd_refine", description="Refine coarse initial conformation")
    ap.add_argument("trajfile")
    a = ap.parse_args(argv)
    prep = _module("prepare")
    with tempfile.TemporaryDirectory() as tmp:
        config = json.loads(h5tool("strings", a.trajfile, "/metadata/config", capture=True))
        if config["init_refinement_method"] != "spline":
            raise SystemExit(f"unknown refinement method {config['init_refinement_method']!r}")     # refinement.py:22-24
        n_particles = _dataset(tmp, a.trajfile, "/metadata/ab_factors", 2).shape[0]       # == particle_types.shape[0]
        chrom_ranges = _dataset(tmp, a.trajfile, "/metadata/chromosome_ranges", 2, int)
        bonds = _dataset(tmp, a.trajfile, "/metadata/nucleolus_bonds", 2, int)
        init_chains = _dataset(tmp, a.trajfile, "/snapshots/packing/metadata/chromosome_ranges", 2, int)
        last = h5tool("steps", a.trajfile, "packing", capture=True).split()[-1]
        coarse = _dataset(tmp, a.trajfile, f"/snapshots/packing/{last}/positions", 3)
        fine = prep.refine_positions(coarse, init_chains[:len(chrom_ranges)], chrom_ranges, int(config["init_coarse_graining"]), bonds, n_particles)
        out = os.path.join(tmp, "fine.f64")
        fine.astype("<f8").tofile(out)
        h5tool("put-positions-f64", a.trajfile, "relaxation", 0, out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
