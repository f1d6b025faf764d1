#!/usr/bin/env python3
"""Generates tests/golden/wall_distance_fixtures.npz by IMPORTING the reference's own
5-sim-genome/src/analyze_lamina/geometry.py (the author's second-order distance of a point from an ellipsoid surface -- the
construction the oracle's wall term restates, oracle/gdyn_oracle.c "Ellipsoid wall") in this container and recording its
outputs for fixed points.  Only inputs and outputs are stored.  The module was written for numpy < 1.24 (`np.float`); the alias is
provided for the import, nothing of the reference is changed.  Run here:  python tests/golden/make_wall_fixtures.py"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if not hasattr(np, "float"):
    np.float = float          # the numpy the reference was written for had this alias
spec = importlib.util.spec_from_file_location("ref_geometry", "/root/reference/5-sim-genome/src/analyze_lamina/geometry.py")
geometry = importlib.util.module_from_spec(spec)
spec.loader.exec_module(geometry)


def main():
    rng = np.random.default_rng(20220101)
    out = {}
    for k, semi in enumerate([(1.0, 1.0, 1.0), (1.0, 0.8, 0.6), (1.3, 0.7, 1.1)]):
        semi = np.array(semi)
        v = rng.normal(size=(120, 3))
        v /= np.linalg.norm(v, axis=1)[:, None]
        scale = np.concatenate([rng.uniform(0.88, 0.995, 60), rng.uniform(1.005, 1.4, 60)])      # inside near the wall, outside
        pts = v * semi[None, :] * scale[:, None]
        out[f"semi{k}"] = semi
        out[f"points{k}"] = pts
        out[f"dist{k}"] = geometry.Ellipsoid(semi).distance_from_surface(pts)
    np.savez_compressed(os.path.join(HERE, "wall_distance_fixtures.npz"), **out)
    print("ok", {k: v.shape for k, v in out.items()}, "EPSILON", geometry.EPSILON)


if __name__ == "__main__":
    main()
