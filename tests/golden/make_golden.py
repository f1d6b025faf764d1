#!/usr/bin/env python3
"""Generates tests/golden/golden_small.npz with the CPU oracle (oracle/liboracle.so).

The reference cannot produce vectors for this path (its physics library is an empty
submodule), so these are oracle outputs: they pin the oracle against
accidental change (tests/test_golden.py, CPU) and are what the HIP path is compared to
on the GPU box (tests/test_parity_gpu.py).  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from util import CASES, TERMS, build, g  # noqa: E402

SEED = 20220101


def main():
    lib = g.Lib(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "liboracle.so"))
    out = {}
    for name in CASES:
        s, dt, kT, flags = build(lib, name)
        x0 = s.positions()
        out[f"{name}/x0"] = x0
        for t, m in TERMS.items():
            out[f"{name}/F_{t}"] = s.forces(m)
            out[f"{name}/E_{t}"] = s.energy(m)
        for tag, steps, noise, temp in (("philox1", 1, g.NOISE_PHILOX, kT), ("philox10", 10, g.NOISE_PHILOX, kT),
                                        ("zero20", 20, g.NOISE_ZERO, 0.0)):
            s, dt, kT, flags = build(lib, name)      # fresh system: every run starts from the initial context
            s.begin_phase()
            s.run(steps, dt, temp, seed=SEED, noise=noise, flags=flags)
            out[f"{name}/x_{tag}"] = s.positions()
            c = s.context()
            out[f"{name}/ctx_{tag}"] = np.array([c.step, c.time, c.bead_scale, c.bond_scale, *c.semiaxes, *c.axial_reaction])
        z = np.random.default_rng(SEED).normal(size=(5, 1, s.N, 3))
        s, dt, kT, flags = build(lib, name)
        s.begin_phase()
        s.run(5, dt, kT, noise=g.NOISE_HOST, host_noise=z, flags=flags)
        out[f"{name}/x_host5"] = s.positions()
    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
