#!/usr/bin/env python3
"""Generates tests/golden/ref_1kb_kinetics.npz by running the scenarios of tests/kinetics_util.py through the
REFERENCE's loop simulator and reservoir sampler, compiled in place into oracle/_ref/libref1kb.so
(`make -C oracle ref`; needs /root/reference).  The fixture holds outputs only: loop records per step, the
reservoir contents, and the generator's next draw (which pins the number of random draws consumed)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kinetics_util as ku   # noqa: E402

subprocess.check_call(["make", "-C", os.path.join(ku.ROOT, "oracle"), "ref"])
ref = ku.Probe(ku.REF_LIB)
out = {}
for name, sc in ku.LOOP_SCENARIOS.items():
    loops, nxt = ref.loops(sc)
    out[f"loops_{name}"] = loops
    out[f"loops_{name}_next"] = np.uint64(nxt)
for k, (cap, n, seed) in enumerate(ku.RESERVOIR_CASES):
    items, nxt = ref.reservoir(cap, n, seed)
    out[f"reservoir_{k}"] = items
    out[f"reservoir_{k}_next"] = np.uint64(nxt)
np.savez_compressed(os.path.join(HERE, "ref_1kb_kinetics.npz"), **out)
print("wrote", len(out), "arrays")
