#!/usr/bin/env python3
"""tools/replay_worker.py <lib> <state.npy> [launches]: one library of the replay set (tools/replay.py) timed on the benchmark's relaxed
state -- a standalone program so that rocprofv3 --pmc can wrap it (instruction counts of the replay builds beside the product's)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd"); wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(sys.argv[1])
s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=128)
s.set_positions(np.load(sys.argv[2])); s.begin_phase()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
print(sys.argv[1], min(s.debug_bench(1, n) * 1e3 for _ in range(3)), "us per launch")
