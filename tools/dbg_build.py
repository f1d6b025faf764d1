import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
s.begin_phase()
EQ = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
if EQ: s.run(EQ, 1e-5, 1.0, seed=99, flags=0); s.begin_phase()
for i in range(6):
    t0 = time.time(); tm = s.run(50, 1e-5, 1.0, seed=3, flags=3); el = time.time() - t0
    c = s.context()
    print(f"run{i}: wall {el*1e3:.1f} ms step_ms/launch {tm.step_kernel_ms/max(tm.step_launches,1):.4f} rebuild_ms/build {tm.rebuild_ms/max(tm.rebuild_launches,1):.4f} builds {tm.rebuild_launches} K {c.rebuild_interval} L/bead {c.list_entries/30000:.1f} rollbacks {c.rollbacks}")
