"""Persistent step kernel vs the one-tile-per-workgroup kernel on the same state (debugging aid)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kT = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
out = {}
for path in (2, 3):
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
    s.set_tuning(kernel_path=path, rebuild_interval=50, adapt_interval=0)
    s.begin_phase()
    s.run(steps, info["timestep"], kT, seed=7, flags=3, noise=g.NOISE_PHILOX if kT > 0 else g.NOISE_ZERO)
    out[path] = s.positions()
    print(path, "rollbacks", s.context().rollbacks, "path", s.context().list_path)
d = np.abs(out[2] - out[3]).max(axis=2)
print("max diff", d.max(), "beads > 1e-5:", int((d > 1e-5).sum()), "of", d.size)
bad = np.argwhere(d > 1e-5)
print("replicas", np.bincount(bad[:, 0], minlength=R))
if len(bad):
    print(bad[:20], d[d > 1e-5][:20])
