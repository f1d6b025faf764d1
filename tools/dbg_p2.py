import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load()
orc = g.Lib(os.path.join(ROOT, "oracle", "liboracle.so"))
R, steps = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 12
so, info = wl.genome_interphase(orc, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
so.begin_phase(); so.run(steps, info["timestep"], 1.0, seed=20220101, flags=3)
xo = so.positions()
for path in (3, 0):
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R, bead_scale_init=0.9)
    s.set_tuning(kernel_path=path)
    s.begin_phase()
    s.run(steps, info["timestep"], 1.0, seed=20220101, flags=3)
    d = np.abs(s.positions() - xo).max(axis=2)
    c = s.context()
    print("path", path, "max diff vs oracle", d.max(), "n>1e-4", int((d > 1e-4).sum()), "rollbacks", c.rollbacks, "K", c.rebuild_interval, "rebuilds", c.rebuilds)
    bad = np.argwhere(d > 1e-4)
    if len(bad): print("  replicas", np.bincount(bad[:, 0], minlength=R), bad[:8].tolist())
