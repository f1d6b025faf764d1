import faulthandler, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
faulthandler.dump_traceback_later(60, exit=True)
hip = g.load("libgdyn_dev.so")
R, n = 3, 6000
rng = np.random.default_rng(17)
x0 = ((rng.random((R, n, 3)) - 0.5) * np.array([2.0, 2.4, 2.8])).astype(np.float32).astype(np.float64)
s = g.System(hip, n, R)
s.set_bead_params(a=(np.arange(n) % 2).astype(float), b=((np.arange(n) + 1) % 2).astype(float), mobility=np.ones(n))
s.set_pair_softcore(2.0, 0.3, 2.0, 0.24)
s.set_tuning(kernel_path=2, rebuild_interval=3, adapt_interval=0)
s.set_positions(x0)
s.begin_phase()
print("set up", flush=True)
F = s.forces()
print("forces ok", float(np.abs(F).max()), s.context().list_path, flush=True)
for k in range(8):
    s.run(30, 5e-5, 2.0, seed=7, replica_seeds=[11, 12, 13])
    c = s.context()
    print("run", k, c.rebuilds, c.rollbacks, c.list_path, c.list_entries / n, flush=True)
print("done", flush=True)
