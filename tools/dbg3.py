import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import build, g
hip = g.load()
SEED = 20220101
for rep in range(2):
    for tag, steps, noise, temp in (("philox1", 1, g.NOISE_PHILOX, 1.0), ("philox10", 10, g.NOISE_PHILOX, 1.0), ("zero20", 20, g.NOISE_ZERO, 0.0)):
        s, dt, kT, flags = build(hip, "genome")
        s.set_tuning(kernel_path=2)
        s.begin_phase()
        s.run(steps, dt, temp, seed=SEED, noise=noise, flags=flags)
        x = s.positions()
        c = s.context()
        print(tag, "nan", int(np.isnan(x).sum()), "path", c.list_path, "L", round(c.list_entries / s.N, 1), "rollbacks", c.rollbacks, "rebuilds", c.rebuilds, "K", c.rebuild_interval, flush=True)
