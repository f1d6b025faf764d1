#!/usr/bin/env python3
"""tools/fine_dt_check.py [n_beads] [steps] [relax]: the displacement field of a deterministic fine-time-step run (T = 0, dt = 1e-7:
simulation_fine_sampling/simulation_driver.cc:30-34) on the device, with and without the compensated position update, against the
fp64 oracle from the same relaxed state.  One JSON line: error statistics of x(steps) - x(0) per coordinate."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")

n_beads = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
relax = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
hip = g.load(os.environ.get("GDYN_LIB") or None)
orc = g.Lib(os.path.join(ROOT, "oracle", "liboracle.so"))

s, info = wl.genome_interphase(hip, n_beads=n_beads)
s.begin_phase()
s.run(relax, info["timestep"], info["temperature"], seed=11, flags=0)
x0 = s.positions()[0]
s.close()
out = {"n_beads": n_beads, "steps": steps, "relax": relax, "max_abs_x": float(np.abs(x0).max())}
so, _ = wl.genome_interphase(orc, n_beads=n_beads)
so.set_positions(x0[None]); so.begin_phase()
out["median_abs_force_component"] = float(np.median(np.abs(so.forces())))
t0 = time.perf_counter()
so.run(steps, 1e-7, 0.0, seed=1, flags=g.RUN_WALL_DYNAMICS)
out["oracle_seconds"] = time.perf_counter() - t0
do = so.positions()[0] - x0
out["median_abs_displacement"] = float(np.median(np.abs(do)))
for tag, fl in (("compensated", g.RUN_COMPENSATED), ("auto", 0), ("uncompensated", g.RUN_UNCOMPENSATED)):
    sh, _ = wl.genome_interphase(hip, n_beads=n_beads)
    sh.set_positions(x0[None]); sh.begin_phase()
    t0 = time.perf_counter()
    sh.run(steps, 1e-7, 0.0, seed=1, flags=g.RUN_WALL_DYNAMICS | fl)
    el = time.perf_counter() - t0
    dh = sh.positions()[0] - x0
    err = np.abs(dh - do)
    moving = np.abs(do) > 1e-6
    out[tag] = {"median_err_over_median_disp": float(np.median(err) / np.median(np.abs(do))),
                "p99_err": float(np.percentile(err, 99)), "max_err": float(err.max()),
                "median_rel_err_of_moving_coords": float(np.median(err[moving] / np.abs(do[moving]))),
                "p99_rel_err_of_moving_coords": float(np.percentile(err[moving] / np.abs(do[moving]), 99)),
                "stuck_fraction_of_moving_coords": float((dh[moving] == 0).mean()), "us_per_step": el / steps * 1e6,
                "semiaxis_err": float(abs(sh.context().semiaxes[0] - so.context().semiaxes[0]))}
    sh.close()
print(json.dumps(out))
