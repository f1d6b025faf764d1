"""Soak run from the UNRELAXED start: the dense transient (list overflow, generic fallback), the return to the tiled path,
the list-width give-back and the interval adaptation over tens of thousands of steps.  One line per 2000 steps."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(os.environ.get("GDYN_LIB"))      # developer tools only: GDYN_LIB=libgdyn_dev.so / libgdyn_ablN.so
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
total = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
dt, kT = info["timestep"], info["temperature"]
s.begin_phase()
done = 0
while done < total:
    t0 = time.perf_counter()
    s.run(2000, dt, kT, seed=11, flags=3)
    el = time.perf_counter() - t0
    done += 2000
    c = s.context()
    x = s.positions_f32()
    print(f"step {done:6d}  {30000 * R * 2000 / el / 1e9:6.2f} G bead-steps/s  path {c.list_path}  K {c.rebuild_interval:3d}  L {c.list_entries / 30000:5.1f}  "
          f"rollbacks {c.rollbacks}  lists {c.list_bytes / 1e9:.2f} GB  repairs {c.row_repairs}  R_wall {c.semiaxes[0]:.4f}  E/bead {s.energy()[0] / 30000:.4f}  finite {bool(np.isfinite(x).all())}", flush=True)
