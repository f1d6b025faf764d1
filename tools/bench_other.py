"""Throughput of the other BASELINE.json configurations (informational lines for DESIGN.md)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(os.environ.get("GDYN_LIB"))      # developer tools only: GDYN_LIB=libgdyn_dev.so / libgdyn_ablN.so
which = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
skin = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
path = int(sys.argv[5]) if len(sys.argv) > 5 else 0
relax = int(sys.argv[6]) if len(sys.argv) > 6 else max(steps // 2, 100)      # (long enough for the skin sweep: ~5000 for the 1 kb model)
if which == "1kb":
    s, info = wl.chromatin_1kb(hip, n_beads=250000, n_replicas=R); flags = 0
elif which == "spindle":
    s, info = wl.spindle(hip, n_beads=300, n_replicas=R); flags = 0
elif which == "genome62k":
    s, info = wl.genome_interphase(hip, n_beads=62178, n_replicas=R); flags = 3
elif which == "abbox":
    s, info = wl.ab_box(hip, n_replicas=R); flags = 0
dt, kT = info["timestep"], info["temperature"]
s.set_tuning(skin=skin, kernel_path=path, auto_skin=0 if skin > 0 else int(os.environ.get("AUTO_SKIN", "1" if which == "1kb" else "0")))
# STATE_OUT=<npy>: relax (with the skin selection), save positions + "<file>.skin" and stop; STATE_IN=<npy>: start from such a
# state at the saved width (no selection): what the profiler runs, so that its averages cover the settled state only
if os.environ.get("STATE_IN"):
    s.set_tuning(skin=float(open(os.environ["STATE_IN"] + ".skin").read()), kernel_path=path)
    s.set_positions(np.load(os.environ["STATE_IN"]))
    relax = 0
s.begin_phase()
if relax:
    s.run(relax, dt, kT, seed=5, flags=0)
if os.environ.get("STATE_OUT"):
    np.save(os.environ["STATE_OUT"], s.positions())
    cut = {"1kb": 1.5}.get(which, 0.3)
    open(os.environ["STATE_OUT"] + ".skin", "w").write(repr(s.context().list_radius / cut - 1.0))
    sys.exit(0)
s.begin_phase()
s.run(100, dt, kT, seed=6, flags=flags)
t0 = time.perf_counter(); tm = s.run(steps, dt, kT, seed=7, flags=flags); el = time.perf_counter() - t0
c = s.context(); N = info["n_beads"]
print(json.dumps({"workload": info["workload"], "replicas": R, "bead_steps_per_s": N * R * steps / el, "ms_per_step": el / steps * 1e3,
                  "step_kernel_ms": tm.step_kernel_ms / tm.step_launches, "rebuild_ms_per_step": tm.rebuild_ms / tm.step_launches,
                  "list_path": c.list_path, "L_per_bead": c.list_entries / N, "list_entries_per_bead": c.list_entries / N, "list_radius": c.list_radius, "K": c.rebuild_interval, "rollbacks": c.rollbacks, "largest_tile": c.largest_tile, "tile_capacity": c.tile_capacity, "row_repairs_last_chunk": c.row_repairs, "list_GB": c.list_bytes / 1e9, "near_entries_per_bead": c.near_entries / N, "rebuilds": c.rebuilds, "E_per_bead": float(s.energy().mean() / N)}))
