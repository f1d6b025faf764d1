"""Cost of the time-integrated contact maps (gd_contacts_*) at the reference's cadence: one update every 100 steps
(contactmap_update_interval, config_entries.inc:85) of every replica of the handle, next to the per-replica gd_search_pairs +
host accumulation it replaces.  Prints one JSON line (informational)."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 128
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 30
s, info = wl.genome_interphase(hip, n_beads=N, n_replicas=R)
dt, kT = info["timestep"], info["temperature"]
flags = g.RUN_UPDATE_SCALES | g.RUN_WALL_DYNAMICS
s.begin_phase()
s.run(3000, dt, kT, seed=5, flags=0)
s.begin_phase()
s.run(300, dt, kT, seed=6, flags=flags)
dist = 0.4 * s.context(0).bead_scale
t_run = t_upd = 0.0
for k in range(updates):
    t0 = time.perf_counter(); s.run(100, dt, kT, seed=7, flags=flags); t1 = time.perf_counter()
    s.contacts_update(dist); t2 = time.perf_counter()
    t_run += t1 - t0; t_upd += t2 - t1
t0 = time.perf_counter(); rows = s.contacts(0); t_fetch = time.perf_counter() - t0
t0 = time.perf_counter(); rows = s.contacts(1); t_fetch2 = time.perf_counter() - t0
# the path it replaces: one search + download per replica (host accumulation not included)
t0 = time.perf_counter()
npairs = 0
for r in range(min(R, 16)):
    npairs += len(s.search_pairs(dist, replica=r))
t_old = (time.perf_counter() - t0) / min(R, 16) * R
print(json.dumps({"n_beads": N, "replicas": R, "updates": updates, "ms_per_100_steps": t_run / updates * 1e3,
                  "ms_per_contacts_update_all_replicas": t_upd / updates * 1e3, "pairs_per_replica_per_update": npairs / min(R, 16),
                  "distinct_rows_replica0": len(rows), "max_count": int(rows[:, 2].max()), "ms_fetch_first": t_fetch * 1e3, "ms_fetch": t_fetch2 * 1e3,
                  "ms_per_replica_search_and_download_all_replicas": t_old * 1e3, "rebuild_interval": s.context(0).rebuild_interval}))
