"""Single-replica latency of the S-genome-30k step (what one reference-shaped driver process runs)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
path = int(sys.argv[2]) if len(sys.argv) > 2 else 0
s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
if path: s.set_tuning(kernel_path=path)
dt, kT = info["timestep"], info["temperature"]
s.begin_phase(); s.run(4000, dt, kT, seed=5, flags=0); s.begin_phase(); s.run(300, dt, kT, seed=6, flags=3)
t0 = time.perf_counter(); tm = s.run(2000, dt, kT, seed=7, flags=3); el = time.perf_counter() - t0
c = s.context()
print(json.dumps({"R": R, "path": path, "us_per_step": el / 2000 * 1e6, "k_step_us": tm.step_kernel_ms / tm.step_launches * 1e3,
                  "rebuild_us_per_step": tm.rebuild_ms / tm.step_launches * 1e3, "K": c.rebuild_interval, "L": c.list_entries / 30000 / R}))
