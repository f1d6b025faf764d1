"""Experiment: H handles of R replicas each on ONE GPU, stepped concurrently from H host threads (one HIP stream per handle):
does the overlap of one handle's list builds / launch tails with another handle's steps raise the aggregate throughput?
usage: two_handles.py [H] [R per handle] [steps]"""
import importlib, json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(os.environ.get("GDYN_LIB"))      # developer tools only: GDYN_LIB=libgdyn_dev.so / libgdyn_ablN.so
H = int(sys.argv[1]) if len(sys.argv) > 1 else 2
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
N = 30000
systems = []
for h in range(H):
    s, info = wl.genome_interphase(hip, n_beads=N, n_replicas=R)
    systems.append(s)
dt, kT = info["timestep"], info["temperature"]
def phase(fn):
    ts = [threading.Thread(target=fn, args=(h, s)) for h, s in enumerate(systems)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return time.perf_counter() - t0
phase(lambda h, s: (s.begin_phase(), s.run(20000, dt, kT, seed=100 + h, flags=0), s.begin_phase(), s.run(200 + 7 * h, dt, kT, seed=200 + h, flags=3)))
el = phase(lambda h, s: s.run(STEPS, dt, kT, seed=300 + h, flags=3))
print(json.dumps({"handles": H, "replicas_per_handle": R, "steps": STEPS, "ms_per_step": el / STEPS * 1e3,
                  "bead_steps_per_s": H * R * N * STEPS / el, "K": [s.context().rebuild_interval for s in systems]}))
