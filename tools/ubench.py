"""Micro-benchmark of the build / step kernels on a fixed equilibrated snapshot (gd_debug_bench)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(os.environ.get("GDYN_LIB"))      # developer tools only: GDYN_LIB=libgdyn_dev.so / libgdyn_ablN.so
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
EQ = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
snap = f"/tmp/snap_{R}_{EQ}.npy"
if os.environ.get("UB_WORKLOAD") == "1kb":      # S-1kb-250k x R from a state of `STATE_OUT=<npy> bench_other.py 1kb R ...` (GDYN_STATE=<npy>, width from <npy>.skin)
    s, info = wl.chromatin_1kb(hip, n_beads=250000, n_replicas=R)
    s.set_tuning(skin=float(open(os.environ["GDYN_STATE"] + ".skin").read()))
else:
    s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
N_BEADS = info["n_beads"]
if os.environ.get("GDYN_STATE"):      # relaxed positions written by `bench.py --save-state` (the benchmark's own state)
    s.set_positions(np.load(os.environ["GDYN_STATE"]))
elif os.path.exists(snap):
    s.set_positions(np.load(snap))
else:
    s.begin_phase(); s.run(EQ, 1e-5, 1.0, seed=99, flags=0); np.save(snap, s.positions())
s.begin_phase()
if not os.environ.get("GDYN_NO_RUN"):      # ablation builds: no stepping with the ablated kernels, time them on the relaxed snapshot
    s.run(int(os.environ.get("GDYN_RUN_STEPS", "8")), info["timestep"], info["temperature"], seed=3, flags=0 if os.environ.get("UB_WORKLOAD") == "1kb" else 3)
c = s.context()
print(f"lib {os.environ.get('GDYN_LIB','libgdyn.so')}: build {s.debug_bench(0, 20)*1e3:.1f} us  step {s.debug_bench(1, 40)*1e3:.1f} us   L/bead {c.list_entries/N_BEADS:.1f} K {c.rebuild_interval}")
if os.environ.get("GDYN_STAMPS"):
    names = ["ctx/tail-of-prologue", "barrier", "pairs", "bonds", "bend+ps", "wall", "integrate", "perm", "loads-issue", "rec+desc", "noise", "dma-issue"]
    vals = [s.debug_bench(10 + k, 20) for k in range(12)]
    print("   cycles/wave: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, vals)) + f"  sum {sum(vals):.0f}")
if os.environ.get("GDYN_FSTAMPS"):
    names = ["stage+barrier", "remap", "rowbounds", "tests", "appends", "pad+meta", "perm+count"]
    vals = [s.debug_bench(30 + k, 10) for k in range(7)]
    print("   lane-0 test groups per wave %.1f, append iterations %.1f" % (s.debug_bench(38, 10) / 10, s.debug_bench(39, 10) / 10))
    print("   k_fill cycles/wave: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, vals)) + f"  sum {sum(vals):.0f}")
