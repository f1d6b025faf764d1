"""Per-call overhead of gd_run / gd_compute_energy / snapshots at the reference's observation cadence."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
g = importlib.import_module("2022a-genome-dynamics_amd")
wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(os.environ.get("GDYN_LIB"))      # developer tools only: GDYN_LIB=libgdyn_dev.so / libgdyn_ablN.so
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=R)
s.begin_phase(); s.run(2000, 1e-5, 1.0, seed=1, flags=0); s.begin_phase(); s.run(200, 1e-5, 1.0, seed=2, flags=3)
def t(f, n=1):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("run(1000)        %.2f ms" % t(lambda: s.run(1000, 1e-5, 1.0, seed=3, flags=3)))
print("10 x run(100)    %.2f ms" % t(lambda: [s.run(100, 1e-5, 1.0, seed=3, flags=3) for _ in range(10)]))
print("100 x run(10)    %.2f ms" % t(lambda: [s.run(10, 1e-5, 1.0, seed=3, flags=3) for _ in range(100)]))
print("energy()         %.3f ms" % t(lambda: s.energy(), 10))
print("context()        %.3f ms" % t(lambda: s.context(0), 10))
print("positions_f32(q) %.3f ms" % t(lambda: s.positions_f32(quantize=True), 3))
print("10 x (run(100) + energy) %.2f ms" % t(lambda: [(s.run(100, 1e-5, 1.0, seed=3, flags=3), s.energy()) for _ in range(10)]))
tm = s.timing() if hasattr(s, "timing") else None
print("--- bench sequence")
print("run(2000)        %.2f ms" % t(lambda: s.run(2000, 1e-5, 1.0, seed=9, flags=3)))
t1 = time.perf_counter()
for k in range(10):
    ta = time.perf_counter(); s.run(100, 1e-5, 1.0, seed=9, flags=3); tb = time.perf_counter(); e = s.energy(); tc = time.perf_counter()
    print("   run(100) %.2f ms  energy %.2f ms  rollbacks %d K %d" % ((tb - ta) * 1e3, (tc - tb) * 1e3, s.context(0).rollbacks, s.context(0).rebuild_interval))
snap = s.positions_f32(quantize=True)
print("obs loop total %.2f ms" % ((time.perf_counter() - t1) * 1e3))
