import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
g = importlib.import_module("2022a-genome-dynamics_amd"); wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load("libgdyn_dev.so")
s, info = wl.chromatin_1kb(hip, n_beads=250000)
for sk in (0.5, 0.35):
    s.set_tuning(skin=sk, kernel_path=2)
    f = s.forces()
    print("skin", sk, "path", s.context().list_path, flush=True)
