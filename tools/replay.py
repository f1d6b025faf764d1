#!/usr/bin/env python3
"""tools/replay.py <out.json>: what bounds k_step -- the two replay kernels of VERDICT r03 item 5 on the benchmark's own relaxed state
(S-genome-30k x 128, fresh list, same grid and LDS class for all three):
    t_kstep  the product kernel (developer library, identical kernel code)
    t_mem    its memory pattern with the arithmetic stripped   (libgdyn_abl40.so: records, tile DMA, list + adjacency chunks, LDS gathers, store)
    t_alu    its arithmetic with the operands resident          (libgdyn_abl41.so: noise, pair / bond / wall arithmetic on registers; no DMA, chunks, gathers)
and the achieved overlap (t_mem + t_alu - t_kstep) / min(t_mem, t_alu): 1 = the shorter side is completely hidden behind the longer one,
0 = the two simply add up.  Each library runs in a process of its own (gd_debug_bench, 200 back-to-back launches, three repeats)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import importlib, os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
g = importlib.import_module("2022a-genome-dynamics_amd"); wl = importlib.import_module("2022a-genome-dynamics_amd.workloads")
hip = g.load(sys.argv[2])
s, info = wl.genome_interphase(hip, n_beads=30000, n_replicas=128)
s.set_positions(np.load(sys.argv[3])); s.begin_phase()
ts = [s.debug_bench(1, 200) * 1e3 for _ in range(3)]
c = s.context()
print(min(ts), c.list_entries / 30000, c.tile_capacity)
'''


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    state = "/tmp/state.npy"
    if not os.path.exists(state):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "bench.py"), "--save-state", state], stdout=subprocess.DEVNULL)
    res = {}
    for tag, lib in (("t_kstep_us", "libgdyn_dev.so"), ("t_mem_us", "libgdyn_abl40.so"), ("t_alu_us", "libgdyn_abl41.so")):
        r = subprocess.run([sys.executable, "-c", WORKER, ROOT, lib, state], capture_output=True, text=True, check=True)
        t, L, cap = r.stdout.split()
        res[tag] = float(t); res["list_entries_per_bead"] = float(L); res["tile_capacity"] = int(cap)
    tk, tm, ta = res["t_kstep_us"], res["t_mem_us"], res["t_alu_us"]
    res["overlap"] = (tm + ta - tk) / min(tm, ta)
    res["ceiling_if_fully_overlapped_us"] = max(tm, ta)
    import hashlib
    sha = hashlib.sha256()
    for f in ("gdyn_kernels.hip", "gdyn_types.h"):
        sha.update(open(os.path.join(ROOT, "2022a-genome-dynamics_amd", "csrc", f), "rb").read())
    res["kernel_source_sha"] = sha.hexdigest()[:16]
    res["workload"] = {"n_beads": 30000, "replicas_per_gpu": 128}
    res["note"] = ("k_step on a fresh list (near class only), 3.84 M beads per launch; t_mem / t_alu are the replay builds -DGD_ABL=40 / 41 of the "
                   "same source (gdyn_kernels.hip, GD_REPLAY hooks)")
    open(out, "w").write(json.dumps(res, indent=1) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
