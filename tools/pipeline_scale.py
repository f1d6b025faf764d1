"""The whole-genome pipeline as programs at the benchmark's size, timed end to end on one MI355X (informational):
    gd_prepare.py -> gd_spindle -> gd_refine.py -> gd_interphase file_1 ... file_R      (R trajectory files = R replicas of one handle)
on a synthetic genome of ~30 000 beads in 46 chains with the reference's default cadence (log / energy every 100 steps, a
quantised snapshot every 1000, a contact-map update every 100, config_entries.inc:81-86).  Prints one JSON line: the wall time of
the batched interphase program against the bead-steps it advanced, beside the stepping rate of bench.py.
usage: pipeline_scale.py [replicas [interphase_steps [n_beads]]]"""
import json, os, shutil, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "2022a-genome-dynamics_amd", "host")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
env = dict(os.environ, LD_LIBRARY_PATH=":".join([os.path.join(HOST, "h5lib"), os.path.join(ROOT, "2022a-genome-dynamics_amd", "csrc"),
                                                  os.environ.get("LD_LIBRARY_PATH", "")]))


def run(*cmd):
    t0 = time.perf_counter()
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, env=env)
    if r.returncode:
        if os.environ.get("PIPE_ERR_FILE"):      # (debugging: the whole stderr of the failed program)
            open(os.environ["PIPE_ERR_FILE"], "w").write(r.stderr)
        sys.exit(f"{cmd[0]} failed:\n{r.stderr[-3000:]}")
    return time.perf_counter() - t0, r


subprocess.check_call(["make", "-s", "-C", HOST])
tmp = tempfile.mkdtemp(prefix="gd_pipeline_")
rng = np.random.default_rng(20220101)
# 46 chains with lengths in the proportions of human chromosomes (two copies of 23), bead types in runs of 10-50
rel = np.array([249, 243, 198, 191, 181, 171, 159, 146, 141, 136, 135, 134, 115, 107, 103, 90, 81, 78, 59, 63, 48, 51, 155], float)
lens = np.maximum(5, np.round(np.tile(rel, 2) / (2 * rel.sum()) * N)).astype(int)
with open(os.path.join(tmp, "genome.tsv"), "w") as fh:
    fh.write("chain\tstart\tend\tA\tB\ttags\n")
    for c, n in enumerate(lens):
        i = 0
        while i < n:
            run_len = int(rng.integers(10, 51))
            a, b, tag = ((1.0, 0.0, "A"), (0.0, 1.0, "B"), (0.5, 0.5, "u"))[int(rng.choice(3, p=[0.4, 0.4, 0.2]))]
            for k in range(i, min(n, i + run_len)):
                t = tag + (",cen" if n // 2 - 2 <= k < n // 2 + 2 else "")
                fh.write(f"chr{c + 1}\t{k * 100000}\t{(k + 1) * 100000}\t{a}\t{b}\t{t}\n")
            i += run_len
n_beads = int(lens.sum())
radius = 0.27 * (n_beads / (8 * 0.3)) ** (1 / 3)
cfg = dict(a_core_diameter=0.30, b_core_diameter=0.24, a_core_bond_spring=70.0, a_core_bond_length=0.2, b_core_bond_spring=70.0,
           b_core_bond_length=0.2, a_core_2nd_bond_spring=5.0, b_core_2nd_bond_spring=5.0, wall_init_semiaxes=[radius] * 3,
           init_packing_radius=0.8 * radius, init_packing_spring=0.5, init_spindle_steps=2000, init_packing_steps=2000, init_sampling_interval=1000, init_logging_interval=1000,
           relaxation_steps=3000, relaxation_sampling_interval=1000, relaxation_logging_interval=1000,
           interphase_steps=steps, contactmap_thinning_rate=max(1, steps // 1000))
json.dump(cfg, open(os.path.join(tmp, "config.json"), "w"))
files, t_prep = [], {}
for r in range(R):
    f = os.path.join(tmp, f"output-{r + 1}.h5")
    t_prep["prepare"] = t_prep.get("prepare", 0) + run(sys.executable, os.path.join(HOST, "gd_prepare.py"), "--seed", 1000 + r, os.path.join(tmp, "config.json"),
                                                      os.path.join(tmp, "genome.tsv"), f)[0]
    t_prep["spindle"] = t_prep.get("spindle", 0) + run(os.path.join(HOST, "gd_spindle"), f)[0]
    t_prep["refine"] = t_prep.get("refine", 0) + run(sys.executable, os.path.join(HOST, "gd_refine.py"), f)[0]
    files.append(f)
# (GD_INTERPHASE_WRAP="rocprofv3 --kernel-trace --stats -d <dir> --": the program under the profiler, directly after the "--")
t_inter, res = run(*os.environ.get("GD_INTERPHASE_WRAP", "").split(), os.environ.get("GD_INTERPHASE_BIN") or os.path.join(HOST, "gd_interphase"), "--timing",
                   *os.environ.get("GD_INTERPHASE_ARGS", "").split(), *files)
log = [ln for ln in res.stderr.splitlines() if ln.startswith("[")]
total_steps = cfg["relaxation_steps"] + steps
size = sum(os.path.getsize(f) for f in files)
fine = None
if os.environ.get("FINE_STEPS"):      # gd_fine_sampling on the first file, restarted from the last interphase snapshot (a snapshot every 100 steps)
    t_fine, _ = run(os.path.join(HOST, "gd_fine_sampling"), "--steps", os.environ["FINE_STEPS"], files[0], 0, steps)
    fine = {"steps": int(os.environ["FINE_STEPS"]), "seconds": t_fine}
    if os.environ.get("FINE_AB"):      # (A/B against another build of the program, same restart)
        fine["seconds_" + os.path.basename(os.environ["FINE_AB"])] = run(os.environ["FINE_AB"], "--steps", os.environ["FINE_STEPS"], files[0], 0, steps)[0]
print(json.dumps({"fine_sampling": fine, "n_beads": n_beads, "replicas": R, "relaxation_steps": cfg["relaxation_steps"], "interphase_steps": steps,
                  "seconds_prepare_spindle_refine_per_file": {k: v / R for k, v in t_prep.items()},
                  "seconds_gd_interphase": t_inter, "bead_steps_per_s_end_to_end": n_beads * R * total_steps / t_inter,
                  "output_MB": size / 1e6, "timing": [ln for ln in log if ln.startswith("[timing]")], "last_log_lines": log[-3:-1]}))
shutil.rmtree(tmp)
