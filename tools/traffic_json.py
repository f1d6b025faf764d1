"""profiles/<tag>_traffic.json from the PMC summary and the kernel statistics of tools/profile_round.sh:
HBM bytes per k_step launch = 2 x FETCH_SIZE + WRITE_SIZE (KB = 1024 B; FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 reports
half the bytes of 16-byte-per-lane streaming reads).  usage: traffic_json.py <tag> [dir]"""
import hashlib, json, os, re, sys
tag = sys.argv[1]; d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
# usage: traffic_json.py <tag> [dir [kernel [n_beads replicas [suffix [launches_per_step]]]]]  (bytes and time are per STEP)
kern = sys.argv[3] if len(sys.argv) > 3 else "k_step<0, false, true, 1, true, false>"
n_beads, replicas = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (30000, 128)
suffix = sys.argv[6] if len(sys.argv) > 6 else ""
per_step = int(sys.argv[7]) if len(sys.argv) > 7 else 1      # launches of this kernel per step (a step split by tile class is two)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256()
for f in ("gdyn_kernels.hip", "gdyn_types.h"):
    sha.update(open(os.path.join(root, "2022a-genome-dynamics_amd", "csrc", f), "rb").read())
bench = json.loads(open(f"{d}/{tag}{suffix}_bench_under_rocprof.json").read().strip().splitlines()[-1])
vals, cur = {}, None
for line in open(f"{d}/{tag}{suffix}_bench_pmc_summary.txt"):
    if not line.startswith(" "): cur = line.strip()
    elif cur == "void " + kern or cur == kern:
        m = re.match(r"\s+(\S+)\s+dispatches\s+(\d+)\s+mean/dispatch\s+([\d.]+)", line)
        if m: vals[m.group(1)] = (float(m.group(3)), int(m.group(2)))
avg_us = None
for line in open(f"{d}/{tag}{suffix}_bench_kernel_stats.txt"):
    if kern in line: avg_us = float(re.search(r"avg_us\s+([\d.]+)", line).group(1))
# the list build: every kernel of its chain, same passes (bytes per build = sum over the kernels of their mean per dispatch)
build, cur = {}, None
for line in open(f"{d}/{tag}{suffix}_bench_pmc_summary.txt"):
    if not line.startswith(" "): cur = line.strip()
    elif cur and re.match(r"(void )?k_(bin|scan|members|scatter|tiles|fill|bbox|gridp)\b", cur):
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+)\s+mean/dispatch\s+([\d.]+)", line)
        if m: build.setdefault(cur.replace("void ", "").split("(")[0], {})[m.group(1)] = float(m.group(3))
cold = lambda k: "bbox" in k or "gridp" in k or k.startswith("k_bin<false, false>")      # kernels of builds without a box from the build before
build_bytes = sum((2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 for k, v in build.items() if not cold(k))
fetch, write = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
b = (2 * fetch + write) * 1024 * per_step
if avg_us is not None: avg_us *= per_step
out = {"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing domains: tools/pmc.sh via tools/profile_round.sh) of "
                   "`bench.py --load-state <relaxed state> --warmup 200 --steps 200 --no-cpu-baseline --no-extra`, kernel " + kern + ", mean per dispatch over "
                   f"{vals['FETCH_SIZE'][1]} dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B-per-lane streaming reads); KB = 1024 B",
       "workload": {"n_beads": n_beads, "replicas_per_gpu": replicas}, "launches_per_step": per_step,
       "kernel_source_sha": sha.hexdigest()[:16], "list_entries_per_bead": bench.get("list_entries_per_bead", bench.get("config", {}).get("list_entries_per_bead")),
       "list_radius": bench.get("list_radius", bench.get("config", {}).get("list_radius")),
       "k_step": {"fetch_size_kb": fetch, "write_size_kb": write, "corrected_bytes_per_launch": b, "rocprof_avg_launch_us": avg_us,
                  "hbm_gbs": b / (avg_us * 1e-6) / 1e9, "valu_insts_per_wave": vals.get("SQ_INSTS_VALU", (0, 0))[0] / max(vals.get("SQ_WAVES", (1, 0))[0], 1),
                  "wait_fraction_of_wave_cycles": vals.get("SQ_WAIT_ANY", (0, 0))[0] / max(vals.get("SQ_WAVE_CYCLES", (1, 0))[0], 1),
                  "lds_bank_conflict_fraction": vals.get("SQ_LDS_BANK_CONFLICT", (0, 0))[0] / max(vals.get("SQ_LDS_IDX_ACTIVE", (1, 0))[0], 1)}}
out["build"] = {"corrected_bytes_per_build": build_bytes, "kernels": {k: (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024 for k, v in build.items()},
                "_comment": "2 x FETCH_SIZE + WRITE_SIZE of every kernel of the list build chain (warm builds: k_bbox / k_gridp / the cold k_bin not counted), per build"}
json.dump(out, open(f"profiles/{tag}_traffic{suffix}.json", "w"), indent=1)
print(json.dumps(out["k_step"]))
