#!/bin/bash
# tools/census_pmc.sh <out.txt>: dynamic instruction census of k_step by sections -- SQ_INSTS_VALU / SALU / LDS / VMEM per wave of the product
# kernel and of four census builds (make -C csrc abl N=51 no pair section, 52 no bonds, 53 no noise, 54 no wall; built in-tree before
# the call), each one rocprofv3 --pmc pass (no tracing domains) of the same 100 steps from the benchmark's relaxed state at the 0.9
# width and a fixed interval of 21; a section's count = product - build.
root=$GRAFT_REPO_ROOT; res=${1:-$root/gpurun_out/kstep_census.txt}
cd /tmp; export TMPDIR=/tmp
[ -f /tmp/state.npy ] || python3 $root/bench.py --save-state /tmp/state.npy > /dev/null 2>&1
: > $res.raw
for lib in libgdyn_dev.so libgdyn_abl51.so libgdyn_abl52.so libgdyn_abl53.so libgdyn_abl54.so; do
  [ -f $root/2022a-genome-dynamics_amd/csrc/$lib ] || { echo "missing $lib"; continue; }
  rm -rf /tmp/census_p
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d /tmp/census_p -- python3 $root/bench.py --load-state /tmp/state.npy --lib $lib --skin 0.9 --interval 21 --warmup 40 --steps 100 --no-cpu-baseline --no-extra --allow-stale-traffic > /tmp/census_run.log 2>&1 || { echo "$lib: pass failed"; tail -3 /tmp/census_run.log; continue; }
  python3 - "$lib" >> $res.raw <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("/tmp/census_p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("void k_step<0, false, true, 1, true, false>"):
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
w = agg["SQ_WAVES"][0] / max(agg["SQ_WAVES"][1], 1)
print(sys.argv[1], " ".join(f"{k}={agg[k][0] / max(agg[k][1], 1) / w:.1f}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES")), f"dispatches={agg['SQ_WAVES'][1]}")
PY
done
python3 - $res <<'PY'
import sys
rows = {}
for ln in open(sys.argv[1] + ".raw"):
    p = ln.split(); rows[p[0]] = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in p[1:]}
names = {"libgdyn_abl51.so": "pair section (skin check, list walk, near + far classes as walked)", "libgdyn_abl52.so": "bonds", "libgdyn_abl53.so": "noise (Philox4x32-10, Box-Muller)", "libgdyn_abl54.so": "wall"}
full = rows.get("libgdyn_dev.so")
out = ["# k_step<STEP, open box, tiled, PK=1, S16>: instructions per WAVE (64 beads), mean over the dispatches of 100 steps at the 0.9 width, interval 21",
       "# (rocprofv3 --pmc; a section = product kernel - the census build without it; 'everything else' = prologue (record, descriptor, DMA issue),",
       "#  wave 0's context work, barrier, own position, integration, displacement bound, reductions, store)", ""]
if full:
    out.append("%-72s VALU %7.1f  SALU %7.1f  LDS %6.1f  VMEM rd %5.1f wr %5.1f  wave cycles %8.0f" % ("whole kernel", full["SQ_INSTS_VALU"], full["SQ_INSTS_SALU"], full["SQ_INSTS_LDS"], full["SQ_INSTS_VMEM_RD"], full["SQ_INSTS_VMEM_WR"], full["SQ_WAVE_CYCLES"]))
    rest = dict(full)
    for lib, nm in names.items():
        if lib in rows:
            d = {k: full[k] - rows[lib][k] for k in full if k != "dispatches"}
            for k in d: rest[k] -= d[k]
            out.append("%-72s VALU %7.1f  SALU %7.1f  LDS %6.1f  VMEM rd %5.1f wr %5.1f  wave cycles %8.0f" % (nm, d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"], d["SQ_INSTS_VMEM_RD"], d["SQ_INSTS_VMEM_WR"], d["SQ_WAVE_CYCLES"]))
    out.append("%-72s VALU %7.1f  SALU %7.1f  LDS %6.1f  VMEM rd %5.1f wr %5.1f  wave cycles %8.0f" % ("everything else", rest["SQ_INSTS_VALU"], rest["SQ_INSTS_SALU"], rest["SQ_INSTS_LDS"], rest["SQ_INSTS_VMEM_RD"], rest["SQ_INSTS_VMEM_WR"], rest["SQ_WAVE_CYCLES"]))
open(sys.argv[1], "w").write("\n".join(out) + "\n\n# raw (per wave)\n" + open(sys.argv[1] + ".raw").read())
print("\n".join(out))
PY
