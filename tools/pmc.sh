#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   -- separate --pmc passes (no tracing domains), summaries to gpurun_out/pmc_<tag>/
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
if [ -n "$PMC_SETS_ONLY" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS_ONLY"; else SETS=(); fi
run_set() {
  i=$((i+1))
  rocprofv3 --pmc $1 --output-format csv -d /tmp/pmc_$tag/p$i -- python3 $GRAFT_REPO_ROOT/${PMC_SCRIPT:-bench.py} "${ARGS[@]}" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
}
ARGS=("$@")
if [ ${#SETS[@]} -gt 0 ]; then for set in "${SETS[@]}"; do run_set "$set"; done; else
for set in "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  run_set "$set"
done
fi
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pmc_" + out.rsplit("pmc_", 1)[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(agg):
        fh.write(k + "\n")
        for c in sorted(agg[k]):
            s, n = agg[k][c]
            fh.write(f"   {c:36s} dispatches {n:6d}  mean/dispatch {s/n:16.1f}\n")

PY
