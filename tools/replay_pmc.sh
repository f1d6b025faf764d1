#!/bin/bash
# tools/replay_pmc.sh <out>: VALU instruction counts and issue activity of the product kernel and of its two replay builds (one --pmc
# pass each, no tracing domains), to check that the arithmetic-only replay executes the product's instruction stream
out=$1; root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
[ -f /tmp/state.npy ] || python3 $root/bench.py --save-state /tmp/state.npy > /dev/null 2>&1
for lib in ${REPLAY_LIBS:-libgdyn_dev.so libgdyn_abl40.so libgdyn_abl41.so}; do
  rm -rf /tmp/rp_$lib
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d /tmp/rp_$lib -- python3 $root/tools/replay_worker.py $lib /tmp/state.npy 60 > /tmp/rp_$lib.log 2>&1
  python3 - "$lib" /tmp/rp_$lib >> $out <<'PY'
import csv, glob, sys, collections
lib, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Kernel_Name"].startswith("void k_step<0"):
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print(lib, "k_step<0,...> per dispatch: " + "  ".join(f"{k} {v[0] / max(v[1], 1):.4g}" for k, v in sorted(agg.items())))
PY
done
