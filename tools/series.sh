#!/bin/bash
# tools/series.sh: durations (us) of consecutive k_step (S) / k_fill (F) launches of the relaxed benchmark state
cd /tmp; export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --save-state /tmp/state.npy > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_series -- python3 $GRAFT_REPO_ROOT/bench.py --load-state /tmp/state.npy --warmup 100 --steps 200 --no-cpu-baseline --no-extra > /dev/null 2>&1
f=$(find /tmp/prof_series -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n.startswith("void k_step<0"): seq.append(("S", d))
    elif n.startswith("void k_fill"): seq.append(("F", d))
print(" ".join(f"{k}{d:.0f}" for k, d in seq[-90:]))
PY
