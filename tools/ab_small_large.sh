#!/bin/bash
# tools/ab_small_large.sh: one replica and 128 replicas of S-genome-30k (bench.py lines in short) + the idle gaps of the one-replica trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --replicas 1 --equil 4000 --warmup 500 --steps 3000 --no-extra --no-cpu-baseline 2>/dev/null > /tmp/r1.json
python3 $GRAFT_REPO_ROOT/tools/gaps.py /tmp/prof_r1 | head -8
cd $GRAFT_REPO_ROOT
pr() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 value %.4g us/step %.2f kstep %.2f rebuild %.2f K %d' % (d['value'], d['ms_per_step']*1e3, r['avg_launch_ms']*1e3, r['rebuild_ms_per_step']*1e3, d['config']['rebuild_interval']))
"; }
for i in 1 2; do python bench.py --replicas 1 --equil 4000 --warmup 500 --steps 3000 --no-extra --no-cpu-baseline 2>/dev/null | pr R1; done
python bench.py --save-state /tmp/state.npy > /dev/null 2>&1
for i in 1 2; do python bench.py --load-state /tmp/state.npy --warmup 500 --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | pr R128; done
