#!/bin/bash
# round 5: steady-state cost of the headline workload over list widths with this round's build cost
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05s; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
python3 $root/bench.py --save-state /tmp/state.npy > /dev/null 2>&1
: > $out/sweep.txt
for skin in 0.9 0.95 1.0 0.9 0.95; do
  python3 $root/bench.py --load-state /tmp/state.npy --skin $skin --warmup 600 --steps 3000 --no-cpu-baseline --no-extra --allow-stale-traffic 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']; r = d['roofline']
print('skin $skin value %.3f ms %.4f k_step %.4f build/step %.4f K %d L %.1f rb %d ss %.3f' % (d['value'] / 1e9, d['ms_per_step'], r['avg_launch_ms'], r['rebuild_ms_per_step'], c['rebuild_interval'], c['list_entries_per_bead'], c['rollbacks_in_timed_steps'], c['steady_state_bead_steps_per_s'] / 1e9))" >> $out/sweep.txt
done
cat $out/sweep.txt
