#!/bin/bash
# round 5: S-genome-62k x 32 / x 64 relaxed as long as the headline state (20 000 steps) instead of 4 000
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_62k; mkdir -p $out
cd $root
: > $out/sweepX.txt
for spec in "32 0" "32 0.75" "32 0.9" "64 0"; do
  set -- $spec
  timeout -k 10 300 python3 tools/bench_other.py genome62k $1 600 $2 0 20000 >> $out/sweepX.txt 2>> $out/sweep.err
done
cat $out/sweepX.txt | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('R %d  %.4f  %.3e  k_step %.4f  builds %.4f  K %d  tile %d/%d  L %.1f  rb %d' % (d['replicas'], d['list_radius'], d['bead_steps_per_s'], d['step_kernel_ms'], d['rebuild_ms_per_step'], d['K'], d['largest_tile'], d['tile_capacity'], d['L_per_bead'], d['rollbacks']))
"
