#!/bin/bash
# round 5: S-genome-62k x 32 over list widths (tile class vs rebuild interval)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05_62k; mkdir -p $out
cd $root
for skin in 0 0.75 0.8 0.85 0.9; do
  timeout -k 10 200 python3 tools/bench_other.py genome62k 32 600 $skin 0 4000 >> $out/sweep.txt 2>> $out/sweep.err
done
cat $out/sweep.txt | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('%.4f  %.3e  k_step %.4f  builds %.4f  K %d  tile %d/%d  L %.1f  rb %d' % (d['list_radius'], d['bead_steps_per_s'], d['step_kernel_ms'], d['rebuild_ms_per_step'], d['K'], d['largest_tile'], d['tile_capacity'], d['L_per_bead'], d['rollbacks']))
"
