#!/bin/bash
# tools/skin_sweep.sh <out> <skin>...: steady-state cost of the headline workload over list widths (developer library, relaxed state from
# bench.py --save-state): value over 2000 steps, k_step, builds per step, interval, largest tile (GDYN_DEBUG lines), one line per skin
out=$1; shift
cd "$(dirname "$0")/.."
python bench.py --save-state /tmp/state.npy > /dev/null 2>&1
for sk in "$@"; do
  GDYN_DEBUG=1 GDYN_SKIN=$sk python bench.py --lib libgdyn_dev.so --load-state /tmp/state.npy --warmup 600 --steps 2000 --no-cpu-baseline --no-extra 2> /tmp/sweep.err | python -c "
import json,sys,re
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=[int(m.group(1)) for m in re.finditer(r'largest tile (\d+)', open('/tmp/sweep.err').read())]
print('skin $sk', 'value', round(d['value']/1e9,3), 'ms', round(d['ms_per_step'],4), 'k_step', round(d['roofline']['avg_launch_ms'],4), 'build/step', round(d['roofline']['rebuild_ms_per_step'],4), 'K', d['config']['rebuild_interval'], 'L', round(d['config']['list_entries_per_bead'],1), 'rb', d['config']['rollbacks_in_timed_steps'], 'ss', round(d['config']['steady_state_bead_steps_per_s']/1e9,3), 'tiles', t[-3:])" >> $out
  grep -c "overflow" /tmp/sweep.err >> $out
done
