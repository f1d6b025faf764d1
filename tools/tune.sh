#!/bin/bash
# tools/tune.sh: bench.py (relaxed state, developer library) over near-class fractions / skins; one JSON summary line each
cd "$(dirname "$0")/.."
python bench.py --save-state /tmp/state.npy > /dev/null 2>&1
for cfg in "$@"; do
  nf=${cfg%%:*}; sk=${cfg##*:}
  GDYN_NEAR_FRAC=$nf GDYN_SKIN=$sk python bench.py --lib libgdyn_dev.so --load-state /tmp/state.npy --warmup 300 --steps 1500 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('near_frac $nf skin $sk', round(d['value']/1e9,3), 'ms', round(d['ms_per_step'],4), 'k_step', round(d['roofline']['avg_launch_ms'],4), 'build/step', round(d['roofline']['rebuild_ms_per_step'],4), 'K', d['config']['rebuild_interval'], 'L', round(d['config']['list_entries_per_bead'],1), 'rb', d['config']['rollbacks_in_timed_steps'])"
done
