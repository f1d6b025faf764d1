cd $GRAFT_REPO_ROOT
python bench.py --save-state /tmp/state.npy > /dev/null 2>&1
for kt in 0.85 0.9 0.95 1.0; do
  env GDYN_K_TARGET=$kt python bench.py --lib libgdyn_dev.so --load-state /tmp/state.npy --warmup 300 --steps 3000 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K_TARGET $kt', round(d['value']/1e9,3), 'ms', round(d['ms_per_step'],4), 'k_step', round(d['roofline']['avg_launch_ms'],4), 'build/step', round(d['roofline']['rebuild_ms_per_step'],4), 'K', d['config']['rebuild_interval'], 'rb', d['config']['rollbacks_in_timed_steps'])"
done
