set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r03_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r03_gpu_tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r03_bench_2000steps.json 2> gpurun_out/r03_bench_2000steps.err
python3 -c "
import json
for f in ('gpurun_out/r03_bench.json','gpurun_out/r03_bench_2000steps.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_kind'], d['config'].get('steady_state_bead_steps_per_s'), d['config'].get('bead_steps_per_s_with_reference_cadence_rank0'))
d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['config'].get('other_workloads'), indent=0)[:3000])
"
