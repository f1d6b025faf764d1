cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/exp3_tests.log 2>&1; echo rc=$?; tail -2 gpurun_out/exp3_tests.log
python bench.py --save-state /tmp/state.npy > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --load-state /tmp/state.npy --warmup 200 --steps 600 --no-cpu-baseline --no-extra > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py /tmp/prof | head -12
