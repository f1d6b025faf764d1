#!/bin/bash
# round 5: full GPU suite, then kernel stats of the steady state from a saved relaxed state (clean averages: no relaxation launches)
tag=${1:-r05b}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -x -q -rs -m gpu > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $out/tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 python3 $root/bench.py --save-state /tmp/state.npy > /dev/null 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python3 $root/bench.py --load-state /tmp/state.npy --warmup 300 --steps 1000 --no-cpu-baseline --no-extra --allow-stale-traffic > $out/bench_rocprof.json 2> $out/bench_rocprof.err || exit 1
python3 $root/tools/kstats.py /tmp/prof_a > $out/kstats.txt; cat $out/kstats.txt
cp $(find /tmp/prof_a -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cd $root
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --allow-stale-traffic > $out/bench_short.json 2> $out/bench_short.err; tail -c 900 $out/bench_short.json
