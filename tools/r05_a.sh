#!/bin/bash
# round 5, first GPU call: determinism / thread tests, then a kernel-stat profile of a short bench (cost of k_members + the ranked scatter)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05a; mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "same_seed or two_handles or forces_and_energies_vs_golden or trajectories_vs_golden or neighbor_search or softwell" > $out/tests.log 2>&1
echo "tests rc=$?" | tee -a $out/tests.log
tail -5 $out/tests.log
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python3 $root/bench.py --warmup 100 --steps 400 --no-cpu-baseline --no-extra > $out/bench_rocprof.json 2> $out/bench_rocprof.err
python3 $root/tools/kstats.py /tmp/prof_a > $out/kstats.txt; cat $out/kstats.txt
cd $root
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $out/bench_short.json 2> $out/bench_short.err; tail -c 600 $out/bench_short.json
