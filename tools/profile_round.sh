#!/bin/bash
# tools/profile_round.sh <tag>: the round's evidence in one GPU call -- relaxed state, rocprofv3 kernel stats, PMC passes
# (separate, no tracing domains), plain bench lines; then the same for S-1kb-250k x 16.  Everything lands in gpurun_out/<tag>_*;
# copy what is judged to profiles/ (tools/traffic_json.py writes profiles/<tag>_traffic*.json).
tag=${1:-r04}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp; export TMPDIR=/tmp
python3 $root/bench.py --save-state /tmp/state.npy > /dev/null 2>&1
echo "[profile] state saved"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $root/bench.py --load-state /tmp/state.npy --warmup 300 --steps 1000 --no-cpu-baseline --no-extra --allow-stale-traffic > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_bench_under_rocprof.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
python3 $root/tools/kstats.py /tmp/prof_stats > $out/${tag}_bench_kernel_stats.txt
echo "[profile] kernel stats done"
PMC_SCRIPT=bench.py bash $root/tools/pmc.sh $tag --load-state /tmp/state.npy --warmup 200 --steps 200 --no-cpu-baseline --no-extra --allow-stale-traffic
cp $out/pmc_$tag/summary.txt $out/${tag}_bench_pmc_summary.txt
echo "[profile] pmc done"
# ---- S-1kb-250k x 16 (periodic, tiled rows): relaxed with the skin selection first (not profiled), then kernel stats + the
# traffic counters of the settled state at the selected width
STATE_OUT=/tmp/state1kb.npy python3 $root/tools/bench_other.py 1kb 16 600 0 0 9000 > /dev/null 2>&1
export STATE_IN=/tmp/state1kb.npy
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats_1kb -- python3 $root/tools/bench_other.py 1kb 16 600 > $out/${tag}_1kb_bench_under_rocprof.json 2> $out/${tag}_1kb_bench_under_rocprof.err
cp $(find /tmp/prof_stats_1kb -name "*kernel_stats.csv" | head -1) $out/${tag}_1kb_bench_kernel_stats.csv
python3 $root/tools/kstats.py /tmp/prof_stats_1kb > $out/${tag}_1kb_bench_kernel_stats.txt
PMC_SETS_ONLY="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" PMC_SCRIPT=tools/bench_other.py bash $root/tools/pmc.sh ${tag}_1kb 1kb 16 600
cp $out/pmc_${tag}_1kb/summary.txt $out/${tag}_1kb_bench_pmc_summary.txt
unset STATE_IN
echo "[profile] 1kb done"
cd $root
# ---- what bounds k_step: replay builds (memory pattern only / arithmetic only), section stamps, VALU issue rates (developer builds,
# built in-tree before the call: make -C csrc dev abl N=30 && make abl N=40 && make abl N=41 && make abl N=43; tools/ubench/valu_rate)
if [ -f 2022a-genome-dynamics_amd/csrc/libgdyn_abl41.so ]; then
  python3 tools/replay.py $out/${tag}_replay.json > /dev/null 2> $out/${tag}_replay.err
  rm -f $out/${tag}_sections.txt; bash tools/sections.sh $out/${tag}_sections.txt
  [ -x tools/ubench/valu_rate ] && ./tools/ubench/valu_rate > $out/${tag}_valu_rate.txt 2>&1
  echo "[profile] replay done"
fi
python3 bench.py --gpus 1 --steps 20 --warmup 5 --allow-stale-traffic > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "[profile] driver-line bench done"
python3 bench.py --no-cpu-baseline --no-extra --allow-stale-traffic > $out/${tag}_bench_2000steps.json 2> $out/${tag}_bench_2000steps.err
tail -c 300 $out/${tag}_bench_2000steps.json
