#!/bin/bash
# round 5: per-kernel split of the pipeline's batched interphase program (128 files) under rocprofv3 --kernel-trace --stats
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05pk; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
GD_INTERPHASE_WRAP="rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe --" timeout -k 10 1000 python3 $root/tools/pipeline_scale.py 128 5000 > $out/pipeline_128_under_rocprof.json 2> $out/pipeline_128_under_rocprof.err
python3 $root/tools/kstats.py /tmp/prof_pipe > $out/pipeline_kernel_stats.txt; head -16 $out/pipeline_kernel_stats.txt
