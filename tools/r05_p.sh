#!/bin/bash
# round 5: the pipeline at 32 / 128 files (dense refined start), product library
tag=${1:-r05p}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
echo "[p] pipeline 32" ; timeout -k 10 500 python3 tools/pipeline_scale.py 32 5000 > $out/pipeline_32.json 2> $out/pipeline_32.err; tail -c 700 $out/pipeline_32.json
echo "[p] pipeline 128"; timeout -k 10 900 python3 tools/pipeline_scale.py 128 5000 > $out/pipeline_128.json 2> $out/pipeline_128.err; tail -c 1200 $out/pipeline_128.json
