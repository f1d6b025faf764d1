#!/bin/bash
# round 5: GPU suite (-x), section stamps of k_fill, the pipeline at 32 / 128 files
tag=${1:-r05q}
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -x -q -rs -m gpu > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $out/tests.log
[ $rc -ne 0 ] && exit $rc
GDYN_LIB=libgdyn_abl34.so GDYN_FSTAMPS=1 timeout -k 10 300 python3 tools/ubench.py 128 3000 > $out/fstamps.txt 2>&1; cat $out/fstamps.txt
echo "[p] pipeline 32" ; timeout -k 10 500 python3 tools/pipeline_scale.py 32 5000 > $out/pipeline_32.json 2> $out/pipeline_32.err; tail -c 1500 $out/pipeline_32.json
echo "[p] pipeline 128"; timeout -k 10 900 python3 tools/pipeline_scale.py 128 5000 > $out/pipeline_128.json 2> $out/pipeline_128.err; tail -c 1500 $out/pipeline_128.json
