#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05x; mkdir -p $out
cd $root
timeout -k 10 300 python3 tools/bench_other.py 1kb 4 600 0.446 0 3000 > $out/1kb4.json 2> $out/1kb4.err; cat $out/1kb4.json | cut -c1-900
GDYN_LIB=libgdyn_dev.so GDYN_DEBUG=2 timeout -k 10 300 python3 tools/bench_other.py 1kb 4 100 0.446 0 300 > $out/1kb4d.json 2> $out/1kb4d.err; grep -c "no history" $out/1kb4d.err; grep "build" $out/1kb4d.err | tail -3
