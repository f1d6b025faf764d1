#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/dbg; mkdir -p $out
cd $root
GDYN_DEBUG=2 GDYN_TEST_LIB=libgdyn_dev.so timeout -k 5 120 python3 -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "$1" > $out/dbg.log 2>&1
echo "rc=$?" >> $out/dbg.log
grep -v "^\[gdyn\] build" $out/dbg.log | head -60
grep "^\[gdyn\] build" $out/dbg.log | head -12
