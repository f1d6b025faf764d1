#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/dbg; mkdir -p $out
cd $root
GDYN_DEBUG=2 timeout -k 5 90 python3 -u tools/dbg_r05.py > $out/dbg.log 2>&1
echo "rc=$?" >> $out/dbg.log
head -60 $out/dbg.log
