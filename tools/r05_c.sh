#!/bin/bash
# round 5: GPU suite + the PMC census of k_step
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r05c2; mkdir -p $out
cd $root
timeout -k 10 900 python3 -m pytest tests -x -q -rs -m gpu > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 bash tools/census_pmc.sh $out/kstep_census.txt
