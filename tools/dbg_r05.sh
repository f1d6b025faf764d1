#!/bin/bash
# debug: the 128-file pipeline with a developer-library build of gd_interphase (host prints of the library on stderr)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/dbg; mkdir -p $out/bin
cd $root
make -s -C 2022a-genome-dynamics_amd/host OUTDIR=$out/bin GDYN_LIB=gdyn_dev $out/bin/gd_interphase > $out/make.log 2>&1 || { tail -5 $out/make.log; exit 1; }
GDYN_DEBUG=1 GD_INTERPHASE_BIN=$out/bin/gd_interphase PIPE_ERR_FILE=$out/interphase.err timeout -k 10 900 python3 tools/pipeline_scale.py 128 5000 > $out/pipeline_128.json 2> $out/pipeline_128.err
echo "rc=$?"
grep "^\[gdyn\]" $out/interphase.err | tail -60 > $out/gdyn_tail.txt; grep -c "^\[gdyn\]" $out/interphase.err; cat $out/gdyn_tail.txt | cut -c1-220
rm -rf $out/bin
