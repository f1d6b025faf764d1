#!/bin/bash
# tools/sections.sh <out>: where k_step's time goes, per section of a wave (s_memtime stamps, gdyn_stamps.h) on the benchmark's relaxed
# state x 128: the product kernel (abl30) and the ALU replay (abl43: arithmetic only -- its sections are VALU issue time)
out=$1
cd "$(dirname "$0")/.."
[ -f /tmp/state.npy ] || python bench.py --save-state /tmp/state.npy > /dev/null 2>&1
for lib in libgdyn_abl30.so libgdyn_abl43.so; do
  GDYN_STATE=/tmp/state.npy GDYN_NO_RUN=1 GDYN_STAMPS=1 GDYN_LIB=$lib python tools/ubench.py 128 >> $out 2>&1
done
