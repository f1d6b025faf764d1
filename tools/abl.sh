#!/bin/bash
# timing-only ablation builds of libgdyn (never shipped): tools/abl.sh 21 22 ... ; then GDYN_LIB=libgdyn_ablN.so python tools/ubench.py
set -e
cd "$(dirname "$0")/../2022a-genome-dynamics_amd/csrc"
for n in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DGD_ABL=$n -c gdyn_kernels.hip -o /tmp/gdyn_kernels_abl$n.o
  hipcc -shared -fPIC --offload-arch=gfx950 -o libgdyn_abl$n.so /tmp/gdyn_kernels_abl$n.o gdyn_capi.o
done
