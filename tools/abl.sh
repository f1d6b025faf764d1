#!/bin/bash
# timing-only ablation builds of libgdyn (never shipped): tools/abl.sh 21 22 ... ; then GDYN_LIB=libgdyn_ablN.so python tools/ubench.py
set -e
cd "$(dirname "$0")/../2022a-genome-dynamics_amd/csrc"
for n in "$@"; do make abl N=$n; done
