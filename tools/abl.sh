#!/bin/bash
# section-stamp builds of libgdyn (never shipped): tools/abl.sh 30 34 ; then GDYN_LIB=libgdyn_abl30.so GDYN_STAMPS=1 python tools/ubench.py
set -e
cd "$(dirname "$0")/../2022a-genome-dynamics_amd/csrc"
for n in "$@"; do make abl N=$n; done
