#!/bin/bash
# tools/isa.sh <mangled-kernel-name> [extra hipcc flags]: ISA of one kernel to /tmp/kern.s, resource summary on stdout
set -e
sym=$1; shift
cd "$(dirname "$0")/../2022a-genome-dynamics_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -S --cuda-device-only "$@" -o /tmp/k.s gdyn_kernels.hip 2>/dev/null
L=$(grep -n "^$sym:" /tmp/k.s | cut -d: -f1)
awk -v L=$L 'NR>=L' /tmp/k.s | awk '/^\.Lfunc_end/{exit} {print}' > /tmp/kern.s
grep -A8 "\.name: *$sym\$" /tmp/k.s | grep "vgpr\|sgpr\|private" | tr '\n' ' '; echo; wc -l /tmp/kern.s
