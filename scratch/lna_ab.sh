#!/bin/bash
# usage: scratch/lna_ab.sh "<flags A>" "<flags B>"  -> same-box A/B/A/B of two K22 builds
for v in "$1" "$2" "$1" "$2"; do
  echo "=== [$v]"
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  python scratch/lna_bench.py 2>&1 | grep "n=" | cut -c1-60 | sed -n '1,6p;9,10p'
done
touch fullysparsefusion_amd/csrc/linear_norm_act.hip; python fullysparsefusion_amd/build.py > /dev/null 2>&1
