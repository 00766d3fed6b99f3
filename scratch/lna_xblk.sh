#!/bin/bash
# same-box A/B of K22's cross-block software pipeline
for v in "-DFSF_ABL_LNA_NO_XBLK" "" "-DFSF_ABL_LNA_NO_XBLK" ""; do
  echo "=== [$v]"
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  python scratch/lna_bench.py 2>&1 | grep "n=" | cut -c1-60 | head -6
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c60-140
done
