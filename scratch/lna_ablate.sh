#!/bin/bash
for v in "" "-DFSF_ABL_LNA_NO_MFMA" "-DFSF_ABL_LNA_NO_STORE" "-DFSF_ABL_LNA_NO_MFMA -DFSF_ABL_LNA_NO_STORE"; do
  echo "=== [$v]"
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  python scratch/lna_bench.py 2>&1 | grep -v amdgpu.ids | grep "n= 510652 k= 256\|n= 310615 k= 128 c= 128"
done
touch fullysparsefusion_amd/csrc/linear_norm_act.hip; python fullysparsefusion_amd/build.py > /dev/null 2>&1
