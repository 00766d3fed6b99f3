#!/bin/bash
# K22 / K9b with and without their MFMAs (operand loads, splits, LDS traffic and barriers unchanged), same box
for v in "" "-DFSF_ABL_LNA_NO_MFMA -DFSF_ABL_SCS_NO_MFMA" ""; do
  echo "=== [$v]"
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip fullysparsefusion_amd/csrc/spconv_split.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  python scratch/lna_bench.py 2>&1 | grep "n=" | cut -c1-60 | head -6
  python scratch/scs_synth.py 2>&1 | grep level
done
touch fullysparsefusion_amd/csrc/linear_norm_act.hip fullysparsefusion_amd/csrc/spconv_split.hip; python fullysparsefusion_amd/build.py > /dev/null 2>&1
