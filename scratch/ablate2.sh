#!/bin/bash
# one-at-a-time ablations of the spconv kernel (timing only; results are wrong by construction)
for v in "" "-DFSF_ABL_NO_CRMW" "-DFSF_ABL_NO_AREAD" "-DFSF_ABL_NO_BLOAD" "-DFSF_ABL_NO_BARRIER" "-DFSF_ABL_NO_MFMA" "-DFSF_ABL_NO_GATHER" "-DFSF_ABL_NO_CRMW -DFSF_ABL_NO_AREAD" "-DFSF_ABL_NO_CRMW -DFSF_ABL_NO_BLOAD -DFSF_ABL_NO_AREAD"; do
  echo "=== variant: [$v]"
  touch fullysparsefusion_amd/csrc/spconv.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  python scratch/profile_layers.py 10 2>&1 | grep -E "^ +(3|20|30) |total spconv"
done
touch fullysparsefusion_amd/csrc/spconv.hip; python fullysparsefusion_amd/build.py > /dev/null 2>&1
