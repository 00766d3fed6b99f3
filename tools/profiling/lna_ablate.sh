#!/bin/bash
# usage (GPU box): tools/profiling/lna_ablate.sh -> gpurun_out/lna_ablate.txt   (K22 with ingredients removed; timings only)
out=gpurun_out/lna_ablate.txt; : > $out
while read -r v; do
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  printf "%-64s\n" "${v:-baseline}" >> $out
  python tools/profiling/lna_bench.py 2>/dev/null | head -5 | sed 's/F.linear .*//' >> $out
done <<LIST

-DFSF_ABL_LNA_X_HOT
-DFSF_ABL_LNA_NO_W
-DFSF_ABL_LNA_NO_MFMA
-DFSF_ABL_LNA_NO_STORE
-DFSF_ABL_LNA_X_HOT -DFSF_ABL_LNA_NO_STORE
-DFSF_ABL_LNA_X_HOT -DFSF_ABL_LNA_NO_STORE -DFSF_ABL_LNA_NO_W
-DFSF_ABL_LNA_X_HOT -DFSF_ABL_LNA_NO_STORE -DFSF_ABL_LNA_NO_W -DFSF_ABL_LNA_NO_MFMA
LIST
touch fullysparsefusion_amd/csrc/linear_norm_act.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
