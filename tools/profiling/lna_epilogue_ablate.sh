#!/bin/bash
# usage (GPU box): tools/profiling/lna_epilogue_ablate.sh -> gpurun_out/r5/lna_epilogue_ablate.txt
# What is K22's epilogue tile loop (57-61 % of a wave's time, profiles/r5_lna_timeline.txt) waiting for?  Ablation builds (results are
# garbage: timings only): no row stores; no LDS reads of gamma / beta inside the loop; both.
out=gpurun_out/r5/lna_epilogue_ablate.txt; mkdir -p gpurun_out/r5; : > $out
while read -r v; do
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  echo "## ${v:-baseline}" >> $out
  python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n" | grep -i "K22\|segmax" >> $out
done <<LIST

-DFSF_ABL_LNA_NO_STORE
-DFSF_ABL_LNA_NO_VEC
-DFSF_ABL_LNA_NO_STORE -DFSF_ABL_LNA_NO_VEC
-DFSF_ABL_LNA_NO_MFMA
LIST
touch fullysparsefusion_amd/csrc/linear_norm_act.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
