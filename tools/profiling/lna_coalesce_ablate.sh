#!/bin/bash
# usage (GPU box): tools/profiling/lna_coalesce_ablate.sh -> gpurun_out/r5/lna_coalesce_ablate.txt
# What would K22 / K22s gain if their x loads / row stores were line-coalesced (consecutive lanes on consecutive 16-byte pieces)?
# The ablation builds read / write the same bytes per instruction at consecutive addresses (results are garbage: timings only).
out=gpurun_out/r5/lna_coalesce_ablate.txt; mkdir -p gpurun_out/r5; : > $out
while read -r v; do
  touch fullysparsefusion_amd/csrc/linear_norm_act.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  echo "## ${v:-baseline}" >> $out
  python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n" | grep -i "K22\|segmax" >> $out
done <<LIST

-DFSF_ABL_LNA_COAL_X
-DFSF_ABL_LNA_COAL_ST
-DFSF_ABL_LNA_COAL_X -DFSF_ABL_LNA_COAL_ST
LIST
touch fullysparsefusion_amd/csrc/linear_norm_act.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
