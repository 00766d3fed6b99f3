#!/bin/bash
# usage (GPU box): tools/profiling/sir_input_ablate.sh -> gpurun_out/r5/sir_input_ablate.txt : what is K21 waiting for?  (timings only)
out=gpurun_out/r5/sir_input_ablate.txt; mkdir -p gpurun_out/r5; : > $out
while read -r v; do
  touch fullysparsefusion_amd/csrc/sir_input.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  echo "## ${v:-baseline}" >> $out
  python tools/profiling/sir_bench.py 2>/dev/null | tr "|" "\n" | grep -i "sir_input" >> $out
done <<LIST

-DSI_ABL_NO_STORE
-DSI_ABL_NO_SRC
-DSI_ABL_NO_MFMA3
-DSI_ABL_NO_STORE -DSI_ABL_NO_SRC
-DSI_ABL_NO_STORE -DSI_ABL_NO_SRC -DSI_ABL_NO_MFMA3
LIST
touch fullysparsefusion_amd/csrc/sir_input.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
