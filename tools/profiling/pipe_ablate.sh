#!/bin/bash
# usage (GPU box): tools/profiling/pipe_ablate.sh [layers...]  -> gpurun_out/pipe_ablate.txt
# K9d (spconv_fwd_pipe_kernel) with ingredients removed one at a time and cumulatively (rebuilds libfsf_hip.so per variant; the ablated
# builds compute garbage: timings only)
layers=${@:-2 22 4 10}
out=gpurun_out/pipe_ablate.txt; : > $out
while read -r v; do
  touch fullysparsefusion_amd/csrc/spconv_planes.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  printf "%-60s " "${v:-baseline}" >> $out
  python tools/profiling/planes_one.py $layers 2>/dev/null >> $out
done <<LIST

-DPD_ABL_NO_LOOP
-DPD_ABL_NO_W
-DPD_ABL_NO_G
-DPD_ABL_NO_MFMA
-DPD_ABL_NO_LDS_READ
-DPD_ABL_NO_LDS_WRITE
-DPD_ABL_NO_BARRIER
-DPD_ABL_NO_W -DPD_ABL_NO_G
-DPD_ABL_NO_W -DPD_ABL_NO_G -DPD_ABL_NO_MFMA
-DPD_ABL_NO_W -DPD_ABL_NO_G -DPD_ABL_NO_MFMA -DPD_ABL_NO_LDS_READ -DPD_ABL_NO_LDS_WRITE
-DPD_ABL_NO_W -DPD_ABL_NO_G -DPD_ABL_NO_MFMA -DPD_ABL_NO_LDS_READ -DPD_ABL_NO_LDS_WRITE -DPD_ABL_NO_BARRIER
LIST
touch fullysparsefusion_amd/csrc/spconv_planes.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
