#!/bin/bash
# usage (GPU box): tools/profiling/planes_ablate.sh  -> gpurun_out/planes_ablate.txt
# K9c with ingredients removed (rebuilds libfsf_hip.so per variant; the ablated builds compute garbage: timings only)
out=gpurun_out/planes_ablate.txt; : > $out
while read -r v; do
  touch fullysparsefusion_amd/csrc/spconv_planes.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  printf "%-72s " "${v:-baseline}" >> $out
  python tools/profiling/planes_one.py 2 22 4 2>/dev/null >> $out
done <<LIST

-DSP_ABL_NO_LOOP
-DSP_ABL_NO_X -DSP_ABL_NO_W -DSP_ABL_NO_MFMA
-DSP_ABL_NO_X -DSP_ABL_NO_W -DSP_ABL_NO_MFMA -DSP_ABL_NO_LDS_READ
-DSP_ABL_NO_X -DSP_ABL_NO_W -DSP_ABL_NO_MFMA -DSP_ABL_NO_LDS_READ -DSP_ABL_NO_LDS_WRITE
-DSP_ABL_NO_X -DSP_ABL_NO_W -DSP_ABL_NO_MFMA -DSP_ABL_NO_LDS_READ -DSP_ABL_NO_LDS_WRITE -DSP_ABL_NO_BARRIER
LIST
touch fullysparsefusion_amd/csrc/spconv_planes.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
