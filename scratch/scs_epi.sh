#!/bin/bash
# K9b epilogue variants, same box: HEAD^ kernel is not available here, so compare TB values (1 = per-tile loads)
for tb in 4 1 4 1; do
  echo "=== SCS_EPI_TB=$tb"
  touch fullysparsefusion_amd/csrc/spconv_split.hip
  FSF_EXTRA_HIPCC_FLAGS="-DSCS_EPI_TB=$tb" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  python scratch/scs_synth.py 2>&1 | grep level | sed -n '1,3p;5p;8p'
  python scratch/scs_split_layers.py 2>/dev/null | tail -1
done
touch fullysparsefusion_amd/csrc/spconv_split.hip; python fullysparsefusion_amd/build.py > /dev/null 2>&1
