#!/bin/bash
# segmented-reduce chunk length, same box: kernel sums of the seg_* kernels + frame time
for c in 32 64 128 32 64; do
  echo "=== SEG_CHUNK_ROWS=$c"
  touch fullysparsefusion_amd/csrc/segment.hip
  FSF_EXTRA_HIPCC_FLAGS="-DSEG_CHUNK_ROWS=$c" python fullysparsefusion_amd/build.py > /dev/null 2>&1 || echo BUILD FAILED
  bash scratch/run_prof.sh segab > /dev/null 2>&1
  grep "seg_" gpurun_out/segab_kernels.txt | awk '{s+=$2; print $2, $3, $NF}' | tr '\n' ';' | cut -c1-400; echo
  grep "seg_" gpurun_out/segab_kernels.txt | awk '{s+=$2} END {print "seg total ms/frame", s}'
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c60-140
done
touch fullysparsefusion_amd/csrc/segment.hip; python fullysparsefusion_amd/build.py > /dev/null 2>&1
