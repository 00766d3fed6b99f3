#!/bin/bash
# usage (GPU box): tools/profiling/lna_timeline.sh <commit> -> gpurun_out/r5/lna_timeline.txt  (rebuilds the library with -DFSF_LNA_TIMELINE, runs, rebuilds it plain)
commit=$1; out=gpurun_out/r5/lna_timeline.txt; mkdir -p gpurun_out/r5
touch fullysparsefusion_amd/csrc/linear_norm_act.hip
FSF_EXTRA_HIPCC_FLAGS="-DFSF_LNA_TIMELINE" python -m fullysparsefusion_amd.build > /dev/null 2>&1
{ echo "# commit $commit, one MI355X box ($(hostname)), $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "# library built with -DFSF_LNA_TIMELINE: s_memtime at phase boundaries of wave 0 of every workgroup (SGPR state only; the reads add an lgkmcnt(0) each)"
  echo "# python tools/profiling/lna_timeline.py"
  python tools/profiling/lna_timeline.py 2>&1 | grep -v amdgpu.ids; } > $out
touch fullysparsefusion_amd/csrc/linear_norm_act.hip
python -m fullysparsefusion_amd.build > /dev/null 2>&1
cat $out
