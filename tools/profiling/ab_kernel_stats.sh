#!/bin/bash
# usage (GPU box): tools/profiling/ab_kernel_stats.sh "<grep pattern>" [bench args...]  — rocprofv3 kernel-trace stats of the serialised frame under the
# previous build (ab_prev/, see ab_lib.sh) and the tree's, same box: the total and the table lines matching the pattern
pat=$1; shift
export TMPDIR=/tmp
FWD="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --serial $*"
for which in prev tree prev tree; do
  rm -rf gpurun_out/prof_tmp
  if [ $which = prev ]; then export FSF_LIB_PATH=$PWD/ab_prev/fullysparsefusion_amd/libfsf_hip.so; else unset FSF_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py $FWD > /dev/null 2>&1
  echo "## [$which]"
  python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db 7 x | grep -i "total kernel time\|$pat" | cut -c1-150
done
rm -rf gpurun_out/prof_tmp
