#!/bin/bash
# usage (GPU box): tools/profiling/ab_env_kernel_stats.sh <ENV_VAR> "<grep pattern>" [bench args...] — rocprofv3 kernel-trace stats of the serialised
# frame with ENV_VAR=0 and =1, interleaved twice on the same box: total kernel time per frame and the table lines matching the pattern
var=$1; pat=$2; shift; shift
export TMPDIR=/tmp
FWD="--steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --serial $*"
for v in 0 1 0 1; do
  rm -rf gpurun_out/prof_tmp
  env $var=$v rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tmp -o fsf -- python bench.py $FWD > /dev/null 2>&1
  echo "## [$var=$v]"
  python tools/profiling/prof_summary.py gpurun_out/prof_tmp/fsf_results.db 7 x | grep -i "total kernel time\|$pat" | cut -c1-150
done
rm -rf gpurun_out/prof_tmp
