#!/bin/bash
# usage: tools/profiling/run_prof.sh <tag> [bench args]   (on the GPU box) -> gpurun_out/<tag>_kernels.txt
tag=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o fsf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_prof.err
python tools/profiling/prof_summary.py gpurun_out/prof_$tag/fsf_results.db 7 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-describe --no-trained-like $* ($tag)" > gpurun_out/${tag}_kernels.txt
rm -rf gpurun_out/prof_$tag
head -45 gpurun_out/${tag}_kernels.txt
