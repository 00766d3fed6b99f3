#!/bin/bash
# usage: tools/profiling/prof_any.sh <tag> <cmd...>  -> per-kernel table of an arbitrary command
tag=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o fsf -- "$@" > gpurun_out/${tag}_out.txt 2> gpurun_out/${tag}_prof.err
python tools/profiling/prof_summary.py gpurun_out/prof_$tag/fsf_results.db 1 "$*" > gpurun_out/${tag}_kernels.txt
rm -rf gpurun_out/prof_$tag
