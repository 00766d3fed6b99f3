#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_gap
rocprofv3 --kernel-trace -d gpurun_out/prof_gap -o fsf -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-describe > gpurun_out/gap_bench.json 2> gpurun_out/gap_prof.err
python tools/profiling/gap_analysis.py gpurun_out/prof_gap/fsf_results.db 8 3 > gpurun_out/gap_analysis.txt 2>&1
rm -rf gpurun_out/prof_gap
cat gpurun_out/gap_analysis.txt
