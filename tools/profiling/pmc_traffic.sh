#!/bin/bash
# usage (on the GPU box): tools/profiling/pmc_traffic.sh <tag>  -> gpurun_out/<tag>_pmc_traffic.json
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), --kernel-trace only.
# Each pass also profiles tools/profiling/pmc_calib.py (known-byte copy + gather kernels): the per-counter correction factors
# are measured in the same session instead of assumed.
tag=$1
export TMPDIR=/tmp
STEPS=4
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c gpurun_out/pmccal_$c
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python bench.py --steps $STEPS --warmup 0 --no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --no-h2d > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmccal_$c -o p -- python tools/profiling/pmc_calib.py > gpurun_out/pmccal_$c.log 2>&1
done
python tools/profiling/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/p_results.db gpurun_out/pmc_WRITE_SIZE/p_results.db $STEPS \
  gpurun_out/pmccal_FETCH_SIZE/p_results.db gpurun_out/pmccal_WRITE_SIZE/p_results.db > gpurun_out/${tag}_pmc_traffic.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmccal_FETCH_SIZE gpurun_out/pmccal_WRITE_SIZE
head -c 2500 gpurun_out/${tag}_pmc_traffic.json
