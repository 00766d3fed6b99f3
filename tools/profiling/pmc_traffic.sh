#!/bin/bash
# usage (on the GPU box): tools/profiling/pmc_traffic.sh <tag>  -> gpurun_out/<tag>_pmc_traffic.json
tag=$1
export TMPDIR=/tmp
STEPS=3  # bench.py runs one extra untimed pass to count the queries (describe_output): STEPS + 1 forwards are profiled
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python bench.py --steps $STEPS --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err
done
python tools/profiling/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/p_results.db gpurun_out/pmc_WRITE_SIZE/p_results.db $((STEPS + 1)) > gpurun_out/${tag}_pmc_traffic.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -c 3000 gpurun_out/${tag}_pmc_traffic.json
