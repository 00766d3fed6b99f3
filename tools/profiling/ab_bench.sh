#!/bin/bash
# usage (GPU box): tools/profiling/ab_bench.sh "<ENV=..>" "<ENV=..>" ...  -> frames/s of each environment, interleaved, twice
for rep in 1 2; do
  for e in "$@"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-roofline --no-describe --no-trained-like --steps 30 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "rep $rep  [$e]  $v"
  done
done
