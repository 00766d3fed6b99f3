#!/bin/bash
# usage (GPU box): tools/profiling/ab_trees.sh [reps] [extra bench args]  -> frames/s of ab_prev/ (tools/profiling/ab_tree.sh <commit>) against the
# working tree, interleaved, 10-sweep and single-sweep frame
reps=${1:-2}; shift
B="--no-cpu-baseline --no-roofline --no-describe --no-trained-like --no-train-block --steps 30 --warmup 6"
get() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in $(seq $reps); do
  for sw in 10 1; do
    a=$(cd ab_prev && python bench.py $B --sweeps $sw "$@" 2>/dev/null | get)
    b=$(python bench.py $B --sweeps $sw "$@" 2>/dev/null | get)
    echo "rep $rep  ${sw}-sweep  [prev] $a   [tree] $b"
  done
done
