#!/bin/bash
# usage (GPU box): tools/profiling/ab_lib.sh <reps> <command...>   — same-box A/B of two builds of the library: the tree's
# (fullysparsefusion_amd/libfsf_hip.so) against a previous commit's, built into ab_prev/ beforehand
# (`git archive <commit> fullysparsefusion_amd include | tar -x -C ab_prev; cd ab_prev; python -m fullysparsefusion_amd.build`); interleaved.
reps=$1; shift
for rep in $(seq $reps); do
  echo "## rep $rep [prev]"; FSF_LIB_PATH=$PWD/ab_prev/fullysparsefusion_amd/libfsf_hip.so "$@" 2>/dev/null
  echo "## rep $rep [tree]"; "$@" 2>/dev/null
done
