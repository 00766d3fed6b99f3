#!/bin/bash
# usage (GPU box): tools/profiling/flag_ab.sh <file.hip> "<flags A>" "<flags B>" -- <command...>
# rebuilds libfsf_hip.so with each set of extra hipcc flags (FSF_EXTRA_HIPCC_FLAGS), runs the command under it, twice, interleaved; restores the plain build
src=$1; shift
sets=()
while [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
for rep in 1 2; do
  for v in "${sets[@]}"; do
    touch $src
    FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
    echo "rep $rep [${v:-plain}]"
    "$@" 2>/dev/null
  done
done
touch $src
python -m fullysparsefusion_amd.build > /dev/null 2>&1
