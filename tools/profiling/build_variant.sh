#!/bin/bash
# usage (build container): tools/profiling/build_variant.sh <name> <source.hip> <hipcc flags...>  -> ab_prev/variants/libfsf_<name>.so :
# the tree's library with ONE source recompiled under extra flags (an ablation / experiment build), for same-box A/Bs through FSF_LIB_PATH.
set -e
name=$1; src=$2; shift; shift
cs=fullysparsefusion_amd/csrc
mkdir -p ab_prev/variants
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -c $cs/$base.hip -o ab_prev/variants/${base}_$name.o
objs=$(ls $cs/build/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ab_prev/variants/libfsf_$name.so $objs ab_prev/variants/${base}_$name.o
echo ab_prev/variants/libfsf_$name.so
