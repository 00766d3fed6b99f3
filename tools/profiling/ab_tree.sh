#!/bin/bash
# usage (build container): tools/profiling/ab_tree.sh <commit>  -> exports that commit's package + header into ab_prev/ and builds its
# library there, so that a GPU call can A/B two SOURCE states on one box:
#   FSF_ROOT=ab_prev python tools/profiling/sir_bench.py ; python tools/profiling/sir_bench.py
# (ab_prev/ is git-ignored; it travels with the gpurun snapshot like the built .so files)
set -e
c=${1:-HEAD}
rm -rf ab_prev && mkdir -p ab_prev
git archive "$c" fullysparsefusion_amd include configs bench.py | tar -x -C ab_prev
(cd ab_prev && python -c "from fullysparsefusion_amd import build; print(build.build())")
