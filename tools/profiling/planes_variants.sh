#!/bin/bash
# usage (GPU box): tools/profiling/planes_variants.sh "<hipcc flags>" ...  -> K9c timings of build variants on three layers
for v in "$@"; do
  touch fullysparsefusion_amd/csrc/spconv_planes.hip
  FSF_EXTRA_HIPCC_FLAGS="$v" python -m fullysparsefusion_amd.build > /dev/null 2>&1
  printf "%-40s " "${v:-default}"; python tools/profiling/planes_one.py 2 22 4 24 2>/dev/null
done
touch fullysparsefusion_amd/csrc/spconv_planes.hip; python -m fullysparsefusion_amd.build > /dev/null 2>&1
