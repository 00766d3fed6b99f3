#!/bin/bash
# usage (GPU box): tools/profiling/planes_variants.sh "<hipcc flags>" ...  -> K9c timings of build variants on five layers
# (the default build is timed first and last: boxes drift by a few % while they warm up)
run() { touch fullysparsefusion_amd/csrc/spconv_planes.hip; FSF_EXTRA_HIPCC_FLAGS="$1" python -m fullysparsefusion_amd.build > /dev/null 2>&1
        printf "%-40s " "${1:-default}"; python tools/profiling/planes_one.py 2 22 4 24 0 2>/dev/null; }
run ""
for v in "$@"; do run "$v"; done
run ""
