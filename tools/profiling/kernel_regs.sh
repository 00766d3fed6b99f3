#!/bin/bash
# usage: tools/profiling/kernel_regs.sh <file.hip> [kernel-substring] [extra hipcc flags...] -> VGPRs / spills / LDS / occupancy per kernel (hipcc -Rpass-analysis)
f=$1; pat=${2:-.}; shift; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kr_$$.o 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|Spill|Occupancy|LDS Size|ScratchSize" | sed 's/^.*remark: //' | paste - - - - - - - - | grep -E "$pat" | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
rm -f /tmp/kr_$$.o
