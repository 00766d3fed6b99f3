#!/bin/bash
# usage: tools/profiling/pmc_kernel.sh <kernel-substring> <counters...> -- <cmd...>   (prints avg counter values for the kernel)
pat=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
export TMPDIR=/tmp
rm -rf gpurun_out/pmck
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d gpurun_out/pmck -o p -- "$@" > gpurun_out/pmck.log 2>&1
python - "$pat" <<'PY'
import sqlite3, sys
cur = sqlite3.connect('gpurun_out/pmck/p_results.db').cursor()
acc = {}
for name, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like ?", ('%' + sys.argv[1] + '%',)):
    a = acc.setdefault(cname, [0, 0.0]); a[0] += 1; a[1] += val
for k, (n, v) in sorted(acc.items()):
    print(f'{k:32s} launches {n:4d}  avg {v / n:16.1f}')
PY
rm -rf gpurun_out/pmck
