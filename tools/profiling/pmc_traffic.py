"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, TCC has 4 slots) per kernel.

usage: pmc_traffic.py <fetch_results.db> <write_results.db> <steps_profiled> > profiles/rN_pmc_traffic.json
Units/corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are KiB derived from the L2's
fabric-side request counters; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so wide coalesced reads
(what these kernels issue: 16 B/lane global_load / global_load_lds) are doubled.  WRITE_SIZE is taken as reported.
"""
import json, sqlite3, sys, re
from collections import defaultdict


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    acc = defaultdict(lambda: [0, 0.0])
    for name, value in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        a = acc[name]
        a[0] += 1
        a[1] += value
    return acc


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
rows = []
for name in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(name, [0, 0.0])
    nw, w = write.get(name, [0, 0.0])
    n = max(nf, nw)
    rows.append(dict(kernel=short(name), launches=n, launches_per_step=round(n / steps, 2),
                     fetch_kib_raw_per_launch=round(f / max(nf, 1), 2), write_kib_per_launch=round(w / max(nw, 1), 2),
                     hbm_bytes_per_launch=round((2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0, 0),
                     hbm_mb_per_step=round((2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0 * n / steps / 1e6, 3)))
rows.sort(key=lambda r: -r["hbm_mb_per_step"])
spconv = [r for r in rows if "spconv_fwd" in r["kernel"] or "spconv_reduce" in r["kernel"]]
api_launches = sum(r["launches_per_step"] for r in spconv if "spconv_fwd" in r["kernel"])
total_mb = sum(r["hbm_mb_per_step"] for r in spconv)
print(json.dumps(dict(
    how="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction)",
    steps_profiled=steps,
    spconv_forward=dict(api_launches_per_step=api_launches, hbm_mb_per_step=round(total_mb, 2),
                        hbm_bytes_per_api_launch=round(total_mb * 1e6 / max(api_launches, 1), 0)),
    frame_total_hbm_mb_per_step=round(sum(r["hbm_mb_per_step"] for r in rows), 1),
    kernels=rows[:40]), indent=1))
