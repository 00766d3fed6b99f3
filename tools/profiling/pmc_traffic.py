"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, TCC has 4 slots) per kernel.

usage: pmc_traffic.py <fetch.db> <write.db> <steps_profiled> [<calib_fetch.db> <calib_write.db>] > profiles/rN_pmc_traffic.json

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md "HBM"): both counters are KiB derived from the L2's fabric-side
request counters (Infinity-Cache hits included).  On gfx950 FETCH_SIZE tallies 128-B read requests at 64 B for wide coalesced
reads — the guide's factor 2 — while other access patterns and WRITE_SIZE are uncalibrated.  When the calibration passes
(tools/profiling/pmc_calib.py: a 1 GiB copy and a 1 GiB gather of 512-byte rows, known bytes) are given, the factors are
MEASURED in the same session: streaming factor from the copy kernel, gather factor from the row gather; every kernel gets the
streaming factor except the ones listed in GATHER_KERNELS (whose reads are row gathers), and both raw and corrected values are
reported.  Without calibration the guide's 2.0 / 1.0 are used for every kernel (an upper bound for gather-type kernels).
"""
import json, sqlite3, sys, re
from collections import defaultdict

GATHER_KERNELS = ("spconv_fwd", "gather_rows", "seg_reduce", "cam_select", "project_gather", "project_score", "voxel2point",
                  "index", "gather", "rs_onesweep")


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


def calibrate(fetch_db, write_db):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    gib = float(1 << 30)
    out = {}
    for kind, pat, read_b, write_b in (("stream", "copy", gib, gib), ("gather", "gather", gib + 8 * (1 << 20), gib)):
        fk = [(n, v) for n, v in f.items() if pat in n.lower() and v[0] >= 3 and v[1] / v[0] > 1e5]
        wk = [(n, v) for n, v in w.items() if pat in n.lower() and v[0] >= 3 and v[1] / v[0] > 1e5]
        if fk and wk:
            fn, fv = max(fk, key=lambda t: t[1][1])
            wn, wv = max(wk, key=lambda t: t[1][1])
            out[kind] = dict(kernel=short(fn), fetch_kib_raw_per_launch=round(fv[1] / fv[0], 1), write_kib_raw_per_launch=round(wv[1] / wv[0], 1),
                             known_read_bytes=read_b, known_write_bytes=write_b,
                             fetch_factor=round(read_b / (fv[1] / fv[0] * 1024.0), 4), write_factor=round(write_b / (wv[1] / wv[0] * 1024.0), 4))
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
cal = calibrate(sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else {}
f_stream = cal.get("stream", {}).get("fetch_factor", 2.0)
w_stream = cal.get("stream", {}).get("write_factor", 1.0)
f_gather = cal.get("gather", {}).get("fetch_factor", f_stream)
rows = []
for name in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(name, [0, 0.0])
    nw, w = write.get(name, [0, 0.0])
    n = max(nf, nw)
    is_gather = any(p in name for p in GATHER_KERNELS)
    ff = f_gather if is_gather else f_stream
    fk, wk = f / max(nf, 1), w / max(nw, 1)
    rows.append(dict(kernel=short(name), launches=n, launches_per_step=round(n / steps, 2), fetch_factor_used=ff,
                     fetch_kib_raw_per_launch=round(fk, 2), write_kib_raw_per_launch=round(wk, 2),
                     hbm_bytes_per_launch=round((ff * fk + w_stream * wk) * 1024.0, 0),
                     hbm_mb_per_step=round((ff * fk + w_stream * wk) * 1024.0 * n / steps / 1e6, 3)))
rows.sort(key=lambda r: -r["hbm_mb_per_step"])
spconv = [r for r in rows if "spconv_fwd" in r["kernel"]]
api_launches = sum(r["launches_per_step"] for r in spconv)
total_mb = sum(r["hbm_mb_per_step"] for r in spconv)
print(json.dumps(dict(
    how="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only; bytes = (fetch_factor x FETCH_SIZE + "
        "write_factor x WRITE_SIZE) x 1024 with the factors measured on known-byte kernels in the same session (calibration below)",
    unit="HBM-side bytes per launch (calibrated FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, separate passes; Infinity-Cache hits are counted)",
    calibration=cal, steps_profiled=steps,
    spconv_forward=dict(api_launches_per_step=api_launches, hbm_mb_per_step=round(total_mb, 2),
                        hbm_bytes_per_api_launch=round(total_mb * 1e6 / max(api_launches, 1), 0)),
    frame_total_hbm_mb_per_step=round(sum(r["hbm_mb_per_step"] for r in rows), 1),
    kernels=rows[:40]), indent=1))
