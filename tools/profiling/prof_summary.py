"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table.  usage: prof_summary.py <db> <frames> [title]
Round 6: when the trace holds the full forward (one `bt_select_kernel` = the box tail's last kernel per frame) the table covers the
frames BETWEEN the first and the last of those markers only — weight preparation, input upload and allocator warm-up of the process
start were counted into "launches / frame" before (r5's 624 is ~60 too many)."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); frames = int(sys.argv[2]); title = sys.argv[3] if len(sys.argv) > 3 else ''
rows = db.cursor().execute("select name, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if 'bt_select_kernel' in r[0]]
note = f"all dispatches of the process / {frames} frames"
if len(marks) >= 3:
    rows = rows[marks[0] + 1:marks[-1] + 1]
    frames = len(marks) - 1
    note = f"the {frames} frames between the first and the last box-tail selection kernel (process start-up excluded)"
agg = collections.OrderedDict()
for name, s, e in rows:
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tab = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in tab)
print(f"# {title}")
print(f"# {note}")
print(f"# total kernel time {tot/1e3:.1f} ms over {frames} frames = {tot/frames/1e3:.2f} ms/frame, {sum(v[0] for _, v in tab)/frames:.0f} launches/frame")
print(f"{'calls/f':>8} {'ms/frame':>9} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  name")
for name, v in tab[:70]:
    print(f"{v[0]/frames:8.1f} {v[1]/frames/1e3:9.3f} {v[1]/v[0]:10.1f} {v[2]:10.1f} {v[3]:10.1f} {100*v[1]/tot:6.2f}  {name[:130]}")
