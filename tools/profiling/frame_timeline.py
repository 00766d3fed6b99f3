"""The LAST frame of a rocprofv3 kernel trace as a timeline: every dispatch in start order with its queue, the gap since the previous
dispatch on that queue ended, its duration and grid.  usage: frame_timeline.py <results.db> <frames>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); frames = int(sys.argv[2])
rows = db.cursor().execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
per = len(rows) // frames
last = rows[len(rows) - per:]
marks = [i for i, r in enumerate(rows) if 'bt_select_kernel' in r[0]]
if len(marks) >= 2:  # the full forward: a frame = what lies between two box-tail selection kernels (round 6; was: an equal share of ALL
    last = rows[marks[-2] + 1:marks[-1] + 1]  # dispatches of the process, start-up included)
    per = len(last)
t0 = last[0][1]
prev_end = {}
busy = collections.Counter(); gaps = collections.Counter(); n = collections.Counter()
print(f"# {per} dispatches in the last frame, {(max(r[2] for r in last) - t0) / 1e3:.1f} us from the first start to the last end")
for name, s, e, q, gx, wx in last:
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = max(e, prev_end.get(q, 0))
    busy[q] += (e - s) / 1e3; n[q] += 1
    if 0 < gap < 200: gaps[q] += gap
    short = name.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} q{q} gap {gap:7.1f} dur {(e - s) / 1e3:7.1f}  wgs {gx // max(wx, 1):6d}  {short}")
for q in sorted(busy): print(f"# queue {q}: {n[q]} dispatches, busy {busy[q]:.0f} us, gaps (< 200 us each) {gaps[q]:.0f} us")
