"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table.  usage: prof_summary.py <db> <frames> [title]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); frames = int(sys.argv[2]); title = sys.argv[3] if len(sys.argv) > 3 else ''
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# {title}")
print(f"# total kernel time {tot/1e3:.1f} ms over {frames} frames = {tot/frames/1e3:.2f} ms/frame, {sum(r[1] for r in rows)/frames:.0f} launches/frame")
print(f"{'calls/f':>8} {'ms/frame':>9} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  name")
for r in rows[:70]:
    print(f"{r[1]/frames:8.1f} {r[2]/frames/1e3:9.3f} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f} {100*r[2]/tot:6.2f}  {r[0][:130]}")
