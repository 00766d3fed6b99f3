"""In-situ durations of one kernel family, dispatch by dispatch, in the LAST frame of a rocprofv3 kernel trace.
usage: dispatch_list.py <results.db> <frames> <name substring> [...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); frames = int(sys.argv[2])
cur = db.cursor()
for pat in sys.argv[3:]:
    rows = cur.execute("select name, start, end from kernels where name like ? order by start", (f"%{pat}%",)).fetchall()
    per = len(rows) // frames
    last = rows[len(rows) - per:]
    print(f"# {pat}: {per} dispatches per frame, {sum(e - s for _, s, e in last) / 1e3:.1f} us in the last frame")
    t0 = last[0][1] if last else 0
    for n, s, e in last:
        print(f"  +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {n[:90]}")
