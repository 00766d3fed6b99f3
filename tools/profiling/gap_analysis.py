"""GPU busy fraction and the largest idle gaps of the steady-state frames from a rocprofv3 kernel trace db.
usage: gap_analysis.py <db> <frames_total> <frames_skip>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); nfr = int(sys.argv[2]); skip = int(sys.argv[3])
rows = db.cursor().execute("select start, end, name from kernels order by start").fetchall()
# frame boundaries: use the voxelize kernel as the frame marker
marks = [r[0] for r in rows if 'voxelize' in r[2] and 'divfloor' not in r[2]]
print('voxelize launches', len(marks))
t0 = marks[skip] if len(marks) > skip else rows[0][0]
rows = [r for r in rows if r[0] >= t0]
t_end = max(r[1] for r in rows)
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]; gaps = []
prev_name = rows[0][2]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: prev_name = n
busy += cur_e - cur_s
wall = t_end - t0
fr = len(marks) - skip
print(f'frames {fr}: wall {wall/1e6/fr:.2f} ms/frame, GPU busy {busy/1e6/fr:.2f} ms/frame ({100*busy/wall:.1f} %), idle {(wall-busy)/1e6/fr:.2f} ms/frame in {len(gaps)/fr:.0f} gaps/frame')
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1e3) // 10 * 10, 200)] += g
print('idle time by gap length (us bucket -> ms/frame):', {k: round(v / 1e6 / fr, 3) for k, v in sorted(hist.items())})
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    k = (a.split('(')[0][-60:], b.split('(')[0][-60:]); agg[k][0] += 1; agg[k][1] += g
for k, (c, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'{g/1e6/fr:7.3f} ms/frame {c/fr:5.1f}x  after [{k[0]}] before [{k[1]}]')
