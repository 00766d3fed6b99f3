import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
pt = [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()]
print(pt)
for t in pt[:6]:
    try:
        cols = [d[1] for d in cur.execute(f"pragma table_info({t})")]
        print(t, cols)
    except Exception as e:
        print(t, e)
