"""prints a window of the kernel timeline of a rocprofv3 rocpd database (stream, start, end, duration in us)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = c.execute("select name,stream_id,start,end from kernels order by start").fetchall()
rows = [r for r in rows if not r[0].startswith('__amd')]
idx = [i for i, r in enumerate(rows) if 'track_step' in r[0]]
mid = idx[int(len(idx) * float(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2)]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 70
w = rows[mid:mid + n]; t0 = w[0][2]
for r in w:
    print(f"{r[0].split('(')[0][:26]:<26} s{r[1]} {(r[2]-t0)/1e3:9.1f} {(r[3]-t0)/1e3:9.1f} {(r[3]-r[2])/1e3:8.1f}")
