"""Per-stream occupancy of the 4-context timed region from a rocprofv3 rocpd database: for every HIP stream the share of wall time it has a kernel
running, the gaps between consecutive kernels, and how many kernels run concurrently on the device over time.   python tools/timeline_stats.py kt_results.db"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = c.execute("select name,stream_id,start,end from kernels order by start").fetchall()
rows = [r for r in rows if not r[0].startswith('__amd') and 'synth' not in r[0] and 'at::' not in r[0] and 'rocprim' not in r[0] and not r[0].startswith('void')]
# the window, as fractions of the pipeline's kernels in start order (argv 2, 3; default 0.35 .. 0.95). bench.py's kernels come in this order: warm-up steps (all contexts),
# the untimed bookkeeping frames (ONE context at a time: F - 1 frames each — inside round 5's default window, which is why its figures were diluted by "1 kernel running"),
# the timed steps (all contexts), the single-context pass. With --steps 4 --warmup 1 the timed region is kernels 0.34 .. 0.99: pass 0.42 0.96.
n = len(rows)
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.35, 0.95)
rows = rows[int(lo * n): int(hi * n)]
t0, t1 = rows[0][2], max(r[3] for r in rows)
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(rows)))
streams = sorted(set(r[1] for r in rows))
for s in streams:
    rs = [r for r in rows if r[1] == s]
    busy = sum(r[3] - r[2] for r in rs)
    gaps = np.array([max(0, rs[i + 1][2] - rs[i][3]) for i in range(len(rs) - 1)]) / 1e3
    print("stream %s: %d kernels, busy %.1f %% of the window, gaps between consecutive kernels: mean %.1f us, median %.1f, p90 %.1f, max %.1f, total %.1f %%" %
          (s, len(rs), 100.0 * busy / (t1 - t0), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max(), 100.0 * gaps.sum() * 1e3 / (t1 - t0)))
# concurrency histogram
ev = sorted([(r[2], 1) for r in rows] + [(r[3], -1) for r in rows])
cur, last, hist = 0, t0, {}
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += d
tot = sum(hist.values())
print("kernels running concurrently (share of the window):", {k: round(100.0 * v / tot, 1) for k, v in sorted(hist.items())})
# which kernels run while nothing else streams: time with no streaming kernel (minz / classify / label) running
S = ('polar_minz', 'classify_compact', 'label_stats')
ev = sorted([(r[2], 1) for r in rows if r[0].startswith(S)] + [(r[3], -1) for r in rows if r[0].startswith(S)])
cur, last, none = 0, t0, 0
for t, d in ev:
    if cur == 0: none += t - last
    last = t; cur += d
print("share of the window with NO streaming kernel (min-z, compaction, label) running: %.1f %%" % (100.0 * none / (t1 - t0)))
