#!/usr/bin/env python
"""The last burst of kernels of a rocprofv3 (rocpd sqlite) trace -- e.g. the timed region of `bench.py --steps 20` -- one line
per kernel: start offset, stream, duration.   python tools/timeline_tail.py <results.db> [gap_us=300]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 3e5
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
# the last burst that contains a pair-stage kernel (bench.py's roofline section after the timed region has none)
last = max(i for i, r in enumerate(rows) if "k_ransac_finish" in r[0])
rows = rows[:last + 1]
i = len(rows) - 1
lo = rows[i][1]
while i > 0:
    prev_end = max(r[2] for r in rows[max(0, i - 40):i])
    if rows[i][1] - prev_end > gap:
        break
    i -= 1
sel = rows[i:]
t0 = sel[0][1]
streams = sorted({r[3] for r in sel})
print("# %d kernels in the last burst, span %.1f us" % (len(sel), (max(r[2] for r in sel) - t0) / 1e3))
for r in sel:
    print("%9.1f us  s%-2d %8.1f us  %s" % ((r[1] - t0) / 1e3, streams.index(r[3]), (r[2] - r[1]) / 1e3, r[0][:50]))
