#!/usr/bin/env python
"""Kernel timeline of the LAST pipeline run in a rocprofv3 (rocpd sqlite) trace -- everything after the last gap of more than
`gap_us` between kernels: per stream, start offset and duration.   python tools/timeline_last_run.py <results.db> [gap_us=300]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 3e5
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
cut, t_end = 0, rows[0][2]
for i, r in enumerate(rows):
    if r[1] - t_end > gap:
        cut = i
    t_end = max(t_end, r[2])
sel = rows[cut:]
t0 = sel[0][1]
streams = sorted({r[3] for r in sel})
print("# last run: %d kernels, %.0f us from the first start to the last end" % (len(sel), (max(r[2] for r in sel) - t0) / 1e3))
for r in sel:
    print("%8.1f us  s%-2d %7.1f us  %s" % ((r[1] - t0) / 1e3, streams.index(r[3]), (r[2] - r[1]) / 1e3, r[0].split("(")[0].replace("void ", "")[:40]))
