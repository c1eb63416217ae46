#!/usr/bin/env python
"""Per-stream timeline of a rocprofv3 (rocpd sqlite) kernel trace: for a window in the middle of the run, every kernel
with its stream, start offset and duration; then per-stream busy share and the per-kernel average inside the window.
    python tools/timeline.py <results.db> [window_us=1500] [max_rows=120]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1.5e6
max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
t_first, t_last = rows[0][1], rows[-1][2]
# the timed region is the last long burst of pipeline kernels: take the window that ends 30 % before the last
# encoder kernel of the run (well inside the steady state)
enc = [r for r in rows if ("k_enc_stage1" in r[0] or "k_enc_conv2" in r[0])]
big = sorted(e[2] - e[1] for e in enc)[len(enc) // 2]
enc = [e for e in enc if e[2] - e[1] >= 0.6 * big]          # the pipeline's batched launches, not single-frame profiling calls
mid = enc[int(len(enc) * 0.7)][1]
sel = [r for r in rows if r[1] >= mid and r[1] < mid + win]
print("# %d kernels in the run, span %.1f ms; window of %.0f us inside the steady state: %d kernels" % (len(rows), (t_last - t_first) / 1e6, win / 1e3, len(sel)))
streams = sorted({r[3] for r in sel})
for r in sel[:max_rows]:
    print("%9.1f us  s%-3s %8.1f us  %s" % ((r[1] - mid) / 1e3, streams.index(r[3]), (r[2] - r[1]) / 1e3, r[0][:60]))
print("# per-stream busy share inside the window")
for s in streams:
    busy = sum(min(r[2], mid + win) - r[1] for r in sel if r[3] == s)
    print("#   stream s%d (%s): %5.1f %% busy, %d kernels" % (streams.index(s), s, 100 * busy / win, sum(1 for r in sel if r[3] == s)))
agg = {}
for r in sel:
    a = agg.setdefault(r[0][:48], [0, 0.0])
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
print("# per kernel inside the window")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("#   %-48s calls %4d  avg %8.1f us  total %8.1f us" % (k, c, t / c, t))
