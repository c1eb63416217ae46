#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / avg / min / max / share."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by name order by sum(end-start) desc"))
tot = sum(r[5] for r in rows)
for line in sys.argv[2:]:
    print("# " + line)
print("%-64s %6s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
for r in rows:
    print("%-64s %6d %10.2f %10.2f %10.2f %6.1f%%" % (r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100 * r[5] / tot))
print("# total kernel time %.3f ms" % (tot / 1e6))
# per stream: the same kernel is launched by bench.py's roofline section (one frame per launch, torch's stream) and by the
# frame pipeline (two frames per launch, the encoder stream)
print("# encoder kernels per stream")
for r in db.execute("select name, stream_id, count(*), avg(end-start) from kernels where name like '%k_enc_%' group by name, stream_id "
                    "order by name, stream_id"):
    print("#   %-44s stream %3s  calls %5d  avg_us %8.2f" % (r[0][:44], r[1], r[2], r[3] / 1e3))
