#!/usr/bin/env python
"""HIP API calls longer than a threshold inside the timed region of a `rocprofv3 --hip-trace --kernel-trace` run of bench.py (rocpd sqlite):
which host call is it that holds a batch up?   python tools/long_api_calls.py <results.db> K [min_us=80]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); K = int(sys.argv[2]); thr = float(sys.argv[3]) if len(sys.argv) > 3 else 80.0
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ker = list(db.execute("select name, start, end from kernels order by start"))
hy = [r for r in ker if "k_ransac_hyp" in r[0]][-K:]
pp = [r for r in ker if "k_project_points" in r[0] and r[1] < hy[-1][2]][-K:]
t0, t1 = pp[0][1] - 300000, hy[-1][2]
view = [n for n in names if n == "regions"] or [n for n in names if "region" in n]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view[0])]
print("# view %s columns %s" % (view[0], cols))
q = "select name, start, end, tid from %s where start >= ? and start <= ? and (end - start) >= ? order by start" % view[0]
for name, s, e, tid in db.execute(q, (t0, t1, thr * 1e3)):
    print("%9.1f us  %8.1f us  tid %s  %s" % ((s - pp[0][1]) / 1e3, (e - s) / 1e3, tid, name))
