"""Per-kernel PMC averages from a rocprofv3 --pmc rocpd database (counters_collection view)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [n for n in names if n.startswith("counters_collection")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for k, c, v in db.execute("select %s, counter_name, value from %s" % (kcol, view)):
    if flt in k:
        acc[k.split("(")[0][:40]][c].append(v)
for k, d in acc.items():
    print(k)
    for c, vs in sorted(d.items()):
        print("   %-28s n=%4d  avg %16.1f" % (c, len(vs), sum(vs) / len(vs)))
