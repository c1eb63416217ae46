#!/usr/bin/env python
"""profiles/r01_pmc_traffic*.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command:
   python tools/pmc_traffic_json.py fetch.db write.db "source note" > profiles/r01_pmc_traffic_vNN.json"""
import collections, json, sqlite3, sys


def avg(path, counter):
    db = sqlite3.connect(path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [n for n in names if n.startswith("counters_collection")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    acc = collections.defaultdict(list)
    for k, c, v in db.execute("select %s, counter_name, value from %s" % (kcol, view)):
        if c == counter:
            acc[k.split("(")[0].replace("void ", "").split("<")[0]].append(v)
    return {k: round(sum(v) / len(v), 1) for k, v in acc.items()}


f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {"source": sys.argv[3], "kernels": {k: {"FETCH_SIZE_KB": f.get(k, 0.0), "WRITE_SIZE_KB": w.get(k, 0.0)} for k in sorted(set(f) | set(w))}}
print(json.dumps(out, indent=1))
