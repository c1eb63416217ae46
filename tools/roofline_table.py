#!/usr/bin/env python
"""Per-kernel roofline table from four rocprofv3 runs of the same command (rocpd sqlite files):
   python tools/roofline_table.py trace.db fetch.db write.db mfma.db
trace: --kernel-trace --stats; fetch / write / mfma: separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES)."""
import collections, sqlite3, sys

ALG_GFLOP = {"k_enc_stage1": 16.307, "k_enc_conv3": 5.436, "k_enc_dense1": 2.517, "k_match_mfma": 0.1258, "k_respond": 0.257}
PEAK_TF = {"k_enc_stage1": 157.3, "k_enc_conv3": 157.3, "k_enc_dense1": 157.3, "k_match_mfma": 78.6, "k_respond": 157.3}


def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "").split("<")[0]


def pmc(path, counter):
    db = sqlite3.connect(path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [n for n in names if n.startswith("counters_collection")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    acc = collections.defaultdict(list)
    for k, c, v in db.execute("select %s, counter_name, value from %s" % (kcol, view)):
        if c == counter:
            acc[short(k)].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


trace = sqlite3.connect(sys.argv[1])
rows = list(trace.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"))
fetch, write, mfma = pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE"), pmc(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES")
print("%-18s %6s %9s %9s %9s %7s %10s %9s %9s" % ("kernel", "calls", "avg_us", "MB/launch", "GB/s", "%HBM", "alg TF/s", "%f32pk", "MFMAbusy%"))
for name, calls, avg_ns, _ in rows:
    k = short(name)
    if not k.startswith("k_"):
        continue
    us = avg_ns / 1e3
    mb = (fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / 1e6
    gbs = mb / us * 1e3 if us else 0.0
    tf = "%10.1f %8.0f%%" % (ALG_GFLOP[k] / us * 1e3, 100 * ALG_GFLOP[k] / us * 1e3 / PEAK_TF[k]) if k in ALG_GFLOP else "%10s %9s" % ("-", "-")
    busy = "%8.0f%%" % (100 * mfma[k] / 1024 / 2400.0 / us) if mfma.get(k, 0) > 0 else "%9s" % "-"
    print("%-18s %6d %9.2f %9.2f %9.0f %6.1f%% %s %s" % (k, calls, us, mb, gbs, 100 * gbs / 8000.0, tf, busy))
