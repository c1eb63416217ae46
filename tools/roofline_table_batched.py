#!/usr/bin/env python
"""Per-kernel table of the batched launches (8 frames / 8 pairs per launch, one stream, nothing beside them) from five rocprofv3
runs of `CAELO_PIPE_STREAMS=1 python tools/match_time.py` (rocpd sqlite files):
   python tools/roofline_table_batched.py trace.db fetch.db write.db mfma.db busy.db
trace: --kernel-trace --stats; the others: separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES)."""
import collections, sqlite3, sys

FRAMES = 8
# algorithmic GFLOP per LAUNCH where the launch's work is known without the de-duplication tables (SURVEY 8d figures x 8)
ALG_GFLOP = {"k_respond": 0.257 * FRAMES, "k_match_mfma": 0.1258 * FRAMES}
PEAK_TF = {"k_respond": 157.3, "k_match_mfma": 78.6}
# minimum HBM bytes per launch (SURVEY 8d "min-traffic bytes" x 8 frames), MB
ALG_MB = {"k_project_points": 2.03 * FRAMES, "k_respond": (1.376 + 3.670) * FRAMES, "k_kp_score": 3.670 * FRAMES, "k_vox_points": 2.03 * FRAMES}


def short(n):
    return n.split("(")[0].replace("void ", "").split("<")[0]


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
    # the run starts with one single-frame extract: keep the batched launches (the upper half of the values)
    out = {}
    for k, v in acc.items():
        v = sorted(v)
        out[k] = sum(v[len(v) // 4:]) / max(len(v[len(v) // 4:]), 1)
    return out


trace = sqlite3.connect(sys.argv[1])
names = [r[0] for r in trace.execute("select name from sqlite_master where type in ('table','view')")]
kv = "kernels" if "kernels" in names else [n for n in names if n.startswith("kernels")][0]
rows = collections.defaultdict(list)
for name, s, e in trace.execute("select name, start, end from %s" % kv):
    rows[short(name)].append((e - s) / 1e3)
fetch, write = pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE")
mfma, busy = pmc(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES"), pmc(sys.argv[5], "SQ_BUSY_CYCLES")
print("%-22s %5s %8s %9s %8s %6s %8s %9s %8s %8s" % ("kernel", "calls", "avg_us", "MB/launch", "GB/s", "%HBM", "min MB", "alg TF/s", "%peak", "MFMAbusy"))
order = sorted(rows.items(), key=lambda kv_: -sum(sorted(kv_[1])[len(kv_[1]) // 4:]))
for k, d in order:
    if not k.startswith("k_"):
        continue
    d = sorted(d)[len(d) // 4:]          # the batched launches
    us = sum(d) / len(d)
    mb = (fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / 1e6
    gbs = mb / us * 1e3 if us else 0.0
    amb = "%8.1f" % ALG_MB[k] if k in ALG_MB else "%8s" % "-"
    tf = "%9.1f %7.0f%%" % (ALG_GFLOP[k] / us * 1e3, 100 * ALG_GFLOP[k] / us * 1e3 / PEAK_TF[k]) if k in ALG_GFLOP else "%9s %8s" % ("-", "-")
    b = "%7.0f%%" % (100 * (mfma[k] / 1024.0) / (busy[k] / 32.0)) if mfma.get(k, 0) > 0 and busy.get(k, 0) > 0 else "%8s" % "-"
    print("%-22s %5d %8.2f %9.2f %8.0f %5.1f%% %s %s %s" % (k, len(d), us, mb, gbs, 100 * gbs / 8000.0, amb, tf, b))
