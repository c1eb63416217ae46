#!/usr/bin/env python
"""When does each batch of the timed region start and finish?  From a rocprofv3 kernel trace of `bench.py --steps K --no-secondary
--no-cpu-baseline --no-pmc` (rocpd sqlite): the hypothesis kernel only runs inside the pipeline, so its last K launches are the timed
batches; the front / encoder launches of the same batches are the K before the ones that come after the timed region (tables, profiles).
    python tools/batch_cadence.py <results.db> K"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); K = int(sys.argv[2])
rows = list(db.execute("select name, start, end from kernels order by start"))
hy = [r for r in rows if "k_ransac_hyp" in r[0]][-K:]
t_end = hy[-1][2]
def timed(name):
    sel = [r for r in rows if name in r[0] and r[1] < t_end and r[2] - r[1] > 0]
    return sel[-K:]
pp, s1, hd = timed("k_project_points"), timed("k_enc_stage1x"), timed("k_enc_head_mfma")
t0 = pp[0][1]
print("# batch: front start | stage 1 start .. end | encoder (head) end | hypotheses end      us since the first front kernel; (+) = since the batch before")
p_e = p_h = None
for b in range(K):
    e, h = (hd[b][2] - t0) / 1e3, (hy[b][2] - t0) / 1e3
    print("%3d  %8.1f | %8.1f .. %8.1f | %8.1f %-9s | %8.1f %s" % (b, (pp[b][1] - t0) / 1e3, (s1[b][1] - t0) / 1e3, (s1[b][2] - t0) / 1e3, e,
          "" if p_e is None else "(+%.0f)" % (e - p_e), h, "" if p_h is None else "(+%.0f)" % (h - p_h)))
    p_e, p_h = e, h
