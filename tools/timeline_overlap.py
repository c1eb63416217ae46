"""Concurrency analysis of a rocprofv3 kernel trace of bench.py: how much of the timed region each
kernel family covers and how many kernels run at once (lanes = the streams with the most dispatches)."""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(db.execute("select name,start,end,queue_id,stream_id from kernels order by start"))
c = collections.Counter((r[3], r[4]) for r in rows)
main = min(c, key=lambda k: k[1])
lanes = [k for k in c if k != main]
lrows = [r for r in rows if (r[3], r[4]) in lanes]
s1 = [r for r in lrows if r[0].startswith('k_enc_stage1')]
tstart = s1[-frames][1] - 120e3
sel = [r for r in lrows if r[1] >= tstart]
span = sel[-1][2] - sel[0][1]
print("%d lanes, %d kernels, %.1f us span, %.1f us/frame" % (len(lanes), len(sel), span / 1e3, span / frames / 1e3))
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
enc = ('k_enc_stage1', 'k_enc_conv3', 'k_enc_dense1', 'k_enc_head')
print('any kernel busy %.1f%%' % (100 * union([(r[1], r[2]) for r in sel]) / span))
print('encoder busy %.1f%%' % (100 * union([(r[1], r[2]) for r in sel if r[0].startswith(enc)]) / span))
for k in enc + ('k_patches', 'k_kp_select', 'k_ransac', 'void k_match', 'k_vox', 'k_respond', 'k_kp_score', 'k_project', 'k_ring', 'k_clear', '__amd'):
    iv = [(r[1], r[2]) for r in sel if r[0].startswith(k)]
    if iv:
        print('%-14s busy %5.1f%%  sum %6.1f us/frame' % (k, 100 * union(iv) / span, sum(e - s for s, e in iv) / frames / 1e3))
def conc(pred):
    ev = []
    for r in sel:
        if pred(r): ev += [(r[1], 1), (r[2], -1)]
    ev.sort(); cur = 0; last = ev[0][0]; hist = collections.Counter()
    for t, d in ev:
        hist[cur] += t - last; last = t; cur += d
    return {k: round(100 * v / span, 1) for k, v in sorted(hist.items())}
print('concurrent big encoder kernels (% of time):', conc(lambda r: r[0].startswith(enc[:3])))
print('concurrent kernels (% of time):', conc(lambda r: True))
