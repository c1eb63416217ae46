#!/usr/bin/env python
"""Per-kernel table of the batched launches (8 frames / 8 pairs per launch, one stream, nothing beside them) from five rocprofv3
runs of `CAELO_PIPE_STREAMS=1 python tools/match_time.py` (rocpd sqlite files):
   python tools/roofline_table_batched.py trace.db fetch.db write.db mfma.db busy.db
trace: --kernel-trace --stats; the others: separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES)."""
import collections, sqlite3, sys

FRAMES = 8
# ---- algorithmic work per LAUNCH of the batched pipeline (8 frames / 8 pairs), SURVEY 8d figures; the encoder rows are scaled by the
# distinct patches per batch the run reports (argv[6], default: 0.55 x 24 576).  Per frame: N = 126.7 k points (2.03 MB), ring 69 x 1800,
# network grid 64 x 1792 (fused path: rows 6..57 of the response), K = 1024 key points, 3072 patches, 27 k / 4 k bricks at the 16 / 64 cm scales.
DISTINCT = float(sys.argv[6]) if len(sys.argv) > 6 else 0.55 * 24576
ROWS = 52.0 / 64.0
# kernel -> (minimum HBM MB per launch, what it is)
ALG_MB = {
    "k_clear_set": (8 * (0.497 + 0.46), "winner image + score / candidate accumulators wiped"),
    "k_vox_clear_lists": (8 * 31e3 * 72 / 1e6, "31 k bricks x (8 B key + 64 B payload) wiped from their lists"),
    "k_project_points": (8 * 2.03, "points read once"),
    "k_ring_fill": (8 * (0.497 + 2.48 + 1.8), "winners read, winning points gathered, ring written"),
    "k_respond_mfma": (8 * (1.376 + 3.670) * ROWS, "x, y, z of the ring rows read, response written"),
    "k_kp_score": (8 * (3.670 * ROWS + 0.46), "response read, candidates appended"),
    "k_kp_hist": (8 * 0.9, "candidate keys read"), "k_kp_gather": (8 * 0.9, "candidate keys read"), "k_kp_emit": (8 * 0.05, "selected keys -> key points"),
    "k_vox_points": (8 * (2.03 + 27e3 * 72 / 1e6), "points read, scale-0 bricks written"),
    "k_vox_coarse": (8 * (27e3 * 72 + 4.2e3 * 72) / 1e6, "scale-0 bricks read, scale-1 bricks written"),
    "k_vox_coarse2": (8 * (4.2e3 * 72 + 700 * 72) / 1e6, "scale-1 bricks read, scale-2 bricks written"),
    "k_vox_suspects_resolve": (8 * 0.01, "a few dozen face points"),
    "k_patches": (8 * (31e3 * 72 / 1e6 + 1.57), "every brick once, bit-packed patches written"),
    "k_dd_verify": (8 * 1.57, "patches read"), "k_dd_scan": (8 * 0.05, "tables"),
    "k_enc_stage1x": ((64 * 8 + 4096) * DISTINCT / 1e6, "patch bits in, P2 out (distinct patches)"),
    "k_enc_conv3": ((4096 + 8192) * DISTINCT / 1e6, "P2 in, F3 out"),
    "k_enc_dense1p": ((8192 + 4 * 832) * DISTINCT / 1e6, "F3 in, four k-slice partial sums out"),
    "k_enc_head_mfma": ((4 * 832) * DISTINCT / 1e6 + 8 * 0.246, "partial sums in, descriptors out"),
    "k_match_prep": (8 * (0.262 + 0.27), "rows read, fragment image written"),
    "k_match_screen": (8 * (0.27 + 0.262 + 0.008), "fragment image + frame-1 rows read once, indices written"),
    "k_ransac_hyp": (8 * 0.05, "pairs + draws"), "k_ransac_finish": (8 * 0.06, "pairs, counts, mask"),
}
# kernel -> (algorithmic GFLOP per launch, pipe peak TFLOP/s, pipe)
ALG_GFLOP = {
    "k_respond_mfma": (0.257 * ROWS * 8, 157.3, "f32 MFMA"),
    "k_kp_score": (8 * 48 * 1776 * 24 * 23 / 1e9, 157.3, "f32 VALU"),
    "k_enc_stage1x": ((1.769 + 3.539) * DISTINCT / 1e3, 2500.0, "f16 MFMA, dense-Keras FLOPs (the kernel skips exact zeros)"),
    "k_enc_conv3": (1.769 * DISTINCT / 1e3, 2500.0 / 3, "f16 MFMA / 3 terms"),
    "k_enc_dense1p": (0.8192 * DISTINCT / 1e3, 2500.0 / 3, "f16 MFMA / 3 terms"),
    "k_enc_head_mfma": (0.008 * DISTINCT / 1e3 * (24576 / DISTINCT), 157.3, "f32 MFMA"),
    "k_match_screen": (8 * 0.1258, 2500.0 / 12, "f16 MFMA, 2 sweeps x 6 MFMAs per f32-grade product block"),
    "k_ransac_hyp": (8 * 0.05, 78.6, "f64 VALU (upper bound of the work)"),
}


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
print("%-24s %5s %8s %9s %8s %6s %8s %6s %9s %7s %8s  %s" % ("kernel", "calls", "avg_us", "MB/launch", "GB/s", "%HBM", "min MB", "min/us", "alg TF/s", "%pipe", "MFMAbusy", "bound / what the minimum is"))
order = sorted(rows.items(), key=lambda kv_: -sum(sorted(kv_[1])[len(kv_[1]) // 4:]))
tot_us = 0.0
for k, d in order:
    if not k.startswith("k_"):
        continue
    d = sorted(d)[len(d) // 4:]          # the batched launches
    us = sum(d) / len(d)
    tot_us += us
    mb = (fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024 / 1e6
    gbs = mb / us * 1e3 if us else 0.0
    amb, what = ALG_MB.get(k, (None, ""))
    ambs = "%8.1f %5.1f%%" % (amb, 100 * (amb / us * 1e3) / 8000.0) if amb is not None else "%8s %6s" % ("-", "-")   # min MB and the share of 8 TB/s it would be
    if k in ALG_GFLOP:
        gf, peak, pipe = ALG_GFLOP[k]
        tf = "%9.1f %6.1f%%" % (gf / us * 1e3, 100 * gf / us * 1e3 / peak)
        what = pipe + "; " + what
    else:
        tf = "%9s %7s" % ("-", "-")
    b = "%7.0f%%" % (100 * (mfma[k] / 1024.0) / (busy[k] / 32.0)) if mfma.get(k, 0) > 0 and busy.get(k, 0) > 0 else "%8s" % "-"
    print("%-24s %5d %8.2f %9.2f %8.0f %5.1f%% %s %s %s  %s" % (k, len(d), us, mb, gbs, 100 * gbs / 8000.0, ambs, tf, b, what))
print("# sum of the batched launches: %.1f us per 8 frames (every kernel alone on the GPU); distinct patches per batch used for the encoder rows: %.0f" % (tot_us, DISTINCT))
