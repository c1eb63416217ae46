#!/usr/bin/env python
"""HBM traffic and matrix-pipe busy share of the encoder / match launches, measured NOW with rocprofv3 (bench.py's `roofline.traffic`,
`mfma_busy_pmc`; VERDICT r4 item 3: not read from a committed file).

Three separate `rocprofv3 --pmc` passes of `tools/roofline_launch.py 6 8` (8 frames = 24 576 patches per encoder launch, 8 pairs per
match launch -- the pipeline's launch shapes; counters only, no other trace domain):
    pass 1  FETCH_SIZE                                   (KB per dispatch; 3 of the 4 TCC slots)
    pass 2  WRITE_SIZE                                   (KB per dispatch)
    pass 3  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES      (summed over 1024 SIMDs / over 32 shader engines)
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies a 128-byte request of a wide coalesced read as 64 bytes -- the
`*_corrected` figures double it; WRITE_SIZE is uncalibrated there and is reported as it comes.

    python tools/pmc_live.py [out.json]      -> JSON on stdout (and to the file)
Importable: collect() -> dict or None (rocprofv3 missing / a pass failed: the caller falls back to the committed file and says so).
"""
import collections
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")]


def _short(k):
    return k.split("(")[0].replace("void ", "").split("<")[0].strip()


def _averages(db_path):
    db = sqlite3.connect(db_path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [n for n in names if n.startswith("counters_collection")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v in db.execute("select %s, counter_name, value from %s" % (kcol, view)):
        acc[_short(k)][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def collect(repeats=6, frames=8, timeout=90, keep_dir=None):
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    work = keep_dir or tempfile.mkdtemp(prefix="caelo_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    merged = collections.defaultdict(dict)
    try:
        for i, counters in enumerate(PASSES):
            d = os.path.join(work, "pass%d" % i)
            cmd = [exe, "--pmc"] + list(counters) + ["-d", d, "-o", "pm", "--", sys.executable, os.path.join(REPO, "tools", "roofline_launch.py"),
                                                      str(repeats), str(frames), "match"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return None
            for k, vals in _averages(dbs[0]).items():
                merged[k].update(vals)
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error, IndexError):
        return None
    finally:
        if keep_dir is None:
            shutil.rmtree(work, ignore_errors=True)
    out = {}
    for k, v in merged.items():
        if not (k.startswith("k_enc") or k.startswith("k_match")):
            continue
        e = {}
        if "FETCH_SIZE" in v:
            e["fetch_kb"] = round(v["FETCH_SIZE"], 1)
            e["fetch_kb_corrected"] = round(2.0 * v["FETCH_SIZE"], 1)
        if "WRITE_SIZE" in v:
            e["write_kb"] = round(v["WRITE_SIZE"], 1)
        if v.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            e["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["SQ_BUSY_CYCLES"] / 32.0), 4)
        out[k] = e
    if not out:
        return None
    return {"source": "rocprofv3 --pmc, three separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES) of "
                      "tools/roofline_launch.py %d %d match, run by this process; KB per dispatch; fetch_kb_corrected = 2 x FETCH_SIZE "
                      "(gfx950 tallies a 128-byte request as 64: MI355X_MICROARCH.md), WRITE_SIZE as reported" % (repeats, frames),
            "launch": "%d frames = %d patches per encoder launch, %d pairs per match launch" % (frames, frames * 3072, frames),
            "kernels": out}


if __name__ == "__main__":
    res = collect()
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 1 and res is not None:
        open(sys.argv[1], "w").write(txt + "\n")
    sys.exit(0 if res is not None else 1)
