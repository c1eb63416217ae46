#!/bin/bash
# run_sequence.py from page-cache files: N synthetic scans (49 distinct, written N times) as KITTI .bin files under /tmp, read twice (the second run is the warm one)
R=${GRAFT_REPO_ROOT:-$PWD}; N=${1:-1920}
rm -rf /tmp/velo; mkdir -p /tmp/velo
python - <<P
import sys, os
sys.path.insert(0, "$R/cae-lo_amd")
import numpy as np
from caelo import synth
from concurrent.futures import ThreadPoolExecutor
pool = 49
def mk(i): return synth.make_scan(i, quantum=1e-3).astype(np.float32)
with ThreadPoolExecutor(16) as ex: scans = list(ex.map(mk, range(pool)))
def walk(i):
    i %= 2 * (pool - 1); return i if i < pool else 2 * (pool - 1) - i
def wr(i): scans[walk(i)].tofile("/tmp/velo/%06d.bin" % i)
with ThreadPoolExecutor(16) as ex: list(ex.map(wr, range($N)))
print("wrote", $N)
P
for thr in ${THREADS:-16 16 32 64}; do echo "loader threads $thr"; ( cd $R && timeout 600 python cae-lo_amd/run_sequence.py --scans /tmp/velo --chunk ${CHUNK:-960} --loader-threads $thr --out /tmp/poses_files.txt 2>&1 | tail -3 ); done
rm -rf /tmp/velo
