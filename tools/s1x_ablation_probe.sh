#!/bin/bash
# s1x_ablation_probe.sh -- where the per-item floor of k_enc_stage1x sits: builds of the kernel with parts compiled out
# (EXTRA=-DS1X_ABL=bits, a throw-away patch of enc_stage1x.inc kept in profiles/r06_s1x_ablation.txt), 24 576 EMPTY and sparse patches each.
# bits: 1 no P2 stores, 2 no background-output loads, 4 no conv1 phase (and its barrier), 8 no mask phase, 16 no row prefetch / queue atomic,
# 32 no active-slab ballot, 64 no end-of-item barrier
cd "$(dirname "$0")/.."
for v in ${ABL_LIST:-0 1 3 7 15 16 144 31 159 191 255}; do
    lib=variants/libcaelo_abl$v.so
    [ $v = 0 ] && lib=cae-lo_amd/caelo/libcaelo.so
    [ -f $lib ] || continue
    echo "== S1X_ABL=$v"
    CAELO_LIB=$PWD/$lib python tools/stage1_density_sweep.py 0.0 0.002 0.03 2>&1 | tail -3
done
