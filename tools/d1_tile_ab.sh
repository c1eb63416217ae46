#!/bin/bash
# A/B of the Dense(200) row tile: profiled launches of every patch on 128-row tiles only, then with the default rule (192 above 16 384 rows)
cd "$(dirname "$0")/.."
for rep in 1 2; do
    CAELO_D1_TILE3_FROM=100000000 python tools/d1_tile_ab.py 2>&1 | grep tiles=
    python tools/d1_tile_ab.py 2>&1 | grep tiles=
done
