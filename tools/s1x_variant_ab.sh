#!/bin/bash
# A/B of a stage-1 variant build (variants/libcaelo_$1.so, made with `make BUILD=... OUT=...`) against the library: profiled launches of every patch,
# empty / sparse patches, and the bench at 120 batches
cd "$(dirname "$0")/.."
V=$PWD/variants/libcaelo_$1.so
for rep in 1 2; do
  for lib in "" $V; do
    echo "== ${lib:-library}"
    CAELO_LIB=$lib python tools/stage1_density_sweep.py 0.0 0.0005 0.01 2>&1 | tail -3
    CAELO_LIB=$lib python tools/enc_table.py 2>&1 | grep "frame(s)"
    CAELO_LIB=$lib python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench120', d['value'])"
  done
done
