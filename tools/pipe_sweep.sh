#!/bin/bash
# frames/s of the default bench (120 batches) against the stage-1 grid size and the encoder's yield bits inside the pipeline
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"; }
for s in 480 512 544; do for y in 1 3 5; do echo -n "CAELO_S1X_SLOTS=$s CAELO_ENC_YIELD=$y: "; CAELO_S1X_SLOTS=$s CAELO_ENC_YIELD=$y run; done; done
echo -n "again 512/5: "; CAELO_S1X_SLOTS=512 CAELO_ENC_YIELD=5 run
echo -n "20 steps 512/5: "; CAELO_S1X_SLOTS=512 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])"
