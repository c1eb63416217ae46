#!/bin/bash
# stage-1 grid-size sweep (workgroups: 256 = one per CU, 512 = two, 768 = three) on the 8-frame launch, HIP events (tools/enc_table.py)
cd ${GRAFT_REPO_ROOT:-.}
for s in 256 512 768; do echo "S1X_SLOTS=$s"; CAELO_S1X_SLOTS=$s python tools/enc_table.py 2>&1 | tail -2; done
