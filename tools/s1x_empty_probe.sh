#!/bin/bash
# stage 1 on EMPTY and nearly empty patches against the grid size: what an item costs when there is nothing to compute
cd ${GRAFT_REPO_ROOT:-.}
for s in 256 512 768; do echo "S1X_SLOTS=$s"; CAELO_S1X_SLOTS=$s python tools/stage1_density_sweep.py 2>&1 | sed -n 2,4p; done
