#!/bin/bash
# SQ counters of one kernel (name filter $1) while running a command ($2...), separate --pmc passes.
#   gpurun -- 'bash tools/pmc_kernel.sh k_match_screen python tools/match_time.py'
R=${GRAFT_REPO_ROOT:-$PWD}
K=$1; shift
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pmk; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmk -o pm -- "$@" > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pmk/pm_results.db $K 2>&1 | grep -v "^$K\|^void"
done
