#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_s1x_empty.txt; mkdir -p $R/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH"; do
  rm -rf /tmp/pme; timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pme -o pm -- python $R/tools/s1x_empty_launch.py 6 > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pme/pm_results.db k_enc_stage1 >> $OUT 2>&1
done
cat $OUT
