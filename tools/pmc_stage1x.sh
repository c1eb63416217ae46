#!/bin/bash
# SQ counters of the stage-1 kernel on the 8-frame launch (tools/roofline_launch.py N 8), three separate --pmc passes + a kernel trace.
#   gpurun -- 'bash tools/pmc_stage1x.sh [out file]'      (CAELO_LIB / CAELO_ENC_S1 select the build / kernel)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/pmc_stage1x.txt}
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
: > $OUT
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_BRANCH" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pms; timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pms -o pm -- python $R/tools/roofline_launch.py 6 8 > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pms/pm_results.db k_enc_stage1 >> $OUT 2>&1
done
rm -rf /tmp/rls; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rls -o rl -- python $R/tools/roofline_launch.py 20 8 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/rls/rl_results.db 2>&1 | grep "k_enc" | head -4 >> $OUT
cat $OUT
