#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_s1x_empty.txt; mkdir -p $R/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_WAIT_IFETCH" "SQ_IFETCH SQ_ITEMS SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM"; do
  rm -rf /tmp/pme; timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pme -o pm -- python $R/tools/s1x_empty_launch.py 6 > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pme/pm_results.db k_enc_stage1 >> $OUT 2>&1
done
cat $OUT
