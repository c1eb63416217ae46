#!/bin/bash
# SQ counter passes on the Dense(200) kernel of an 8-frame launch, for the default kernel and CAELO_D1_PLAIN=1 (gpurun_out/d1pmc.txt)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; : > $O/d1pmc.txt
for v in 0 1; do
  echo "== CAELO_D1_PLAIN=$v" >> $O/d1pmc.txt
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
    rm -rf /tmp/pmd; CAELO_D1_PLAIN=$v timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmd -o pm -- python $R/tools/roofline_launch.py 6 8 > /dev/null 2>&1
    python $R/tools/pmc_summary.py /tmp/pmd/pm_results.db k_enc_dense1 >> $O/d1pmc.txt 2>&1
  done
done
