#!/bin/bash
# SQ counters of the front-half kernels of the batched pipeline (8 frames per launch, one stream), four --pmc passes.
#   gpurun -- 'bash tools/pmc_front.sh > gpurun_out/pmc_front.txt'
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
KS="k_project_points k_ring_fill k_respond_mfma k_kp_score k_kp_hist k_kp_gather k_kp_emit k_vox_points k_vox_coarse k_vox_clear_lists k_patches"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/pmk; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmk -o pm -- python $R/tools/match_time.py > /dev/null 2>&1
  for k in $KS; do echo "== $k"; python $R/tools/pmc_summary.py /tmp/pmk/pm_results.db $k 2>&1 | grep -v "^$k\|^void"; done
done
