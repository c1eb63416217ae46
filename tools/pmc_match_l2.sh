#!/bin/bash
# L2 side of k_match_screen (and the stage-1 kernel beside it): requests / hits / misses per dispatch, separate --pmc passes
# (rocprofv3 --pmc only with --kernel-trace: MI355X_MICROARCH.md).   gpurun -- 'bash tools/pmc_match_l2.sh > gpurun_out/pmc_match_l2.txt'
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for grp in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  rm -rf /tmp/pml
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pml -o pm -- python $R/tools/roofline_launch.py 6 8 match > /dev/null 2>&1
  echo "== $grp"
  python $R/tools/pmc_summary.py /tmp/pml/pm_results.db k_match_screen 2>&1 | tail -6
done
