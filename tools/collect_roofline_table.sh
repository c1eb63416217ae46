#!/bin/bash
# per-kernel table of the batched launches (one stream): gpurun_out/prof/roofline_table_batched.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rt_trace; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rt_trace -o t -- python $R/tools/match_time.py > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/rt_$c; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/rt_$c -o p -- python $R/tools/match_time.py > /dev/null 2>&1
done
DISTINCT=$(python $R/tools/distinct_patches.py 2>/dev/null | tail -1)
( echo "# round 4: per-kernel table of the BATCHED launches (8 frames / 8 pairs per launch), ONE stream (CAELO_PIPE_STREAMS=1 python tools/match_time.py),"
  echo "# de-duplication on (the encoder kernels work on the ~1 900 distinct patches of each frame).  Durations: rocprofv3 --kernel-trace; HBM bytes ="
  echo "# (FETCH_SIZE + WRITE_SIZE) * 1024 from two separate --pmc passes, uncorrected (FETCH_SIZE counts wide streaming reads at half their bytes on gfx950);"
  echo "# min MB = minimum HBM traffic of the launch (SURVEY 8d) and min/us = the share of 8 TB/s the kernel would reach if it moved only that; alg TF/s and %pipe: algorithmic FLOPs against the pipe the kernel runs on; MFMAbusy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 SEs (two more passes)."
  python $R/tools/roofline_table_batched.py /tmp/rt_trace/t_results.db /tmp/rt_FETCH_SIZE/p_results.db /tmp/rt_WRITE_SIZE/p_results.db /tmp/rt_SQ_VALU_MFMA_BUSY_CYCLES/p_results.db /tmp/rt_SQ_BUSY_CYCLES/p_results.db $DISTINCT ) > $O/roofline_table_batched.txt 2>&1
cat $O/roofline_table_batched.txt
