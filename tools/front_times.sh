#!/bin/bash
# per-kernel durations of the batched launches (8 frames per launch), ONE stream:  gpurun -- 'bash tools/front_times.sh [filter]'
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k1; CAELO_PIPE_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/k1 -o k1 -- python $R/tools/match_time.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/k1/k1_results.db "one stream, 8 frames per launch" | grep -E "${1:-k_}" | grep -v "k_enc\|k_match\|k_ransac"
