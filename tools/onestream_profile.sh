cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/k1; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k1 -o k1 -- python $R/tools/match_time.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/k1/k1_results.db x 2>&1 | head -24 | cut -c1-112
