#!/bin/bash
# quick_check.sh TAG [pytest -k expression] -- one GPU-box round trip while iterating on a kernel: the selected GPU tests, the
# bench at the driver's shape, per-kernel durations inside the pipeline and alone (one stream).  Output: gpurun_out/q_TAG/
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}; K=${2:-}
O=$R/gpurun_out/q_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -n "$K" ]; then ( cd $R && timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -15 ) > $O/tests.log; fi
timeout 400 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_20.json 2> $O/bench_20.err
timeout 400 python $R/bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary > $O/bench_120.json 2>/dev/null
rm -rf /tmp/kb; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kb -o kb -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/kb/kb_results.db "bench.py --steps 60 --warmup 6 (inside the pipeline)" > $O/kernel_stats_bench.txt 2>&1
rm -rf /tmp/k1; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k1 -o k1 -- python $R/tools/match_time.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/k1/k1_results.db "CAELO_PIPE_STREAMS=1 tools/match_time.py (8 frames per launch, one stream: every kernel alone)" > $O/kernel_stats_one_stream.txt 2>&1
tail -3 $O/tests.log 2>/dev/null
python - <<PY
import json
for n in ("bench_20", "bench_120"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], "stage1 frac", d["roofline"]["frac"], "ms", {k: v["ms"] for k, v in d["roofline"]["encoder_kernels"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
