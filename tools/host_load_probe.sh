#!/bin/bash
# How the certified 120-batch run holds up when the host is busy: N busy-loop processes beside the bench, CAELO_CERT_THREADS 3 (default) / 6 / 8
cd "$(dirname "$0")/.."
run() { CAELO_CERT_THREADS=$1 python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('host_issue_us_per_frame'))"; }
nproc
for load in 0 ${LOAD:-}; do   # LOAD=224 on a 256-thread host oversubscribes it: the runs then take minutes (frames/s in the hundreds)
  pids=""
  for i in $(seq $load); do ( while :; do :; done ) & pids="$pids $!"; done
  sleep 1
  for t in 3 6 8 3 6 8; do echo "busy loops $load, certifier threads $t: $(run $t)"; done
  echo "busy loops $load, no certificates: $(run 3 --no-certify)"
  [ -n "$pids" ] && kill $pids 2>/dev/null
  wait 2>/dev/null
done
