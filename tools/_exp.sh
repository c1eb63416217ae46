cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"; }
for b in 2 3 4; do for p in 0 1 2; do [ $p -lt $b ] || continue; echo -n "buffers=$b pace=$p: "; CAELO_PIPE_PACE=$p run --buffers $b; done; done
