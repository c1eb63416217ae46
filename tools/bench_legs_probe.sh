cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
nproc; uptime
for i in 1 2 3; do
python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('120', d['value'], d['config'].get('host_issue_us_per_frame'))"
done
python bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc --no-certify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('120 no-certify', d['value'], d['config'].get('host_issue_us_per_frame'))"
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20', d['value'], d['config'].get('host_issue_us_per_frame'))"
done
uptime
