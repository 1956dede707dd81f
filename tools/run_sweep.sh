run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch $1 --contexts $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['frames_per_step_per_gpu'], d['config']['contexts_per_gpu'], d['value'], d['ms_per_step'], 'host_issue', d.get('host_issue_ms_per_step'))"; }
for cfg in "$@"; do run $cfg; done
