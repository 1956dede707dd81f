python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch $1 --contexts $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['frames_per_step_per_gpu'], d['config']['contexts_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for cfg in ${SWEEP:-"128 1" "384 3"}; do run $cfg; done
