cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
run() { MOT_BENCH_LIB=$2 timeout 300 python bench.py --scene plaza --batch 512 --contexts 4 --steps 3 --warmup 1 --no-aux --no-cpu-baseline --no-all-outputs 2>/dev/null | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-12s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2 3; do run product ""; run oldlabel $PWD/variants/libmot_oldlabel.so; done | tee gpurun_out/plaza_ab.txt
