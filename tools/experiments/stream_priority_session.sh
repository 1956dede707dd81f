# EXPERIMENT (round 5): the four contexts' streams at DIFFERENT priorities (variants/libmot_prio.so reads MOT_STREAM_PRIORITY in mot_create; needs
# tools/experiments/stream_priority.patch applied), so that they stop doing their streaming kernels — and then their latency-bound ones — all at once
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
run() { MOT_BENCH_LIB=$2 MOT_PRIO_PATTERN=$3 timeout 300 python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline --no-all-outputs 2>/dev/null | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-24s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2; do
  run product "" ""
  run prio_-1,0,0,1 $PWD/variants/libmot_prio.so -1,0,0,1
  run prio_-1,-1,1,1 $PWD/variants/libmot_prio.so -1,-1,1,1
  run prio_-1,0,1,1 $PWD/variants/libmot_prio.so -1,0,1,1
  run prio_all_high $PWD/variants/libmot_prio.so -1,-1,-1,-1
done | tee gpurun_out/prio_ab.txt
