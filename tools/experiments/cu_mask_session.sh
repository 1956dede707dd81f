# EXPERIMENT (round 5): every context's stream restricted to its own slice of the CUs (variants/libmot_cumask.so: MOT_CU_SLICE="i/n[/mode]" read by mot_create;
# mode 0 = a contiguous range of the 256 mask bits, 1 = interleaved) against the product, interleaved.   bash tools/cumask_experiment.sh
cd ${GRAFT_REPO_ROOT:-.}   # needs tools/experiments/cu_mask.patch applied and built as variants/libmot_cumask.so; mkdir -p gpurun_out
run() { MOT_BENCH_LIB=$2 MOT_CU_SLICE_MODE=$3 timeout 300 python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline --no-all-outputs 2>/dev/null | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-24s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2; do
  run product "" ""
  run cumask_contiguous $PWD/variants/libmot_cumask.so 0
  run cumask_interleaved $PWD/variants/libmot_cumask.so 1
done
