# interleaved A/B: every library of variants/ and the product, `rounds` times in the same order, `steps` timed steps each
#   bash tools/bench_variants_ab.sh [steps] [rounds]      -> gpurun_out/variants_ab.txt
cd ${GRAFT_REPO_ROOT:-.}; S=${1:-10}; R=${2:-3}; mkdir -p gpurun_out
run() { MOT_BENCH_LIB=$2 timeout 300 python bench.py --steps $S --warmup 1 --no-aux --no-cpu-baseline 2> gpurun_out/variant_$1.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-20s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in $(seq $R); do
  run product ""
  for f in variants/libmot_*.so; do n=$(basename $f .so); run ${n#libmot_} $PWD/$f; done
done | tee gpurun_out/variants_ab.txt
