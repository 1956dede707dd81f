# bench.py (short, no auxiliary lines) on the product library and on every prebuilt variant in variants/ (tools/prebuild.py):
#   bash tools/bench_variants.sh [steps] [extra bench args...]        -> gpurun_out/variants.txt
cd ${GRAFT_REPO_ROOT:-.}; S=${1:-5}; shift; mkdir -p gpurun_out
run() { # name, lib ('' = product)
  MOT_BENCH_LIB=$2 timeout 300 python bench.py --steps $S --warmup 1 --no-aux --no-cpu-baseline "${@:3}" 2> gpurun_out/variant_$1.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('%-28s %9.0f frames/s  %8.2f ms/step  K3 solo %.4f ms  pipeline_frac %.4f' % ('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['mean'], d['roofline']['pipeline_frac']))
except Exception as e:
    print('$1 failed', e)
"
}
{ run product "" "$@"
  for f in variants/libmot_*.so; do [ -e "$f" ] || continue; n=$(basename $f .so); run ${n#libmot_} $PWD/$f "$@"; done
  run product_again "" "$@"; } | tee gpurun_out/variants.txt
