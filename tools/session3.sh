# round 3, GPU session 3: Infinity-Cache probe; compaction kernel in reverse frame order (1 and 4 contexts), interleaved A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s3; mkdir -p $O
timeout 300 ./variants/mall_probe 2>&1 | tee $O/mall_probe.txt
run() { MOT_BENCH_LIB=$2 timeout 300 python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline $3 2> $O/variant_$1.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-24s %9.0f frames/s  %8.2f ms/step  K3 solo %.1f us' % ('$1 $3', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['mean']*1e3))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2; do
  run product "" ""
  run k3rev $PWD/variants/libmot_k3rev.so ""
  run product "" "--contexts 1 --batch 512"
  run k3rev $PWD/variants/libmot_k3rev.so "--contexts 1 --batch 512"
done | tee $O/k3rev_ab.txt
