# round 3, GPU session 9: full parity suite, tracker load at the bench's 512 streams (2 vs 3 waves per SIMD in the update kernel), default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s9; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.txt
timeout 300 python tools/tracker_load.py 512 2>&1 | grep -v amdgpu.ids | tee $O/tracker_load_512_product.txt
timeout 300 python tools/tracker_load.py 512 lib=$PWD/variants/libmot_uw3.so 2>&1 | grep -v amdgpu.ids | tee $O/tracker_load_512_uw3.txt
run() { MOT_BENCH_LIB=$2 timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2> $O/err_$1.txt | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-12s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2 3; do run product ""; run uw3 $PWD/variants/libmot_uw3.so; done | tee $O/ab.txt
