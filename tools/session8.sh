# round 3, GPU session 8: tracker tests after the update-kernel diet + bounded slots; tracker load (product = 2 waves/SIMD no scratch, uw3 = 3 waves with scratch); A/B bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s8; mkdir -p $O
timeout 1500 python -m pytest tests/test_tracker_gpu.py tests/test_cluster_box_gpu.py tests/test_sequence_gpu.py tests/test_gather_gpu.py -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.txt
timeout 300 python tools/tracker_load.py 128 2>&1 | grep -v amdgpu.ids | tee $O/tracker_load_product.txt
timeout 300 python tools/tracker_load.py 128 lib=$PWD/variants/libmot_uw3.so 2>&1 | grep -v amdgpu.ids | tee $O/tracker_load_uw3.txt
timeout 400 python tools/time_kernels.py 512 33,40 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
run() { MOT_BENCH_LIB=$2 timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2> $O/err_$1.txt | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-12s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2; do run product ""; run uw3 $PWD/variants/libmot_uw3.so; done | tee $O/ab.txt
