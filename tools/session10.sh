# round 3, GPU session 10: the persistent label kernel (product: cells + labels ahead, 8 waves/SIMD) and its variants; parity of the box stage; bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s10; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py tests/test_ground_gpu.py tests/test_api_v2_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_box.txt
timeout 600 python tools/time_kernels.py 512 30,34,2 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
timeout 600 python tools/time_kernels.py 1 30,34,2 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels_b1.txt
run() { MOT_BENCH_LIB=$2 timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2> $O/err_$1.txt | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-12s %9.0f frames/s  %8.2f ms/step' % ('$1', d['value'], d['ms_per_step']))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2; do run product ""; run lg16 $PWD/variants/libmot_lg16.so; run lw7 $PWD/variants/libmot_lw7.so; done | tee $O/ab.txt
