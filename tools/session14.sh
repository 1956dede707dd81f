# round 3, GPU session 14: gather kernel with the two-deep pipelined walk (product: 4 points per thread and trip; variants), index kernel with running positions (head = before both)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s14; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 400 python tools/time_kernels.py 512 30,31,34,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
timeout 300 python tools/time_gather.py 2>&1 | grep -v amdgpu.ids | tee $O/time_gather.txt
timeout 300 python tools/time_b1.py 2>&1 | grep -v amdgpu.ids | tee $O/time_b1_index.txt
for r in 1 2; do for v in product head; do L=""; [ $v = head ] && L=$PWD/variants/libmot_head.so; MOT_BENCH_LIB=$L timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$v %9.0f frames/s' % d['value'])"; done; done | tee $O/ab.txt
