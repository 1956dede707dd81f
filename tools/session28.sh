cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s28; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py tests/test_api_v2_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 400 python tools/time_kernels.py 512 30,34,31,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
for r in 1 2; do timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | head -c 230; echo; done | tee $O/bench.txt
