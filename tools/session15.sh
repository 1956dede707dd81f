# round 3, GPU session 15: rectangle kernel phase clocks; compaction kernel after batching its occupancy atomics (head = before); ground/box parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s15; mkdir -p $O
timeout 900 python -m pytest tests/test_ground_gpu.py tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py tests/test_api_v2_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python tools/time_rect.py 2>&1 | grep -v amdgpu.ids | tee $O/time_rect.txt
timeout 400 python tools/time_kernels.py 512 12,33,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
