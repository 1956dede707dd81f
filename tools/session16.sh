# round 3, GPU session 16: rectangle kernel after the DPP / scalar-register changes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s16; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python tools/time_rect.py 2>&1 | grep -v amdgpu.ids | tee $O/time_rect.txt
timeout 400 python tools/time_kernels.py 512 33,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
