# round 3, GPU session 4: phase clocks of the box-stage kernels and the labelling kernel (timing-instrumented variant builds)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s4; mkdir -p $O
timeout 300 python tools/time_b1.py 2>&1 | grep -v amdgpu.ids | tee $O/time_b1.txt
timeout 300 python tools/time_b1.py index 2>&1 | grep -v amdgpu.ids | tee $O/time_b1_index.txt
timeout 300 python tools/time_gather.py 2>&1 | grep -v amdgpu.ids | tee $O/time_gather.txt
timeout 300 python tools/time_ccl.py 2>&1 | grep -v amdgpu.ids | tee $O/time_ccl.txt
