cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s23; mkdir -p $O
timeout 400 python tools/time_kernels.py 512 30,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
