cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s31; mkdir -p $O
timeout 300 python tools/time_gather.py 2>&1 | grep -v amdgpu.ids | tee $O/gather_phases.txt
