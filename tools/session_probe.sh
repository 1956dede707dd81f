cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 200 python tools/check_gather_gpu.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -3
