# round 3, GPU session 17: rectangle kernel's caliper walk on LDS broadcast reads instead of readlanes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s17; mkdir -p $O
timeout 300 python tools/time_rect.py $PWD/variants/dbg_libmot_rect_lds.so 2>&1 | grep -v amdgpu.ids | tee $O/time_rect_lds.txt
timeout 400 python tools/time_kernels.py 512 33,2 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
