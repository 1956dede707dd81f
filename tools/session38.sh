cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s38; mkdir -p $O
timeout 400 python tools/time_kernels.py 512 10,12,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
bash tools/bench_variants_ab.sh 8 2 2>&1 | tee $O/ab.txt
