cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s26; mkdir -p $O
timeout 900 python tests/explore_gpu.py 300 40 2>&1 | tail -8 | tee $O/explore_gpu.txt
