cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s20; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
