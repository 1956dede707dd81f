cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s32; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
bash tools/bench_variants_ab.sh 8 3 2>&1 | tee $O/ab.txt
