cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s34; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.err; head -c 250 $O/bench_default.json; echo
