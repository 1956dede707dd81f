# end-of-milestone GPU checkpoint: parity tests, smoke, the default bench line (with the CPU baseline) and a kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests -q -m gpu 2>&1 | tail -8     # no -x: see every failure of a first run on new code
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.json
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
ls gpurun_out/prof_kt
timeout 300 python tests/explore_gpu.py 60 10 2>&1 | tail -8          # randomised tracker sequences / irregular clouds on the real kernels
