# round-2 GPU session 1: parity suite, smoke, the new bench line, kernel traces (4 contexts and isolated), launch-size sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/s1/pytest.txt; cat gpurun_out/s1/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/s1/bench_default.json 2> gpurun_out/s1/bench_default.err; tail -c 600 gpurun_out/s1/bench_default.err; head -c 3000 gpurun_out/s1/bench_default.json; echo
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s1/prof_4ctx -o kt -- python bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline > gpurun_out/s1/prof_4ctx.log 2>&1; tail -c 300 gpurun_out/s1/prof_4ctx.log
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s1/prof_1ctx -o kt -- python bench.py --steps 2 --warmup 1 --batch 128 --contexts 1 --no-aux --no-cpu-baseline > gpurun_out/s1/prof_1ctx.log 2>&1; tail -c 300 gpurun_out/s1/prof_1ctx.log
for cfg in "256 8" "128 4" "128 8" "64 4"; do set -- $cfg
  timeout 300 python bench.py --steps 3 --warmup 1 --batch $1 --contexts $2 --no-aux --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $1 ctx $2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_in_run'], d['roofline']['pipeline_frac'], d['host_issue_ms_per_step'])"
done
ls gpurun_out/s1/prof_4ctx gpurun_out/s1/prof_1ctx
