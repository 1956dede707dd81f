cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s25; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; head -c 300 $O/bench_default.json; echo
for r in 1 2; do timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | head -c 230; echo; done | tee $O/bench.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/bench_under_rocprof_1ctx.json 2> $O/prof_1ctx.log
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -22
rm -rf $O/prof_1ctx
