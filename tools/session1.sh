# round 3, GPU session 1: parity (incl. the new 154-frame sequence tests), smoke, bench default (phase 38) + lock-step (phase 0), 1-context trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s1; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err; head -c 900 $O/bench_default.json; echo
timeout 600 python bench.py --phase 0 --no-aux --no-cpu-baseline > $O/bench_phase0.json 2> $O/bench_phase0.err; head -c 500 $O/bench_phase0.json; echo
timeout 600 python bench.py --phase 38 --no-aux --no-cpu-baseline > $O/bench_phase38.json 2> $O/bench_phase38.err; head -c 500 $O/bench_phase38.json; echo
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/prof_1ctx.log 2>&1
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -24
rm -rf $O/prof_1ctx
