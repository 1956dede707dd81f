# end-of-milestone GPU checkpoint (round 3): parity tests, smoke, the default bench line, the kernel traces the bench line's roofline is
# checked against (4 contexts = the bench's timed region; 1 context = the kernels in isolation), PMC passes, the 200 k-point and forced-gather lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/ckpt; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt     # no -x: see every failure
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash tools/pmc_pass.sh FETCH_SIZE WRITE_SIZE 2>&1 | tail -2
python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE/p_results.db gpurun_out/pmc_WRITE_SIZE/p_results.db --json=$O/pmc_B512.json > $O/pmc_B512.txt 2>&1; grep -v "at::native\|rocprim" $O/pmc_B512.txt | head -18
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cp $O/pmc_B512.json profiles/r03_pmc_B512.json   # the bench line below quotes its traffic figure from here
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; head -c 600 $O/bench_default.json; echo
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_4ctx -o kt -- python bench.py --no-aux --no-cpu-baseline > $O/bench_under_rocprof_4ctx.json 2> $O/prof_4ctx.log
python profiles/summarize_rocpd.py $O/prof_4ctx/kt_results.db --tail=60 > $O/kernel_trace_B2048_4ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B2048_4ctx.txt | head -22
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/bench_under_rocprof_1ctx.json 2> $O/prof_1ctx.log
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -22
rm -rf $O/prof_4ctx $O/prof_1ctx
timeout 600 python bench.py --points 200000 --no-aux --no-cpu-baseline > $O/bench_200k.json 2> /dev/null; head -c 300 $O/bench_200k.json; echo
timeout 600 python bench.py --force-gather --no-aux --no-cpu-baseline > $O/bench_force_gather.json 2> /dev/null; head -c 300 $O/bench_force_gather.json; echo
timeout 600 python bench.py --no-aux --no-cpu-baseline > $O/bench_nogather.json 2> /dev/null; head -c 300 $O/bench_nogather.json; echo
timeout 300 python tools/tracker_load.py 128 2>&1 | grep -v amdgpu.ids | tee $O/tracker_load.txt
