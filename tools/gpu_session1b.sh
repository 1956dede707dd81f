cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s1
timeout 300 python tools/tracker_load.py > gpurun_out/s1/tracker_load.txt 2>&1; cat gpurun_out/s1/tracker_load.txt | grep -v amdgpu.ids
timeout 600 python bench.py > gpurun_out/s1/bench_default.json 2> gpurun_out/s1/bench_default.err; tail -c 300 gpurun_out/s1/bench_default.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s1/prof_4ctx -o kt -- python bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline > gpurun_out/s1/prof_4ctx.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s1/prof_1ctx -o kt -- python bench.py --steps 2 --warmup 1 --batch 128 --contexts 1 --no-aux --no-cpu-baseline > gpurun_out/s1/prof_1ctx.log 2>&1
python profiles/summarize_rocpd.py gpurun_out/s1/prof_4ctx/kt_results.db > gpurun_out/s1/kernel_trace_4ctx.txt 2>&1; head -30 gpurun_out/s1/kernel_trace_4ctx.txt
python profiles/summarize_rocpd.py gpurun_out/s1/prof_1ctx/kt_results.db > gpurun_out/s1/kernel_trace_1ctx.txt 2>&1; head -30 gpurun_out/s1/kernel_trace_1ctx.txt
rm -rf gpurun_out/s1/prof_4ctx gpurun_out/s1/prof_1ctx
