cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s29; mkdir -p $O
timeout 400 python tools/time_kernels.py 512 31,2 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/bench_under_rocprof_1ctx.json 2> $O/prof_1ctx.log
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -22
rm -rf $O/prof_1ctx
