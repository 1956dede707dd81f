# round 3, GPU session 11: CCL kernel after the run-balanced union pass (phase clocks, isolated time, parity), 1-context trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s11; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_box.txt
timeout 300 python tools/time_ccl.py 2>&1 | grep -v amdgpu.ids | tee $O/time_ccl.txt
timeout 400 python tools/time_kernels.py 512 21,30,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/prof_1ctx.log 2>&1
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -22
rm -rf $O/prof_1ctx
timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | head -c 300; echo
