# round 3, GPU session 6: box-stage kernels after the pixel-run change (isolated timings of product + variants, phase clocks, box parity tests, 1-context trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s6; mkdir -p $O
timeout 600 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_box.txt
timeout 400 python tools/time_kernels.py 512 30,34,31,33,2 2>&1 | grep -v amdgpu.ids | tee $O/time_kernels.txt
timeout 300 python tools/time_gather.py 2>&1 | grep -v amdgpu.ids | tee $O/time_gather.txt
timeout 300 python tools/time_b1.py index 2>&1 | grep -v amdgpu.ids | tee $O/time_b1_index.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/prof_1ctx.log 2>&1
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -24
rm -rf $O/prof_1ctx
