# round 3, GPU session 12: filter / CCL / index kernels after the latency work: parity, isolated timings, 1-context trace, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s12; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py tests/test_ground_gpu.py tests/test_api_v2_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 400 python tools/time_kernels.py 512 11,21,34,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/prof_1ctx.log 2>&1
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -22
rm -rf $O/prof_1ctx
for r in 1 2; do timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | head -c 200; echo; done
