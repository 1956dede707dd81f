cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s39; mkdir -p $O
timeout 900 python -m pytest tests/test_cluster_box_gpu.py tests/test_property_gpu.py tests/test_sequence_gpu.py tests/test_api_v2_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest.txt
timeout 400 python tools/time_kernels.py 512 31,33,2,100 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt
bash tools/bench_variants_ab.sh 8 2 2>&1 | tee $O/ab.txt
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/bench_under_rocprof_1ctx.json 2> $O/prof_1ctx.log
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep "gather\|cluster_rect\|cluster_index\|label_stats\|finalize" $O/kernel_trace_B512_1ctx.txt
rm -rf $O/prof_1ctx
