cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s4
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -5
for v in "" "-DMOT_TRACK_ITEM_WAVES=1" "-DMOT_TRACK_ITEM_WAVES=4" "-DMOT_PREDICT_WAVES=2 -DMOT_UPDATE_WAVES=2" "-DMOT_PREDICT_WAVES=4 -DMOT_UPDATE_WAVES=3 -DMOT_TRACK_ITEM_WAVES=1"; do
  timeout 300 python tools/tracker_load.py 128 $v 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s4/tracker_load.txt
done
timeout 600 python bench.py > gpurun_out/s4/bench_default.json 2> gpurun_out/s4/bench_default.err; tail -c 300 gpurun_out/s4/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/s4/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_in_run'], d['roofline']['pipeline_frac']); print(d['roofline'].get('kernel_ms_isolated')); print(d['tracker_stress']); print(d['host_boundary_pipelined']['value'])"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s4/prof_4ctx -o kt -- python bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline > gpurun_out/s4/prof_4ctx.log 2>&1
python profiles/summarize_rocpd.py gpurun_out/s4/prof_4ctx/kt_results.db > gpurun_out/s4/kernel_trace_4ctx.txt 2>&1; head -18 gpurun_out/s4/kernel_trace_4ctx.txt
rm -rf gpurun_out/s4/prof_4ctx
