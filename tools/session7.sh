# round 3, GPU session 7: full parity suite (bounded track slots, wide clusters, packed gather), default bench, 1-context trace, 200 k-point line, forced gather
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s7; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'parity', {k:v for k,v in d['parity_check'].items() if k in ('masks_boxes_bit_exact','track_sets_equal','states_within_1e-4','max_rel_state_err','frames')})
print('single', d['single_stream']['latency_ms'], d['single_stream'].get('kernel_chain_us'))
print('stress', {k:(v['us_per_launch'], v['stream0_vs_oracle_after_all_frames']) for k,v in d['tracker_stress'].items()})
print('iso', d['roofline'].get('kernel_ms_isolated'))"
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_1ctx -o kt -- python bench.py --steps 3 --warmup 1 --batch 512 --contexts 1 --no-aux --no-cpu-baseline > $O/prof_1ctx.log 2>&1
python profiles/summarize_rocpd.py $O/prof_1ctx/kt_results.db > $O/kernel_trace_B512_1ctx.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_B512_1ctx.txt | head -24
rm -rf $O/prof_1ctx
timeout 600 python bench.py --points 200000 --no-aux --no-cpu-baseline > $O/bench_200k.json 2> $O/bench_200k.err; tail -c 300 $O/bench_200k.err; head -c 700 $O/bench_200k.json; echo
timeout 600 python bench.py --force-gather --no-aux --no-cpu-baseline > $O/bench_force_gather.json 2> $O/bench_force_gather.err; tail -c 300 $O/bench_force_gather.err; head -c 400 $O/bench_force_gather.json; echo
timeout 600 python bench.py --no-aux --no-cpu-baseline > $O/bench_nogather.json 2> /dev/null; head -c 400 $O/bench_nogather.json; echo
