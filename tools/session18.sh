# round 3, GPU session 18: hipGraph launch mode (parity, single-stream latency), full suite, default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s18; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'parity', {k:v for k,v in d['parity_check'].items() if k in ('masks_boxes_bit_exact','track_sets_equal','states_within_1e-4','max_rel_state_err','frames')})
s=d['single_stream']; print('single', s['latency_ms'], s['frames_per_s_back_to_back'], 'graphs', s.get('with_launch_graphs'), 'chain', s['kernel_chain_us']['sum'])
print('iso', d['roofline'].get('kernel_ms_isolated'))"
