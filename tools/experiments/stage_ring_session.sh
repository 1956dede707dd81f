# GPU session: mot_ring parity + bench with stage rings + prebuilt block-size variants + the default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s1; mkdir -p $O
timeout 600 python -m pytest tests/test_ring_gpu.py tests/test_abi.py -q 2>&1 | tail -4 | tee $O/pytest_ring.txt
b() { n=$1; shift; timeout 300 python bench.py --steps 5 --warmup 1 --no-aux --no-cpu-baseline "$@" 2> $O/$n.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('%-22s %9.0f frames/s  %8.2f ms/step  K3 solo %.4f in-run %.4f  pipeline_frac %.4f' % ('$n', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['mean'], d['roofline']['kernel_ms_in_timed_region']['mean'], d['roofline']['pipeline_frac']))
except Exception as e:
    print('$n failed', e)
"; }
{ b free_b2048c4
  b ring2_b1024c2 --ring 2 --contexts 2 --batch 1024
  b ring2_b2048c2 --ring 2 --contexts 2 --batch 2048
  b ring2_b2048c4 --ring 2 --contexts 4
  b ring4_b2048c4 --ring 4 --contexts 4
  b ring3_b1536c3 --ring 3 --contexts 3 --batch 1536
  b free_b1024c2 --contexts 2 --batch 1024; } | tee $O/ring.txt
bash tools/bench_variants.sh 5 > /dev/null; cp gpurun_out/variants.txt $O/
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; python -c "
import json; d = json.load(open('$O/bench_default.json')); print(d['value'], d['roofline']['frac'], d['roofline']['pipeline_frac']); print(json.dumps(d.get('single_stream'))); print(json.dumps(d.get('tracker_stress')))"
