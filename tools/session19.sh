# round 3, GPU session 19: full parity suite + the default bench line after the checker fix (the traces / PMC of tools/checkpoint_gpu.sh stand)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s19; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity_check'])"
