# round 3, GPU session 2: full parity suite, default bench (with parity_check), variants per kernel, PMC passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 | tee $O/pytest_gpu.txt
timeout 400 python tools/time_kernels.py 512 12,21,30,34,31,33,2,100 2>&1 | grep -v amdgpu.ids | tee $O/time_kernels.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err; python -c "
import json,sys
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('parity_check'), d['roofline']['frac'], d['roofline']['pipeline_frac'], d.get('single_stream'), d.get('tracker_stress'))
print(d['roofline'].get('kernel_ms_isolated')); print(d['cpu_baseline']['value'] if d.get('cpu_baseline') else None, d['cpu_baseline'].get('parallel') if d.get('cpu_baseline') else None)"
bash tools/pmc_pass.sh FETCH_SIZE WRITE_SIZE 2>&1 | tail -2
python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE/p_results.db gpurun_out/pmc_WRITE_SIZE/p_results.db --json=$O/pmc_B512.json > $O/pmc_B512.txt 2>&1; grep -v "at::native\|rocprim" $O/pmc_B512.txt | head -20
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
