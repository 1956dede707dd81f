cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s9
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "K3 in-run", r["kernel_ms_in_run"]["mean"], "pipeline_frac", r["pipeline_frac"], "host_issue/step", d["host_issue_ms_per_step"])'
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | python -c "$P" threads4ctx | tee -a gpurun_out/s9/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --batch 768 --contexts 6 2>/dev/null | python -c "$P" threads6ctx768 | tee -a gpurun_out/s9/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --batch 384 --contexts 6 2>/dev/null | python -c "$P" threads6ctx384 | tee -a gpurun_out/s9/sweep.txt
timeout 300 python bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline --force-gather > gpurun_out/s9/force_gather.out 2> gpurun_out/s9/force_gather.err; wc -l gpurun_out/s9/force_gather.out; cut -c1-300 gpurun_out/s9/force_gather.out
timeout 1200 python tools/ablate.py 10,11,12,21,30,34,31,33,32 baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s9/kernels.txt
