cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s8
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "K3 in-run", r["kernel_ms_in_run"]["mean"], "pipeline_frac", r["pipeline_frac"], "host_issue/step", d["host_issue_ms_per_step"], "threads", d["issue_threads"])'
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | python -c "$P" threads4ctx | tee -a gpurun_out/s8/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --issue-threads 0 2>/dev/null | python -c "$P" single4ctx | tee -a gpurun_out/s8/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --contexts 8 2>/dev/null | python -c "$P" threads8ctx | tee -a gpurun_out/s8/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --contexts 2 2>/dev/null | python -c "$P" threads2ctx | tee -a gpurun_out/s8/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --batch 768 --contexts 6 2>/dev/null | python -c "$P" threads6ctx768 | tee -a gpurun_out/s8/sweep.txt
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --force-gather 2> gpurun_out/s8/force_gather.err | python -c "$P" threads4ctx_force_gather | tee -a gpurun_out/s8/sweep.txt; tail -3 gpurun_out/s8/force_gather.err
