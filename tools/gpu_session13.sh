cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s13
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "pipeline_frac", r["pipeline_frac"], "host_issue/step", d["host_issue_ms_per_step"], "threads", d["issue_threads"])'
timeout 500 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | python -c "$P" default | tee -a gpurun_out/s13/sweep.txt
timeout 500 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --force-gather > gpurun_out/s13/fg.out 2>gpurun_out/s13/fg.err; wc -l gpurun_out/s13/fg.out; python -c "$P" force_gather < gpurun_out/s13/fg.out | tee -a gpurun_out/s13/sweep.txt
timeout 500 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --force-gather --issue-threads 1 2>/dev/null | python -c "$P" force_gather_threads | tee -a gpurun_out/s13/sweep.txt
