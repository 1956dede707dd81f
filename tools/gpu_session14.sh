cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s14
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "pipeline_frac", r["pipeline_frac"], "host_issue/step", d["host_issue_ms_per_step"])'
for cfg in "2048 4" "3072 6" "4096 8" "2048 8" "1536 3"; do set -- $cfg
timeout 600 python bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline --batch $1 --contexts $2 2>/dev/null | python -c "$P" b$1c$2 | tee -a gpurun_out/s14/sweep.txt
done
