cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s12
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "K3 solo", r["kernel_ms"]["mean"], "frac", r["frac"], "K3 shared", r["kernel_ms_in_timed_region"]["mean"], "pipeline_frac", r["pipeline_frac"], "host_issue/step", d["host_issue_ms_per_step"])'
for cfg in "512 4" "1024 4" "1024 2" "768 3" "1536 6" "2048 4"; do set -- $cfg
timeout 500 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --batch $1 --contexts $2 2>/dev/null | python -c "$P" b$1c$2 | tee -a gpurun_out/s12/sweep.txt
done
