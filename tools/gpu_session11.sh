cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s11
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
timeout 1200 python tools/ablate.py 10,11,12,21,30,34,31,33,32 baseline 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s11/kernels.txt
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(sys.argv[1], d["value"], "ms/step", d["ms_per_step"], "K3 in-run", r["kernel_ms_in_run"]["mean"], "pipeline_frac", r["pipeline_frac"], "host_issue/step", d["host_issue_ms_per_step"])'
timeout 300 python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline 2>/dev/null | python -c "$P" b512c4 | tee -a gpurun_out/s11/sweep.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --batch 1024 --contexts 4 2>/dev/null | python -c "$P" b1024c4 | tee -a gpurun_out/s11/sweep.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --batch 768 --contexts 3 2>/dev/null | python -c "$P" b768c3 | tee -a gpurun_out/s11/sweep.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --batch 1024 --contexts 8 2>/dev/null | python -c "$P" b1024c8 | tee -a gpurun_out/s11/sweep.txt
timeout 300 python bench.py --steps 3 --warmup 1 --no-aux --no-cpu-baseline --force-gather 2>/dev/null | python -c "$P" b512c4_gather | tee -a gpurun_out/s11/sweep.txt
