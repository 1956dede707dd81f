cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s3
for cfg in "512 4" "512 8" "256 4" "256 8" "128 4" "128 8" "64 4" "64 8" "32 4"; do set -- $cfg
  timeout 300 python bench.py --steps 3 --warmup 1 --batch $1 --contexts $2 --no-aux --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; fl=d['config']['frames_per_launch']
print('batch $1 ctx $2 frames/launch', fl, 'value', d['value'], 'K3 in-run us/frame', round(r['kernel_ms_in_run']['mean']*1e3/fl,3), 'min', round(r['kernel_ms_in_run']['min']*1e3/fl,3), 'pipeline_frac', r['pipeline_frac'], 'host_issue/step', d['host_issue_ms_per_step'], 'step', d['ms_per_step'])" | tee -a gpurun_out/s3/sweep.txt
done
