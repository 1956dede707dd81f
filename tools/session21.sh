cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s21; mkdir -p $O
# split launch sequence (ground stage on its own stream, the rest on a high-priority stream): parity first, then A/B
for m in 1 2; do
  MOT_SPLIT_STREAMS=$m timeout 600 python -m pytest tests/test_sequence_gpu.py tests/test_api_v2_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest_split$m.txt
done
B="python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline"
run() { # name, env...
  name=$1; shift
  line=$(env "$@" timeout 600 $B 2>$O/err_$name.txt | tail -1)
  python - "$name" "$line" <<'PY' | tee -a $O/ab.txt
import json,sys
try:
    d=json.loads(sys.argv[2]); print(f"{sys.argv[1]:28s} {d['value']:10.0f} frames/s  {d['ms_per_step']:8.2f} ms/step  host_cpu {d.get('host_cpu_ms_per_step')}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e, sys.argv[2][:200])
PY
}
for rep in 1 2; do
  run product X=0
  run product_q12 GPU_MAX_HW_QUEUES=12
  run split1 MOT_SPLIT_STREAMS=1 GPU_MAX_HW_QUEUES=12
  run split2 MOT_SPLIT_STREAMS=2 GPU_MAX_HW_QUEUES=12
  run split2_lowA MOT_SPLIT_STREAMS=2 MOT_SPLIT_PRIO=1,-1 GPU_MAX_HW_QUEUES=12
  run split2_noprio MOT_SPLIT_STREAMS=2 MOT_SPLIT_PRIO=0,0 GPU_MAX_HW_QUEUES=12
done
# N = 2 dry run on the one device (gloo): spawn, sharding, gather, JSON line
timeout 900 python bench.py --gpus 2 --shared-gpu-dryrun --batch 512 --contexts 2 --steps 2 --warmup 1 --no-aux --no-cpu-baseline > $O/dryrun_n2.json 2> $O/dryrun_n2.err
echo "dryrun rc=$?" | tee -a $O/ab.txt; tail -c 600 $O/dryrun_n2.json; tail -5 $O/dryrun_n2.err
