# round 3, GPU session 5: do the streaming kernels need every wave slot? (dynamic-LDS pad caps their workgroups per CU, leaving slots to other contexts' kernels)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s5; mkdir -p $O
run() { MOT_K1_LDS_PAD=$2 MOT_K3_LDS_PAD=$3 timeout 300 python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline $4 2> $O/err_$1.txt | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline()); print('%-34s %9.0f frames/s  %8.2f ms/step  K3 solo %.1f us' % ('$1 $4', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['mean']*1e3))
except Exception as e:
    print('$1 failed', e)
"; }
for r in 1 2; do
  run "k1x8_k3x3(default)" 0 0 ""
  run "k1x6_k3x3" 8192 0 ""
  run "k1x4_k3x3" 14336 0 ""
  run "k1x4_k3x2" 14336 40960 ""
  run "k1x2_k3x2" 40960 40960 ""
  run "k1x8_k3x2" 0 40960 ""
  run "k1x2_k3x1" 40960 72000 ""
done | tee $O/lds_pad_sweep.txt
run "k1x4_k3x3" 14336 0 "--contexts 1 --batch 512" | tee -a $O/lds_pad_sweep.txt
run "k1x8_k3x3(default)" 0 0 "--contexts 1 --batch 512" | tee -a $O/lds_pad_sweep.txt
