cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s5; mkdir -p $O
timeout 600 python -m pytest tests/test_frame_kernel_gpu.py tests/test_cluster_box_gpu.py tests/test_property_gpu.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 300 python tools/time_k3_modes.py 512 "" variants/libmot_lf256.so variants/libmot_lf1024.so variants/libmot_lf512d2.so variants/libmot_lf512d8.so 2>&1 | grep -v amdgpu.ids | tee $O/k3_modes.txt
b() { n=$1; shift; timeout 300 python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline "$@" 2> $O/$n.err | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.readline())
    print('%-22s %9.0f frames/s  %8.2f ms/step  K3 solo %.4f in-run %.4f  pipeline_frac %.4f' % ('$n', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['mean'], d['roofline']['kernel_ms_in_timed_region']['mean'], d['roofline']['pipeline_frac']))
except Exception as e:
    print('$n failed', e)
"; }
{ b chunk --compaction chunk; b frame --compaction frame; b chunk2 --compaction chunk; b frame2 --compaction frame; } | tee $O/bench_modes.txt
