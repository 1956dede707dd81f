# One gpurun call = one invocation of this script:  gpurun --timeout S -- 'bash tools/gpu_session.sh NAME STEP [STEP …]'
# Outputs go to gpurun_out/NAME/. Every step runs under its own `timeout`. A step is `name` or `name:arg[:arg…]`:
#
#   tests[:pytest args]        pytest -m gpu (default: the whole suite, no -x)
#   smoke                      __graft_entry__.smoke()
#   bench[:tag[:bench args]]   python bench.py <args> > bench_<tag>.json        (tag default: "default")
#   quick[:tag[:bench args]]   bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline <args>, twice
#   skip[:m1,m2,...]           bench.py with MOT_BENCH_SKIP masks (mot_debug_skip_kernels): 256 * k = k extra EMPTY launches per sequence (the cost of a launch boundary; 0 = the product)
#   timeline                   tools/timeline_stats.py on the timed region of a 4-step four-context run under rocprofv3 --kernel-trace
#   traceo:ORDER               trace1 on firing / random point order (bench.py --point-order)
#   trace1 / trace4            rocprofv3 --kernel-trace --stats of one context alone (B 512) / of the default 4-context line
#   pmc[:CTR,CTR…]             one rocprofv3 --pmc pass per counter (default FETCH_SIZE,WRITE_SIZE), summarised
#   pmcsq                      two multi-counter SQ passes (instruction mix; wave / wait / stall cycles), raw means + the per-wave table
#   kernels:ids[:B]            tools/time_kernels.py B ids  (isolated kernel timings, product and variants/)
#   tracker[:streams]          tools/tracker_load.py
#   ab[:steps[:rounds]]        tools/bench_variants_ab.sh (product vs prebuilt variants/, interleaved)
#   py:script.py[:args]        python <script> <args>
#   sh:command                 anything else (colons allowed)
#
# (round 3's calls were 39 one-off scripts: profiles/r03_sessions.md)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; export TMPDIR=/tmp
NAME=$1; shift; O=gpurun_out/$NAME; mkdir -p $O
trace() {   # $1 tag, rest: bench args
  local tag=$1; shift
  timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o kt -- python bench.py --no-aux --no-cpu-baseline --no-all-outputs "$@" > $O/bench_under_rocprof_$tag.json 2> $O/prof_$tag.log
  python profiles/summarize_rocpd.py $O/prof_$tag/kt_results.db --tail=60 > $O/kernel_trace_$tag.txt 2>&1; grep -v "at::native\|rocprim" $O/kernel_trace_$tag.txt | head -24
  rm -rf $O/prof_$tag
}
for step in "$@"; do
  IFS=: read -r what a1 a2 <<< "$step"
  echo "=== $step"
  case $what in
    tests) timeout 1700 python -X faulthandler -m pytest ${a1:-tests} -q -m gpu -p no:cacheprovider > $O/pytest_gpu_full.txt 2>&1; grep -n "Fatal Python error\|Segmentation" -A12 $O/pytest_gpu_full.txt | head -40; tail -40 $O/pytest_gpu_full.txt | tee $O/pytest_gpu.txt ;;
    smoke) timeout 180 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt ;;
    bench) timeout 900 python bench.py $a2 > $O/bench_${a1:-default}.json 2> $O/bench_${a1:-default}.err; tail -c 300 $O/bench_${a1:-default}.err; head -c 700 $O/bench_${a1:-default}.json; echo ;;
    quick) for r in 1 2; do timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline $a2 2>/dev/null | tee -a $O/quick_${a1:-default}.jsonl | head -c 260; echo; done ;;
    trace1) trace B512_1ctx --steps 3 --warmup 1 --batch 512 --contexts 1 ;;
    trace4) trace B2048_4ctx ;;
    skip) for m in $(echo ${a1:-0,256,768,1536} | tr , ' '); do for r in 1 2; do MOT_BENCH_SKIP=$m timeout 300 python bench.py --steps 8 --warmup 1 --no-aux --no-cpu-baseline --no-all-outputs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('skip mask %3d  %9.0f frames/s  %8.2f ms/step' % ($m, d['value'], d['ms_per_step']))" | tee -a $O/skip.txt; done; done ;;
    timeline) timeout -k 5 600 rocprofv3 --kernel-trace -d $O/prof_tl -o kt -- python bench.py --steps 4 --warmup 1 --no-aux --no-cpu-baseline --no-all-outputs > $O/bench_under_rocprof_tl.json 2> $O/prof_tl.log
              python tools/timeline_stats.py $(find $O/prof_tl -name 'kt_results.db' | head -1) 0.42 0.96 | tee $O/timeline_stats.txt; rm -rf $O/prof_tl ;;
    traceo) trace B512_1ctx_order_$a1 --steps 3 --warmup 1 --batch 512 --contexts 1 --point-order $a1 ;;
    pmc) for ctr in $(echo ${a1:-FETCH_SIZE,WRITE_SIZE} | tr , ' '); do
           timeout -k 5 170 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$ctr -o p -- python bench.py --steps 1 --warmup 0 --frames 12 --no-aux --no-cpu-baseline --no-all-outputs --batch 512 --contexts 1 > $O/pmc_$ctr.log 2>&1; echo "$ctr rc=$?"
         done
         python profiles/summarize_pmc.py $(for ctr in $(echo ${a1:-FETCH_SIZE,WRITE_SIZE} | tr , ' '); do echo $O/pmc_$ctr/p_results.db; done) --json=$O/pmc_B512.json > $O/pmc_B512.txt 2>&1
         grep -v "at::native\|rocprim" $O/pmc_B512.txt | head -20; rm -rf $O/pmc_*/ ;;
    pmcsq) sqpass() { local tag=$1; shift
             timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o p -- python bench.py --steps 1 --warmup 0 --frames 12 --no-aux --no-cpu-baseline --no-all-outputs --batch 512 --contexts 1 > $O/pmc_$tag.log 2>&1; echo "$tag rc=$?"; }
           sqpass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
           sqpass cycles SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
           python profiles/summarize_pmc.py $(find $O/pmc_insts $O/pmc_cycles -name '*.db') --json=$O/pmc_sq_B512.json > $O/pmc_sq_B512.txt 2>&1
           grep -v "at::native\|rocprim" $O/pmc_sq_B512.txt | tail -18; rm -rf $O/pmc_insts $O/pmc_cycles ;;
    kernels) timeout 500 python tools/time_kernels.py ${a2:-512} $a1 2>&1 | grep -v "amdgpu.ids\|^stream" | tee $O/time_kernels.txt ;;
    tracker) timeout 400 python tools/tracker_load.py ${a1:-128} 2>&1 | grep -v amdgpu.ids | tee $O/tracker_load.txt ;;
    ab) bash tools/bench_variants_ab.sh ${a1:-8} ${a2:-2} 2>&1 | tee $O/ab.txt ;;
    py) timeout 900 python $a1 $a2 2>&1 | grep -v amdgpu.ids | tee $O/$(basename $a1 .py).txt ;;
    sh) timeout 900 bash -c "${step#sh:}" 2>&1 | tee -a $O/sh.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
