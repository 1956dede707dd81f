# one rocprofv3 counter pass per invocation (the guide: separate --pmc passes; never combined with the trace domains gpurun refuses)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ctr in "$@"; do
  timeout -k 5 170 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_$ctr -o p -- python bench.py --steps 1 --warmup 0 --frames 12 --no-aux --no-cpu-baseline --batch 512 --contexts 1 > gpurun_out/pmc_$ctr.log 2>&1
  echo "$ctr rc=$?"; ls gpurun_out/pmc_$ctr 2>/dev/null
done
