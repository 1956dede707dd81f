cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s27; mkdir -p $O
# review item 2a, literally: FETCH_SIZE of the streaming kernels with FOUR contexts, in lock-step (--phase 0) and on disjoint frames (--phase 3 of 12)
for ph in 0 3; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_ph$ph -o p -- python bench.py --steps 1 --warmup 0 --frames 12 --phase $ph --no-aux --no-cpu-baseline > $O/bench_ph$ph.json 2> $O/pmc_ph$ph.log
  echo "phase $ph rc=$?"; head -c 200 $O/bench_ph$ph.json; echo
  python profiles/summarize_pmc.py $(find $O/pmc_ph$ph -name '*.db') 2>&1 | grep -v "at::native\|rocprim\|^void" | head -8 | tee $O/fetch_ph$ph.txt
  rm -rf $O/pmc_ph$ph
done
# the same two runs without counters, for the rate the contexts reach when nothing serialises them
for ph in 0 3; do timeout 300 python bench.py --steps 1 --warmup 0 --frames 12 --phase $ph --no-aux --no-cpu-baseline 2>/dev/null | head -c 200; echo; done | tee $O/bench_plain.txt
