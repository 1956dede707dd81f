cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/s22; mkdir -p $O
# SQ counter passes of the final build (one context, 512 frames per launch, 12 frames): two passes, no trace domains beside --kernel-trace
pass() { # tag, counters...
  tag=$1; shift
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$tag -o p -- python bench.py --steps 1 --warmup 0 --frames 12 --no-aux --no-cpu-baseline --batch 512 --contexts 1 > $O/pmc_$tag.log 2>&1
  echo "$tag rc=$?"
}
pass insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass cycles SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
python profiles/summarize_pmc.py $(find $O/pmc_insts $O/pmc_cycles -name '*.db') --json=$O/pmc_sq.json > $O/pmc_sq_raw.txt 2>&1; tail -20 $O/pmc_sq_raw.txt
rm -rf $O/pmc_insts $O/pmc_cycles
# randomised tracker sequences and irregular clouds on the real kernels
timeout 600 python tests/explore_gpu.py 300 40 2>&1 | tail -8 | tee $O/explore_gpu.txt
