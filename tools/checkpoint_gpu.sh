cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d gpurun_out/prof_pmc1 -o p1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 128 --contexts 1 > gpurun_out/prof_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -d gpurun_out/prof_pmc2 -o p2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 128 --contexts 1 > gpurun_out/prof_pmc2.log 2>&1
ls gpurun_out/prof_kt gpurun_out/prof_pmc1 gpurun_out/prof_pmc2
