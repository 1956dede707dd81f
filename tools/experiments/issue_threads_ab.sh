cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
run() { timeout 200 python bench.py --steps 6 --warmup 1 --no-aux --no-cpu-baseline --no-all-outputs --issue-threads $1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('issue_threads=$1 %9.0f frames/s  %8.2f ms/step  host_issue %s' % (d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step')))
"; }
for r in 1 2; do run 0; run 1; done | tee gpurun_out/issue_threads_ab.txt
