cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s5
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/s5/prof_stress -o kt -- python tools/tracker_load.py 128 stress-only > gpurun_out/s5/prof_stress.log 2>&1
python - <<'PY'
import sqlite3
db = sqlite3.connect("gpurun_out/s5/prof_stress/kt_results.db")
rows = db.execute("select name, start, end-start from kernels where name like 'track_%' order by start").fetchall()
# three stress runs (T = 8, 32, 64), 40 steps each of 4 launches: print the mean of the last 20 steps of each
names = ["track_prep_kernel", "track_predict_kernel", "track_update_kernel", "track_finish_kernel"]
per = {n: [r[2] for r in rows if r[0].startswith(n)] for n in names}
for i, T in enumerate((8, 32, 64)):
    print("T=%d" % T, {n: round(sum(per[n][i * 40 + 20:(i + 1) * 40]) / 20 / 1e3, 1) for n in names})
PY
rm -rf gpurun_out/s5/prof_stress
bash tools/pmc_pass.sh FETCH_SIZE WRITE_SIZE 2>&1 | tail -4
python profiles/summarize_pmc.py gpurun_out/pmc_FETCH_SIZE/p_results.db gpurun_out/pmc_WRITE_SIZE/p_results.db --json=gpurun_out/s5/pmc_B128.json > gpurun_out/s5/pmc_B128.txt 2>&1; head -20 gpurun_out/s5/pmc_B128.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
