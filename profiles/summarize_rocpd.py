"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this prints the per-kernel summary (the `--stats` view)
as a text table so that it can be committed under profiles/.
   python profiles/summarize_rocpd.py kt_results.db [--tail=N]
--tail=N adds a second table over the LAST N launches of every kernel. bench.py ends (with --no-aux --no-cpu-baseline) with its
single-context pass — 60 frames on one context, nothing else on the GPU, in the SAME process and memory layout as the timed
region — so `--tail=60` of a trace of the default command is the rocprofv3 view of the launches `roofline.kernel_ms` times."""
import sqlite3
import sys

HDR = f"{'kernel':<58} {'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}  grid(x,y) wg vgpr sgpr lds"


def table(rows):
    tot = sum(r[2] for r in rows) or 1
    print(HDR)
    for r in rows:
        name = r[0].split("(")[0][:58]
        print(f"{name:<58} {r[1]:>6} {r[2]/1e3:>11.1f} {r[3]/1e3:>9.2f} {r[4]/1e3:>9.2f} {r[5]/1e3:>9.2f} {100*r[2]/tot:>6.2f}  "
              f"({r[6]},{r[7]}) {r[8]} {r[9]} {r[10]} {r[11]}")


def main(path, tail=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = ("name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
            "max(grid_x), max(grid_y), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(static_lds_size)")
    table(cur.execute(f"select {cols} from kernels group by name order by sum(duration) desc").fetchall())
    if tail:
        print(f"\nthe last {tail} launches of every kernel (bench.py's single-context pass when the run ends with it):")
        table(cur.execute(f"select {cols} from (select *, row_number() over (partition by name order by start desc) as rn from kernels) "
                          f"where rn <= {int(tail)} group by name order by sum(duration) desc").fetchall())


if __name__ == "__main__":
    t = [int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("--tail=")]
    main(sys.argv[1], t[0] if t else 0)
