"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this prints the per-kernel summary (the `--stats` view)
as a text table so that it can be committed under profiles/."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(grid_x), max(grid_y), max(workgroup_x), max(vgpr_count), max(sgpr_count), max(static_lds_size) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<58} {'calls':>6} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}  grid(x,y) wg vgpr sgpr lds")
    for r in rows:
        name = r[0].split("(")[0][:58]
        print(f"{name:<58} {r[1]:>6} {r[2]/1e3:>11.1f} {r[3]/1e3:>9.2f} {r[4]/1e3:>9.2f} {r[5]/1e3:>9.2f} {100*r[2]/tot:>6.2f}  "
              f"({r[6]},{r[7]}) {r[8]} {r[9]} {r[10]} {r[11]}")


if __name__ == "__main__":
    main(sys.argv[1])
