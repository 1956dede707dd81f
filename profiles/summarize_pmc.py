"""per-kernel mean of the counters in one or more rocprofv3 --pmc rocpd databases (one counter pass each)"""
import sqlite3
import sys
import json


def collect(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name").fetchall()
    return rows


def main(paths):
    table = {}
    for p in paths:
        for name, ctr, avg, n in collect(p):
            k = name.split("(")[0]
            if k.startswith("__amd"):
                continue
            table.setdefault(k, {})[ctr] = avg
            table[k]["dispatches"] = n
    ctrs = sorted({c for v in table.values() for c in v if c != "dispatches"})
    print(f"{'kernel':<28}" + "".join(f"{c:>16}" for c in ctrs) + f"{'dispatches':>12}")
    for k, v in sorted(table.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        print(f"{k:<28}" + "".join(f"{v.get(c, float('nan')):>16.1f}" for c in ctrs) + f"{v['dispatches']:>12}")
    return table


if __name__ == "__main__":
    t = main([a for a in sys.argv[1:] if not a.startswith("--json=")])
    for a in sys.argv[1:]:
        if a.startswith("--json="):
            json.dump(t, open(a[7:], "w"), indent=1, sort_keys=True)
