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


def sq_table(table):
    """the derived per-wave view of the SQ passes (columns as in r02/r03_pmc_sq_B512.txt)"""
    rows = []
    for k, v in table.items():
        w = v.get("SQ_WAVES")
        if k.startswith("void") or not w or "SQ_WAVE_CYCLES" not in v or "SQ_INSTS_VALU" not in v:
            continue
        wc = v["SQ_WAVE_CYCLES"]
        rows.append((k, v["SQ_INSTS_VALU"] / w, v.get("SQ_INSTS_SALU", 0) / w, v.get("SQ_INSTS_LDS", 0) / w,
                     (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / w, 4 * wc / w,
                     100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
                     v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_BUSY_CYCLES", 1), 1)))
    if rows:
        print()
        print(f"{'kernel':<34}{'VALU/wave':>10}{'SALU/wave':>10}{'LDS/wave':>9}{'VMEM/wave':>10}{'cycles/wave':>12}{'active %':>9}{'waiting %':>10}{'stall %':>8}{'VALU load':>10}")
        for r in sorted(rows, key=lambda r: -r[5]):
            print(f"{r[0]:<34}{r[1]:>10.0f}{r[2]:>10.0f}{r[3]:>9.0f}{r[4]:>10.1f}{r[5]:>12.0f}{r[6]:>9.0f}{r[7]:>10.0f}{r[8]:>8.0f}{r[9]:>10.2f}")


if __name__ == "__main__":
    t = main([a for a in sys.argv[1:] if not a.startswith("--json=")])
    sq_table(t)
    for a in sys.argv[1:]:
        if a.startswith("--json="):
            json.dump(t, open(a[7:], "w"), indent=1, sort_keys=True)
