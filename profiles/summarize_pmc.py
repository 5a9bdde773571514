"""Per-kernel averages of PMC counters from rocprofv3 rocpd databases.
usage: python profiles/summarize_pmc.py <db> [<db> ...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute(
        "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
        "group by kernel_name, counter_name order by avg(value)*count(*) desc").fetchall()
    print(f"## {path}")
    print("| kernel | counter | dispatches | avg value | avg duration us |")
    print("|---|---|---|---|---|")
    for k, c, n, v, d in rows[:14]:
        print(f"| `{k[:70]}` | {c} | {n} | {v:.4g} | {d/1e3:.1f} |")
