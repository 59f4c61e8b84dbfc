#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into small text files kept under profiles/.
usage: rocpd_summary.py <results.db> [...]  -> prints a markdown summary (kernel stats + PMC sums)"""
import sqlite3
import sys

for f in sys.argv[1:]:
    db = sqlite3.connect(f); cur = db.cursor()
    print(f"## {f}\n")
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    if rows:
        print("| kernel | calls | total_ms | avg_ms | % |\n|---|---|---|---|---|")
        for n, c, t, a, p in rows[:8]:
            print(f"| {n.split('(')[1] if n.startswith('(anonymous') else n[:60]} | {c} | {t/1e3:.3f} | {a/1e3:.3f} | {p:.2f} |".replace("anonymous namespace)::", ""))
        print()
    try:
        q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
             "group by kernel_name, counter_name")
        rows = [r for r in cur.execute(q) if "lz4" in r[0] or "xxh32" in r[0] or "pack" in r[0] or "zstd" in r[0]]
        if rows:
            print("| kernel | counter | sum over dispatches | dispatches | per dispatch |\n|---|---|---|---|---|")
            for k, c, v, n in rows:
                k = k.split("::")[1].split("(")[0] if "::" in k else k.split("(")[0][:40]
                print(f"| {k} | {c} | {v:.0f} | {n} | {v/max(n,1):.0f} |")
            print()
    except sqlite3.OperationalError:
        pass
