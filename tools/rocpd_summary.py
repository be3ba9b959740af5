#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / average duration.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md
Also prints the PMC counters per kernel when the db was collected with --pmc."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# kernel trace summary of `{path}`\n")
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    print("|---|---|---|---|---|---|---|")
    for n, c, tot, avg, mn, mx in rows:
        n = n.split("(")[0]
        print(f"| `{n[:90]}` | {c} | {tot/1e6:.3f} | {avg/1e6:.4f} | {mn/1e6:.4f} | {mx/1e6:.4f} | {100*tot/total:.2f} |")
    try:
        pm = cur.execute("select * from counters_collection limit 1").fetchall()
        if pm:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            print("\n## PMC counters (sum over dispatches of the kernel)\n")
            kn = "kernel_name" if "kernel_name" in ccols else "name"
            rows = cur.execute(f"select {kn}, counter_name, sum(value), count(*) from counters_collection "
                               f"group by {kn}, counter_name order by 1, 2").fetchall()
            print("| kernel | counter | sum | dispatches |\n|---|---|---|---|")
            for n, cn, v, c in rows:
                print(f"| `{n.split('(')[0][:70]}` | {cn} | {v:.6g} | {c} |")
    except sqlite3.Error as e:
        print(f"\n(no PMC data: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
