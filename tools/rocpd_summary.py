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


def gaps(path, anchor="k_dvis_v2", min_us=30.0):
    """Idle time of the GPU between consecutive kernels inside one step = the span between the ends of the last two
    launches of `anchor`: total and the largest gaps with their neighbours."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    ks = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    ends = [i for i, k in enumerate(ks) if anchor in k[0]]
    if len(ends) < 2:
        print("(fewer than two launches of", anchor, ")")
        return
    i0, i1 = ends[-2], ends[-1]
    span = ks[i1][2] - ks[i0][2]
    busy = sum(k[2] - k[1] for k in ks[i0 + 1:i1 + 1])
    print(f"\n## one step ({anchor} end -> {anchor} end): {span/1e6:.2f} ms, kernels {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms, "
          f"{i1 - i0} launches\n")
    print("| idle us | after | before |\n|---|---|---|")
    g = []
    for a, b in zip(ks[i0:i1], ks[i0 + 1:i1 + 1]):
        g.append((b[1] - a[2], a[0].split("(")[0][:60], b[0].split("(")[0][:60]))
    for d, a, b in sorted(g, reverse=True)[:25]:
        if d / 1e3 >= min_us:
            print(f"| {d/1e3:.0f} | `{a}` | `{b}` |")


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
        gaps(sys.argv[1])
