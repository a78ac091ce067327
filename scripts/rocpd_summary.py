#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd SQLite output) into a small markdown table.

    python scripts/rocpd_summary.py gpurun_out/prof_x/x_results.db "title" > profiles/rNN_x.md

Kernel-trace runs give the per-kernel call count / total / mean duration
(the `top_kernels` view); --pmc runs additionally give per-kernel counter sums.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    cur = db.cursor()
    print(f"# {title}\n")
    print(f"source: `{sys.argv[1]}` (rocprofv3, rocpd format)\n")
    print("| kernel | calls | total ms | mean ms | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0][-70:]
        print(f"| `{short}` | {calls} | {total/1e3:.3f} | {avg/1e3:.4f} | {pct:.2f} |")
    try:
        rows = cur.execute(
            "select k.name, p.counter_name, count(*), sum(p.value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error as e:
        rows = []
    if rows:
        print("\n| kernel | counter | dispatches | sum | per dispatch |")
        print("|---|---|---:|---:|---:|")
        for name, cn, n, v in rows:
            short = name.split("(")[0][-50:]
            print(f"| `{short}` | {cn} | {n} | {v:.6g} | {v/n:.6g} |")


if __name__ == "__main__":
    main()
