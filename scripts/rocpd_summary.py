#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd SQLite output) into a small markdown table.

    python scripts/rocpd_summary.py gpurun_out/prof_x/x_results.db "title" > profiles/rNN_x.md

Kernel-trace runs give the per-kernel call count / total / mean duration
(the `top_kernels` view); --pmc runs additionally give per-kernel counter sums.
"""
import sqlite3
import sys


def _cols(cur, table):
    return [r[1] for r in cur.execute(f"pragma table_info('{table}')").fetchall()]


def _first(cur, prefix):
    for (name,) in cur.execute("select name from sqlite_master where type in ('table','view') order by type desc, name"):
        if name == prefix or name.startswith(prefix):
            return name
    return None


def pmc_rows(cur):
    """per kernel and counter: dispatches, summed value (rocpd: rocpd_pmc_event x rocpd_kernel_dispatch)."""
    try:
        ev, disp = _first(cur, "rocpd_pmc_event"), _first(cur, "rocpd_kernel_dispatch")
        ksym, pinfo = _first(cur, "rocpd_info_kernel_symbol"), _first(cur, "rocpd_info_pmc")
        if not (ev and disp and ksym and pinfo):
            return []
        kname = next(c for c in ("display_name", "kernel_name", "name") if c in _cols(cur, ksym))
        pname = next(c for c in ("name", "symbol", "description") if c in _cols(cur, pinfo))
        q = (f"select k.{kname}, p.{pname}, count(*), sum(e.value) from {ev} e "
             f"join {disp} d on d.event_id = e.event_id join {ksym} k on k.id = d.kernel_id "
             f"join {pinfo} p on p.id = e.pmc_id group by 1, 2 order by 1, 2")
        return cur.execute(q).fetchall()
    except (sqlite3.Error, StopIteration) as e:
        print(f"\n(pmc query failed: {e})")
        return []


def _short(name, width):
    """the kernel's name without its argument list; a long one keeps its FRONT (namespace and kernel name -- what
    scripts/traffic_json.py and bench.py select kernels by; round 4 cut from the left and lost `polyhip::fq::` of the two
    templated feeder kernels) and loses the tail of its template arguments"""
    base = name.split("(")[0]
    return base if len(base) <= width else base[:width - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    cur = db.cursor()
    print(f"# {title}\n")
    print(f"source: `{sys.argv[1]}` (rocprofv3, rocpd format)\n")
    print("| kernel | calls | total ms | mean ms | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = _short(name, 110)
        print(f"| `{short}` | {calls} | {total/1e3:.3f} | {avg/1e3:.4f} | {pct:.2f} |")
    rows = pmc_rows(cur)
    if rows:
        print("\n| kernel | counter | dispatches | sum | per dispatch |")
        print("|---|---|---:|---:|---:|")
        for name, cn, n, v in rows:
            short = _short(name, 110)
            print(f"| `{short}` | {cn} | {n} | {v:.6g} | {v/n:.6g} |")


if __name__ == "__main__":
    main()
