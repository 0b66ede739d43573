#!/usr/bin/env python
"""Summarise a rocprofv3 result database (rocpd SQLite): per-kernel time table and, when the run
collected counters, per-kernel counter sums.  Usage: prof_summary.py <results.db> [...]"""
import sqlite3
import sys


def main(paths):
    for path in paths:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                                "from kernels group by name order by sum(end-start) desc"))
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':44s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>9s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}")
        for name, calls, total, avg, mn, mx in rows:
            print(f"{name.split('(')[0][:44]:44s} {calls:6d} {total/1e6:10.3f} {avg/1e6:9.4f} {mn/1e6:9.4f} {mx/1e6:9.4f} "
                  f"{100.0*total/tot:6.2f}")
        try:
            rows = list(cur.execute(
                "select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events "
                "group by name, counter_name order by name, counter_name"))
        except sqlite3.Error:
            rows = []
        if rows:
            print(f"{'kernel':32s} {'counter':26s} {'sum over dispatches':>22s} {'dispatches':>10s} {'per dispatch':>18s}")
            for name, c, v, n in rows:
                if name.startswith("k_") or name.startswith("void k_"):
                    print(f"{name.split('(')[0][:32]:32s} {c:26s} {v:22.0f} {n:10d} {v/n:18.0f}")
        regs = list(cur.execute("select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                                "grid_x, workgroup_x from kernels where name like 'k_%' or name like 'void k_%'"))
        print(f"{'kernel':32s} vgpr agpr sgpr lds scratch grid wg")
        for r in regs:
            print(f"{r[0].split('(')[0][:32]:32s} " + " ".join(str(x) for x in r[1:]))


if __name__ == "__main__":
    main(sys.argv[1:])
