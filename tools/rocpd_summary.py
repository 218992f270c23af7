#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as markdown.
usage: tools/rocpd_summary.py results.db [title]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    print("# %s\n" % title)
    print("| kernel | calls | total (us) | avg (us) | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("| `%s` | %d | %.1f | %.3f | %.2f |" % (name, calls, total, avg, pct))
    rows = cur.execute("select name, min(duration), max(duration), avg(duration), count(*), avg(grid_x), avg(vgpr_count), "
                       "avg(sgpr_count), avg(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    print("\n| kernel | min (ns) | max (ns) | mean (ns) | n | mean grid_x | vgpr | sgpr | lds |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        print("| `%s` | %d | %d | %.0f | %d | %.0f | %.0f | %.0f | %.0f |" % r)
    try:
        rows = cur.execute("select name, sum(value), count(*) from counters_collection group by name").fetchall()
        if rows:
            print("\n| counter | sum over dispatches | dispatches |\n|---|---:|---:|")
            for r in rows:
                print("| %s | %.6g | %d |" % r)
    except Exception:
        pass


if __name__ == "__main__":
    main()
