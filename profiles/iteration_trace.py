#!/usr/bin/env python
"""One steady-state LM iteration as the GPU saw it: every kernel dispatch between two consecutive launches of the
iteration's first kernel, with grid size, duration and the idle gap before it.
  python profiles/iteration_trace.py <rocprofv3 rocpd .db> [first-kernel substring] [which occurrence, default: middle]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "prepare"
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(c.execute("select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s "
                          "on d.kernel_id = s.id order by d.start" % (kd, ks)))
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    if len(idx) < 3:
        raise SystemExit("kernel %r not found often enough" % first)
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
    a, b = idx[k], idx[k + 1]
    prev_end = rows[a - 1][2] if a > 0 else rows[a][1]
    t0 = rows[a][1]
    print("%-44s %8s %6s %9s %8s %9s" % ("kernel", "grid", "wg", "dur_us", "gap_us", "t_us"))
    tot = 0.0
    for name, s, e, g, w in rows[a:b]:
        short = name.split("(")[0].replace("cal::", "")[:44]
        print("%-44s %8d %6d %9.2f %8.2f %9.2f" % (short, g, w, (e - s) / 1e3, (s - prev_end) / 1e3, (s - t0) / 1e3))
        tot += (e - s) / 1e3
        prev_end = e
    print("iteration: %.2f us wall, %.2f us in kernels" % ((rows[b][1] - t0) / 1e3, tot))


if __name__ == "__main__":
    main()
