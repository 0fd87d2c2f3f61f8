#!/usr/bin/env python
"""Per-iteration timeline from a rocprofv3 rocpd database: for every kernel, the average duration and the average idle
gap between its start and the end of the kernel dispatched before it (steady-state iterations only: the gap is
counted when it is below 50 us).  python profiles/timeline_gaps.py <results.db>"""
import sqlite3
import sys


def main(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
    stat = {}
    prev_end = None
    for name, st, en in rows:
        short = name.split("(")[0].replace("_ZN3cal", "").split("ENS_")[0]
        if prev_end is not None and 0 <= st - prev_end < 50000 and en - st > 3000:
            a = stat.setdefault(short, [0, 0, 0])
            a[0] += 1; a[1] += en - st; a[2] += st - prev_end
        prev_end = en
    tot_d = tot_g = 0.0
    print("%-40s %8s %10s %10s" % ("kernel", "calls", "avg_us", "gap_before_us"))
    for k, (n, d, g) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        print("%-40s %8d %10.2f %10.2f" % (k[:40], n, d / n / 1e3, g / n / 1e3))
        tot_d += d; tot_g += g
    print("busy fraction in steady state: %.3f" % (tot_d / (tot_d + tot_g)))


if __name__ == "__main__":
    main(sys.argv[1])
