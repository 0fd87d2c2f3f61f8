#!/usr/bin/env python
"""What happens between two solves: kernels, DMA copies and idle gaps from the end of one solve's last working kernel to
the next solve's first level kernel.  python profiles/solve_boundary_gaps.py <rocprofv3 rocpd .db>"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = [(s, e, n) for n, s, e in c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (kd, ks))]
    mc = [t for t in tabs if "memory_copy" in t]
    if mc:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % mc[0])]
        if "start" in cols and "end" in cols:
            rows += [(s, e, "<memory copy>") for s, e in c.execute("select start, end from %s" % mc[0])]
    rows.sort()
    # a solve starts with begin_solve_kernel (round 1 / early round 2: init_state_kernel)
    starts = [i for i, r in enumerate(rows) if "begin_solve" in r[2]] or [i for i, r in enumerate(rows) if "init_state" in r[2]]
    if len(starts) < 4:
        raise SystemExit("not enough solves in the trace")
    # all boundaries: idle time from the last working kernel (> 6 us) of a solve to the next solve's begin kernel, and how
    # many early-exit kernels of the old solve start before / after that begin kernel (two alternating streams: after)
    # (the second half of the trace only: the warm-up solves of bench.py bracket every kernel group with HIP events)
    starts = starts[len(starts) // 2:]
    gaps = []
    for i in starts[1:]:
        j = i - 1
        while j > 0 and ((rows[j][1] - rows[j][0]) < 6000 or rows[j][0] > rows[i][0]):
            j -= 1
        before = sum(1 for r in rows[j + 1:i] if "memory copy" not in r[2])
        k = i + 1
        after = 0
        while k < len(rows) and "bcr_level_kernelILb1" not in rows[k][2]:
            after += (rows[k][1] - rows[k][0]) < 6000
            k += 1
        gaps.append(((rows[i][0] - rows[j][1]) / 1e3, before, after))
    gs = sorted(g[0] for g in gaps)
    print("solve boundaries: %d   idle before the begin kernel: min %.1f  median %.1f  p90 %.1f us" % (len(gs), gs[0], gs[len(gs) // 2], gs[int(len(gs) * 0.9)]))
    print("early-exit kernels between the last working kernel and the begin kernel (median): %d; small kernels after it, before the first level kernel (median): %d" % (
        sorted(g[1] for g in gaps)[len(gaps) // 2], sorted(g[2] for g in gaps)[len(gaps) // 2]))
    med = gs[len(gs) // 2]
    i0 = min(zip(starts[1:], gaps), key=lambda t: abs(t[1][0] - med))[0]     # a boundary of median length, in detail
    # walk back to the last kernel of the previous solve that did work (> 6 us)
    j = i0 - 1
    while j > 0 and (rows[j][1] - rows[j][0]) < 6000:
        j -= 1
    t0 = rows[j][1]
    k = i0
    while "bcr_level_kernelILb1" not in rows[k][2] and "band_cholesky" not in rows[k][2] and k < len(rows) - 1:
        k += 1
    prev = t0
    print("from the end of the previous solve's last working kernel to the first linear solve of the next:")
    for s, e, n in rows[j + 1:k + 1]:
        print("  +%8.2f us  gap %7.2f  dur %7.2f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n.split("(")[0][:60]))
        prev = e
    print("total %.2f us" % ((rows[k][0] - t0) / 1e3))


if __name__ == "__main__":
    main()
