#!/usr/bin/env python
"""Per-kernel statistics (the `rocprofv3 --kernel-trace --stats` table) from a
rocprofv3 rocpd SQLite database:  python profiles/summarize_rocpd.py <results.db> [out.csv]

Besides the plain average, `AverageWorkingNs`/`WorkingCalls` leave out the dispatches that exit at once (kernels of
iterations enqueued ahead check a device-side flag and return when the step was rejected or the solve has
terminated): dispatches shorter than 10% of the kernel's longest one. That is the figure `bench.py` reports as
`roofline.avg_launch_ms` for the Jacobian kernel."""
import sqlite3
import sys


def summarize(db):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         "max(d.end-d.start), max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), "
         "max(d.group_segment_size), max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
         "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    rows = list(c.execute(q))
    working = {}
    for name, dur in c.execute("select s.kernel_name, d.end-d.start from %s d join %s s on d.kernel_id = s.id" % (kd, ks)):
        working.setdefault(name, []).append(dur)
    tot = sum(r[2] for r in rows) or 1
    out = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,WorkingCalls,AverageWorkingNs,VGPR,AGPR,SGPR,LDS_bytes,Scratch_bytes,Grid,Workgroup"]
    for r in rows:
        d = working[r[0]]
        w = [x for x in d if x >= 0.1 * max(d)]
        out.append('"%s",%d,%d,%.1f,%d,%d,%.2f,%d,%.1f,%d,%d,%d,%d,%d,%d,%d' % (
            r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, len(w), sum(w) / len(w), r[6] or 0, r[7] or 0, r[8] or 0,
            r[9] or 0, r[10] or 0, r[11] or 0, r[12] or 0))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = summarize(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)
