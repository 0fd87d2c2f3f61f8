# summary of a CALICO_KERNEL_TIMING=3 dump (stderr of a dev-timing library): when the waves of the last Jacobian launch started and ended
# (100 MHz ticks -> us), by kind. usage: python wave_spans.py < dump
import sys, collections
rows = []
for l in sys.stdin:
    p = l.split()
    if len(p) >= 7 and p[0] == "WAVE" and int(p[4]) > 0:
        rows.append((int(p[1]), p[2], int(p[4]), int(p[6])))
if not rows: sys.exit("no WAVE lines")
med = sorted(r[2] for r in rows)[len(rows) // 2]
rows = [r for r in rows if abs(r[2] - med) < 10000 and 0 < r[3] - r[2] < 10000]      # (one launch: the slots other launches left are dropped)
t0 = min(r[2] for r in rows)
by = collections.defaultdict(list)
for i, k, a, b in rows: by[k].append(((a - t0) / 100.0, (b - t0) / 100.0))
for k, v in by.items():
    st = sorted(x[0] for x in v); en = sorted(x[1] for x in v); du = sorted(x[1] - x[0] for x in v)
    q = lambda s, f: s[min(len(s) - 1, int(f * len(s)))]
    print("%-6s %4d waves | start us: min %.2f median %.2f p90 %.2f max %.2f | duration us: min %.2f median %.2f p90 %.2f max %.2f | end us: median %.2f p90 %.2f max %.2f" % (
        k, len(v), st[0], q(st, .5), q(st, .9), st[-1], du[0], q(du, .5), q(du, .9), du[-1], q(en, .5), q(en, .9), en[-1]))
print("first start to last end: %.2f us" % max(r[3] - t0 for r in rows) * 1 if False else "first start to last end: %.2f us" % (max((r[3] - t0) / 100.0 for r in rows)))
