import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
mc = [t for t in tabs if "memory_copy" in t]
print("tables", [t for t in tabs if "copy" in t or "memory" in t][:6])
# find last init_state kernel = start of last solve
idx = [i for i, r in enumerate(rows) if "init_state" in r[0]]
i0 = idx[-1]
t0 = rows[i0][1]
prev = None
tot_k = 0
for name, st, en in rows[i0:]:
    short = name.split("(")[0].replace("_ZN3cal", "")[:34]
    gap = (st - prev) / 1e3 if prev else 0
    tot_k += en - st
    if gap > 4 or "init_state" in name or "copyBuffer" in name:
        print("%9.1f us  gap %7.1f  dur %6.1f  %s" % ((st - t0) / 1e3, gap, (en - st) / 1e3, short))
    prev = en
print("last solve: span %.1f us, kernel time %.1f us, kernels %d" % ((rows[-1][2] - t0) / 1e3, tot_k / 1e3, len(rows) - i0))
agg = {}
for name, st, en in rows[i0:]:
    short = name.split("(")[0].replace("_ZN3cal", "")[:30]
    a = agg.setdefault(short, [0, 0.0, 0, 0.0])
    if en - st < 4500 and "init" not in short: a[2] += 1; a[3] += (en - st) / 1e3
    else: a[0] += 1; a[1] += (en - st) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-32s working %3d calls %8.1f us (avg %6.1f) | short(<4.5us) %3d calls %6.1f us" % (k, v[0], v[1], v[1] / max(1, v[0]), v[2], v[3]))
print([round((en - st) / 1e3, 1) for name, st, en in rows[i0:] if "band_cholesky" in name])
print([round((en - st) / 1e3, 1) for name, st, en in rows[i0:] if "prepare" in name])
print([round((en - st) / 1e3, 1) for name, st, en in rows[i0:] if "lm_control" in name])
