import sys, re, collections
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"WAVE (\d+) (\w+) t0 (\d+) t1 (\d+) hw ([0-9a-f]+) xcc ([0-9a-f]+)", line)
    if m: rows.append((int(m[1]), m[2], int(m[3]), int(m[4]), int(m[5], 16), int(m[6], 16)))
# keep the last launch: split on blockIdx 0 reappearing
launches = []; cur = {}
for r in rows:
    if r[0] in cur: launches.append(cur); cur = {}
    cur[r[0]] = r
launches.append(cur)
L = launches[int(sys.argv[2])] if len(sys.argv) > 2 else launches[0]
t00 = min(r[2] for r in L.values())
n_items = max(i for i, r in L.items() if r[1] == "item") + 1
print("launch with", len(L), "waves;", n_items, "items; span %.2f us" % ((max(r[3] for r in L.values()) - t00) / 100.0))
def stats(sel, name):
    d = [(r[3] - r[2]) / 100.0 for r in sel]; s = [(r[2] - t00) / 100.0 for r in sel]; e = [(r[3] - t00) / 100.0 for r in sel]
    print("%-8s n %4d  dur us min %.1f avg %.1f max %.1f | start max %.1f | end max %.1f" % (name, len(d), min(d), sum(d)/len(d), max(d), max(s), max(e)))
stats([r for r in L.values() if r[1] == "item"], "items")
stats([r for r in L.values() if r[1] == "frame"], "frames")
# co-residency per (xcc, se, cu, simd)
slot = collections.Counter()
for r in L.values():
    hw, xcc = r[4], r[5] & 0xf
    slot[(xcc, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3)] += 1
print("distinct SIMDs used", len(slot), "max waves on one SIMD", max(slot.values()), "histogram", collections.Counter(slot.values()))
cus = collections.Counter((k[0], k[1], k[2]) for k in slot.elements())
print("distinct CUs used", len(cus), "max waves per CU", max(cus.values()), "hist", sorted(collections.Counter(cus.values()).items()))
late = sorted(L.values(), key=lambda r: -r[3])[:12]
for r in late: print("  late:", r[0], r[1], "start %.1f end %.1f" % ((r[2]-t00)/100.0, (r[3]-t00)/100.0), "simd-mates", slot[(r[5]&0xf, (r[4]>>13)&7, (r[4]>>8)&15, (r[4]>>4)&3)])
import collections as _c
h = _c.Counter(round((r[2] - t00) / 100.0) for r in L.values())
print("start-time histogram (us: waves):", sorted(h.items()))
hi = _c.Counter(round((r[2] - t00) / 100.0) for r in L.values() if r[1] == "item")
print("items only:", sorted(hi.items()))
late_xcc = _c.Counter((r[5] & 0xf) for r in L.values() if (r[2] - t00) / 100.0 > 4)
print("late starters per XCC:", sorted(late_xcc.items()))
