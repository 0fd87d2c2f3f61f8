"""Fills the @@...@@ placeholders of DESIGN.md / README.md / profiles/README.md from the round's committed measurement files
(profiles/rNN_*): run once after `collect_rNN.sh`'s summaries have been copied into profiles/."""
import csv, json, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = "profiles/%s_" % tag
b = json.load(open(P + "bench.json"))
def val(f): return json.load(open(P + f))["value"]
ks = {}
for r in csv.DictReader(open(P + "kernel_stats.csv")):
    ks[r["Name"]] = float(r["AverageWorkingNs"]) / 1e3
def k(sub, excl=None):
    return [v for n, v in ks.items() if sub in n and not (excl and excl in n)][0]
l0, l1 = k("bcr_level_kernelILb1"), k("bcr_level_kernelILb0")
db, ev, ga = k("dense_back_kernel"), k("eval_cells_kernel"), k("gather_kernelI")
ab = [l.split() for l in open(P + "round_ab.txt") if l.strip()]
new = [float(x[1]) for x in ab if x[0] == "base"]; old = [float(x[1]) for x in ab if x[0] != "base"]
mnew, mold = sum(new) / len(new), sum(old) / len(old)
rep = {
    "BENCH": "{:,.0f}".format(b["value"]), "BENCHR": "{:,.0f}".format(round(b["value"], -1)), "MS": "%.4f" % b["ms_per_step"], "FRAC": "%.4f" % b["roofline"]["frac"],
    "BASE": "{:,.0f}".format(mold), "BENCHAB": "{:,.0f}".format(mnew), "GAIN": "+%.1f %%" % (100 * (mnew / mold - 1)),
    "L0": "%.1f" % l0, "L1": "%.1f" % l1, "DB": "%.1f" % db, "EV": "%.1f" % ev, "GA": "%.1f" % ga, "KSUM": "%.1f" % (l0 + l1 + db + ev + ga), "LS": "%.1f" % (l0 + l1 + db),
    "C1": "{:,.0f}".format(val("config1.json")), "C2": "{:,.0f}".format(val("config2.json")), "C4": "{:,.0f}".format(val("config4.json")),
    "C5": "{:,.0f}".format(val("shape_config3_50hz_knots.json")), "C6": "{:,.0f}".format(val("shape_notebook_run.json")),
}
for f in ("DESIGN.md", "README.md", "profiles/README.md"):
    s = open(f).read()
    s2 = re.sub(r"@@([A-Z0-9]+)@@", lambda m: rep[m.group(1)], s)
    if s2 != s:
        open(f, "w").write(s2)
        print(f, "filled")
print(rep)
