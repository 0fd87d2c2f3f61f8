#!/usr/bin/env python
"""Per-kernel FP64 / matrix-core utilisation from one rocprofv3 PMC pass (six SQ counters, no trace domains besides
--kernel-trace) and the kernel durations of the kernel-trace pass:

  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
            SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d DIR -o pmc -- python bench.py ...
  python profiles/fp64_from_pmc.py DIR/..._counter_collection.csv profiles/r02_kernel_stats.csv OUT.csv OUT.json

FP64 flops per launch = 64 · (2·FMA_F64 + ADD_F64 + MUL_F64) + 512 · MFMA_MOPS_F64 (wave-level instruction counts times 64
lanes; one MFMA "MOPS" unit = 512 operations -- the gfx94x convention, ROCm 7.2 ships no gfx950 derived-counter
section). Peak: 78.6 TFLOP/s FP64 on MI355X, vector and matrix pipes alike. `mfma_busy_frac` = SQ_VALU_MFMA_BUSY_CYCLES /
SQ_BUSY_CYCLES as counted (both summed over the shader engines). Launches that exit at once are dropped (below 1 % of
the kernel's largest count)."""
import collections
import csv
import json
import sys

PEAK_FP64 = 78.6e12
NAMES = ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_FMA_F64",
         "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"]


def mangled_fragment(demangled):
    """'void cal::bcr_level_kernel<true>(...)' -> 'bcr_level_kernelILb1EE': enough of the Itanium name to find the
    kernel in the kernel-trace statistics (which carry mangled names)."""
    d = demangled.strip().strip('"')
    if d.startswith("void "):
        d = d[5:]
    head = d.split("(")[0]
    base, targs = (head.split("<")[0], head.split("<")[1].rstrip(">")) if "<" in head else (head, "")
    base = base.split("::")[-1]
    frag = base
    if targs:
        frag += "I"
        for a in [t.strip() for t in targs.split(",")]:
            frag += "Lb%dE" % (1 if a == "true" else 0) if a in ("true", "false") else "Li%sE" % a
        frag += "E"
    return frag


def main():
    pmc_csv, stats_csv, out_csv, out_json = sys.argv[1:5]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc_csv)):
        if r["Counter_Name"] in NAMES:
            per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        dur[r["Name"].replace(".kd", "")] = float(r["AverageWorkingNs"])
    rows = []
    for k, c in per.items():
        busy = c.get("SQ_BUSY_CYCLES", [0.0])
        keep = [i for i, v in enumerate(busy) if v > 0.01 * max(busy)] or list(range(len(busy)))

        def avg(name):
            v = c.get(name, [])
            v = [v[i] for i in keep if i < len(v)]
            return sum(v) / len(v) if v else 0.0
        a = {n: avg(n) for n in NAMES}
        flops_valu = 64.0 * (2 * a["SQ_INSTS_VALU_FMA_F64"] + a["SQ_INSTS_VALU_ADD_F64"] + a["SQ_INSTS_VALU_MUL_F64"])
        flops_mfma = 512.0 * a["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        frag = mangled_fragment(k)
        ns = next((v for n, v in dur.items() if frag in n), None)
        frac = ((flops_valu + flops_mfma) / (ns * 1e-9) / PEAK_FP64) if ns else None
        rows.append((k, ns, flops_valu, flops_mfma, frac,
                     a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"] if a["SQ_BUSY_CYCLES"] else None, len(keep)))
    rows.sort(key=lambda r: -(r[1] or 0))
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "avg_working_ns", "fp64_flops_valu_per_launch", "fp64_flops_mfma_per_launch", "fp64_frac_of_78.6TF",
                    "mfma_busy_cycles_over_sq_busy_cycles", "launches_counted"])
        for r in rows:
            w.writerow([r[0], "%.0f" % (r[1] or 0), "%.0f" % r[2], "%.0f" % r[3], "" if r[4] is None else "%.5f" % r[4],
                        "" if r[5] is None else "%.5f" % r[5], r[6]])
    js = {"source": "rocprofv3 --pmc (SQ counters, own pass) + kernel-trace durations; see profiles/fp64_from_pmc.py",
          "peak_fp64_tflops": 78.6,
          "kernels": {r[0].split("(")[0][:60]: {"us": None if r[1] is None else round(r[1] / 1e3, 2),
                                                "fp64_frac": None if r[4] is None else round(r[4], 5),
                                                "mfma_share_of_flops": round(r[3] / (r[2] + r[3]), 3) if r[2] + r[3] > 0 else None,
                                                "mfma_busy_frac": None if r[5] is None else round(r[5], 5)} for r in rows[:12]}}
    json.dump(js, open(out_json, "w"), indent=1)
    for r in rows[:12]:
        print("%-60s %8.2f us  valu %.3e  mfma %.3e flops  fp64 frac %s  mfma busy %s" % (
            r[0][:60], (r[1] or 0) / 1e3, r[2], r[3], "-" if r[4] is None else "%.4f" % r[4], "-" if r[5] is None else "%.4f" % r[5]))


if __name__ == "__main__":
    main()
