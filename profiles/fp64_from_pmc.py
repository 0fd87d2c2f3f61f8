#!/usr/bin/env python
"""Per-kernel FP64 / matrix-core utilisation from one rocprofv3 PMC pass (six SQ counters, no trace domains besides
--kernel-trace) and the kernel durations of the kernel-trace pass:

  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
            SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d DIR -o pmc -- python bench.py ...
  python profiles/fp64_from_pmc.py DIR/..._counter_collection.csv profiles/r02_kernel_stats.csv OUT.csv OUT.json

FP64 flops per launch = 64 · (2·FMA_F64 + ADD_F64 + MUL_F64) + 512 · MFMA_MOPS_F64 (wave-level instruction counts times 64
lanes; one MFMA "MOPS" unit = 512 operations -- the gfx94x convention, ROCm 7.2 ships no gfx950 derived-counter
section). Peak: 78.6 TFLOP/s FP64 on MI355X, vector and matrix pipes alike.

MFMA utilisation (what fraction of the chip's matrix-pipe cycles a kernel keeps busy), two ways that must agree:
  `mfma_util` = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs). The counter is the sum over all SIMDs
  of the cycles their matrix pipe is held: measured here it is EXACTLY 64 x the number of v_mfma_f64_16x16x4_f64 issued
  (MOPS / 4) for every kernel of the run. The duration is the un-profiled one of the kernel-trace pass.
  `mfma_util_from_flops` = MFMA flops / duration / 78.6 TFLOP/s (2048 flops per instruction, 64 cycles of one SIMD's pipe:
  32 flops per cycle and SIMD = 78.6 TFLOP/s over 1024 SIMDs at 2.4 GHz).
Both are <= 1 by construction. GRBM_GUI_ACTIVE (optional 5th argument: a pass of its own) is NOT used as the denominator: as
rocprofv3 reports it per dispatch it is summed over the eight XCDs and spans the profiler's counter window (a one-thread
kernel reads 177k = 9 us x 8 x 2.4 GHz); it is kept in the CSV (`grbm_gui_active_raw`) for reference. (`mfma_busy_cycles_over_sq_busy_cycles`, the figure round 2 printed, divides a per-SIMD sum
by a per-shader-engine sum and is not a utilisation; kept in the CSV for continuity only.)
Launches that exit at once are dropped (below 1 % of the kernel's largest count)."""
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


N_SIMD = 1024
CLOCK_HZ = 2.4e9


def main():
    pmc_csv, stats_csv, out_csv, out_json = sys.argv[1:5]
    gui = collections.defaultdict(list)
    if len(sys.argv) > 5:
        for r in csv.DictReader(open(sys.argv[5])):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui[r["Kernel_Name"]].append(float(r["Counter_Value"]))
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
        g = gui.get(k, [])
        g = [v for v in g if v > 0.1 * max(g)] if g else []
        gui_raw = (sum(g) / len(g)) if g else None
        active = (ns * 1e-9 * CLOCK_HZ) if ns else None
        util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * N_SIMD) if active else None
        util_flops = (flops_mfma / (ns * 1e-9) / PEAK_FP64) if ns else None
        rows.append((k, ns, flops_valu, flops_mfma, frac,
                     a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"] if a["SQ_BUSY_CYCLES"] else None, len(keep), util, util_flops,
                     active, gui_raw))
    rows.sort(key=lambda r: -(r[1] or 0))
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "avg_working_ns", "fp64_flops_valu_per_launch", "fp64_flops_mfma_per_launch", "fp64_frac_of_78.6TF",
                    "mfma_busy_cycles_over_sq_busy_cycles", "launches_counted", "mfma_util", "mfma_util_from_flops",
                    "duration_x_2.4GHz_cycles", "grbm_gui_active_raw"])
        for r in rows:
            w.writerow([r[0], "%.0f" % (r[1] or 0), "%.0f" % r[2], "%.0f" % r[3], "" if r[4] is None else "%.5f" % r[4],
                        "" if r[5] is None else "%.5f" % r[5], r[6], "" if r[7] is None else "%.5f" % r[7],
                        "" if r[8] is None else "%.5f" % r[8], "" if r[9] is None else "%.0f" % r[9], "" if r[10] is None else "%.0f" % r[10]])
    js = {"source": "rocprofv3 --pmc (SQ counters, own pass) + kernel-trace durations; see profiles/fp64_from_pmc.py",
          "peak_fp64_tflops": 78.6,
          "kernels": {r[0].split("(")[0][:60]: {"us": None if r[1] is None else round(r[1] / 1e3, 2),
                                                "fp64_frac": None if r[4] is None else round(r[4], 5),
                                                "mfma_share_of_flops": round(r[3] / (r[2] + r[3]), 3) if r[2] + r[3] > 0 else None,
                                                "mfma_util": None if r[7] is None else round(r[7], 5),
                                                "mfma_util_from_flops": None if r[8] is None else round(r[8], 5)} for r in rows[:12]}}
    dom = [r for r in rows if "eval_cells_kernel" in r[0] or "eval_jacobian_kernel" in r[0]]
    if dom:
        js["mfma_utilisation_dominant_kernel"] = {
            "kernel": "eval_cells_kernel / eval_jacobian_kernel", "mfma_util": None if dom[0][7] is None else round(dom[0][7], 5),
            "mfma_util_from_flops": None if dom[0][8] is None else round(dom[0][8], 5),
            "definition": "SQ_VALU_MFMA_BUSY_CYCLES (sum over SIMDs) / (kernel duration x 2.4 GHz x 1024 SIMDs); cross-check: MFMA flops / duration / 78.6 TFLOP/s"}
    bad = [r[0] for r in rows if (r[7] is not None and r[7] > 1.0) or (r[8] is not None and r[8] > 1.0)]
    if bad:
        raise SystemExit("utilisation above 1 for %s: the counters are being read wrongly" % bad)
    json.dump(js, open(out_json, "w"), indent=1)
    for r in rows[:12]:
        print("%-60s %8.2f us  valu %.3e  mfma %.3e flops  fp64 frac %s  mfma util %s (from flops %s)" % (
            r[0][:60], (r[1] or 0) / 1e3, r[2], r[3], "-" if r[4] is None else "%.4f" % r[4], "-" if r[7] is None else "%.4f" % r[7],
            "-" if r[8] is None else "%.4f" % r[8]))


if __name__ == "__main__":
    main()
