#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_r01_$c -o pmc -- python bench.py --no-cpu-baseline --steps 16 --warmup 2
  done
  python profiles/hbm_traffic_from_pmc.py gpurun_out/pmc_r01_FETCH_SIZE/pmc_counter_collection.csv \
         gpurun_out/pmc_r01_WRITE_SIZE/pmc_counter_collection.csv profiles/r01_pmc_hbm_by_kernel.csv profiles/hbm_traffic.json

Both counters are reported in KiB per dispatch. Corrections (MI355X_MICROARCH.md, "HBM"): on gfx950 FETCH_SIZE tallies
128-B requests at 64 B, i.e. reports half the bytes of coalesced streaming reads, so it is doubled; WRITE_SIZE is taken
as is (uncalibrated). Launches that exit at once (device-side early exit of enqueued-ahead iterations) are dropped:
only dispatches above 1% of the kernel's maximum count.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in agg.items():
        real = [x for x in v if x > 0.01 * max(v)] or v
        out[k] = (sum(real) / len(real) * 1024.0, len(real), len(v))
    return out


def main():
    fetch_csv, write_csv, out_csv, out_json = sys.argv[1:5]
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, (0,))[0] + w.get(k, (0,))[0])):
        fb, fn, fa = f.get(k, (0.0, 0, 0))
        wb, wn, wa = w.get(k, (0.0, 0, 0))
        rows.append((k, fb, 2 * fb, wb, 2 * fb + wb, max(fn, wn), max(fa, wa)))
    with open(out_csv, "w", newline="") as fh:
        cw = csv.writer(fh)
        cw.writerow(["Kernel", "FETCH_SIZE_bytes_raw", "fetch_bytes_x2_gfx950", "WRITE_SIZE_bytes", "hbm_bytes_per_launch",
                     "launches_counted", "launches_total"])
        for r in rows:
            cw.writerow([r[0]] + ["%.0f" % x for x in r[1:5]] + [r[5], r[6]])
    jac = [r for r in rows if "eval_cells_kernel" in r[0] or "eval_jacobian_kernel" in r[0]]
    js = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; FETCH_SIZE doubled (gfx950), see hbm_traffic_from_pmc.py",
          "eval_jacobian_kernel_bytes_per_launch": jac[0][4] if jac else None,
          "eval_jacobian_kernel_fetch_bytes_x2": jac[0][2] if jac else None,
          "eval_jacobian_kernel_write_bytes": jac[0][3] if jac else None,
          # every kernel, by the name in front of its argument list: HBM bytes per launch (fetch x2 + write)
          "by_kernel": {r[0].split("(")[0].replace("void ", ""): r[4] for r in rows}}
    json.dump(js, open(out_json, "w"), indent=1)
    for r in rows[:12]:
        print("%-70s fetch(x2) %10.0f  write %10.0f  total %10.0f B/launch" % (r[0][:70], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
