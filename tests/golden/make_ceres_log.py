#!/usr/bin/env python
"""Extract the Ceres iteration table the reference ships (the only per-iteration output of ceres::Solve it holds):
demos/imu_camera_calibration.ipynb, the output of the cell that runs BatchOptimizer.Optimize with
minimizer_progress_to_stdout. Run in the build container, where /root/reference exists:

    python tests/golden/make_ceres_log.py            # writes tests/golden/ceres_log_imu_camera.npz

The table is DATA printed by the reference's run (iter, cost, cost_change, |gradient|, |step|, tr_ratio, tr_radius); it
pins the trust-region control of ceres::TrustRegionMinimizer / LevenbergMarquardtStrategy -- the radius schedule, the
/2 /4 /8 /16 rejection ladder, the handling of a candidate whose cost cannot be evaluated (cost 1.797693e+308), and
what the cost column shows on a rejected step -- which tests/test_ceres_log.py replays through the oracle and the
device control kernel."""
import json
import os
import sys

import numpy as np

SRC = "/root/reference/demos/imu_camera_calibration.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ceres_log_imu_camera.npz")


def main():
    nb = json.load(open(SRC))
    rows = []
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        # the notebook stores the cell's stdout in several chunks (another stream's output sits between them) and the
        # cut falls in the middle of row 19: the stdout chunks are stitched back together before the rows are parsed
        text = "".join("".join(out.get("text", [])) for out in cell.get("outputs", [])
                       if out.get("output_type") == "stream" and out.get("name") == "stdout")
        if "tr_ratio" not in text:
            continue
        for line in text.split("\n"):
            f = line.split()
            if len(f) >= 7 and f[0].isdigit():
                rows.append([float(v) for v in f[:7]])
    if not rows:
        sys.exit("no iteration table found in " + SRC)
    a = np.array(rows)
    np.savez(OUT, iteration=a[:, 0].astype(np.int32), cost=a[:, 1], cost_change=a[:, 2], gradient_max_norm=a[:, 3],
             step_norm=a[:, 4], tr_ratio=a[:, 5], tr_radius=a[:, 6])
    print("%d rows -> %s" % (len(rows), OUT))


if __name__ == "__main__":
    main()
