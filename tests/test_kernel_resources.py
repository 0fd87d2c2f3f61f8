"""The register / scratch budget of the evaluation launch's kernel, from the compiler's own remarks (no GPU needed: hipcc
cross-compiles gfx950). `eval_cells_kernel` takes the whole register file by design (one wave per SIMD); what it must not grow
back is scratch: up to round 6 it carried 152 B per lane -- not spills but two 3x3 matrices indexed at run time --, which was half
of the launch's HBM traffic (profiles/r06_eval_scratch_ab.txt)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


@pytest.mark.skipif(shutil.which(entry.HIPCC) is None and not os.path.exists(entry.HIPCC), reason="no hipcc")
def test_evaluation_kernel_scratch_budget():
    src = os.path.join(entry.CSRC, "eval_kernels.hip")
    flags = [f for f in entry.HIP_FLAGS if f != "-fPIC"] + entry.HIP_FILE_FLAGS.get("eval_kernels.hip", [])
    r = subprocess.run([entry.HIPCC] + flags + ["--cuda-device-only", "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[a-zA-Z/]+\])?: (\d+)", line)
        if m and name:
            res[name][m.group(1).strip()] = int(m.group(2))
    cells = [v for k, v in res.items() if "eval_cells_kernel" in k]
    assert len(cells) == 1, sorted(res)
    c = cells[0]
    assert c["ScratchSize"] <= 64, c          # (28 B today: a handful of spilled registers, no private arrays)
    assert c["VGPRs Spill"] <= 32, c
    assert c["Occupancy"] >= 1 and c["VGPRs"] <= 256 and c["AGPRs"] <= 256, c
    # the launches of the route without cell workgroups: no scratch at all
    for k, v in res.items():
        if "eval_frames_kernel" in k or "eval_jacobian_kernel" in k:
            assert v["ScratchSize"] == 0, (k, v)
