#!/bin/bash
# Multi-GPU measurement set, to be run UNATTENDED on a lease with >= 2 MI355X (no such box was available in rounds 1-6: every
# figure this script would produce is unmeasured). One process per GPU, native RCCL exchange inside the library.
#   usage: bash profiles/collect_scale.sh [round-tag]        writes profiles/<tag>_scale_*.{json,txt}
# What it collects:
#   1. pytest -m gpu -k "native_rccl"            -- the 2- / 4-rank native exchange must reach the single-rank solve
#   2. bench.py --gpus {1,2,4,8} at configs[3] and configs[4] (as many as the box has)
#   3. the all-reduce of the packed normal equations alone (0.63 MB at configs[3], 1.9 MB at configs[4]) through RCCL: latency per call
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
N=$(python - <<'PY'
import torch
print(torch.cuda.device_count() if torch.cuda.is_available() else 0)
PY
)
echo "GPUs visible: $N"
if [ "$N" -lt 2 ]; then echo "collect_scale.sh needs at least two GPUs; nothing measured"; exit 2; fi
python -m pytest tests/test_gpu_multirank.py -m gpu -q -k "native_rccl" 2>&1 | tail -5 | tee profiles/${TAG}_scale_pytest.txt
for CFG in 3 4; do
  for G in 1 2 4 8; do
    [ "$G" -le "$N" ] || continue
    timeout 900 python bench.py --gpus $G --config $CFG 2> profiles/${TAG}_scale_cfg${CFG}_g${G}.err | tail -1 > profiles/${TAG}_scale_cfg${CFG}_g${G}.json
    python - <<PY
import json
d = json.load(open("profiles/${TAG}_scale_cfg${CFG}_g${G}.json"))
print("configs[$CFG] gpus $G: %.1f it/s, %.4f ms/step, bound %s" % (d["value"], d["ms_per_step"], d.get("expected_strong_scaling_bound")))
PY
  done
done
# the exchange alone: ncclAllReduce of n doubles, in place, 200 calls behind 20 untimed ones, per world size
for G in 2 4 8; do
  [ "$G" -le "$N" ] || continue
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $G --master-addr 127.0.0.1 --master-port 29517 profiles/allreduce_latency.py \
    | tee profiles/${TAG}_scale_allreduce_g${G}.txt
done
