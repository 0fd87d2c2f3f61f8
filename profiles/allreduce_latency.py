"""Latency of the one collective of the path: all-reduce (sum) of the packed normal equations, FP64, in place, over RCCL.
Sizes: configs[3] 0.63 MB, configs[4] 1.9 MB. Run under torch.distributed.run (profiles/collect_scale.sh)."""
import os
import time

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
dist.init_process_group("nccl", rank=rank, world_size=world)
for name, n in (("configs[3]", 82_000), ("configs[4]", 250_000)):
    buf = torch.ones(n, dtype=torch.float64, device="cuda")
    for _ in range(20):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    t = time.perf_counter()
    for _ in range(200):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 200
    if rank == 0:
        print("all-reduce %s: %d doubles (%.2f MB), world %d: %.1f us per call" % (name, n, n * 8 / 1e6, world, dt * 1e6))
dist.destroy_process_group()
