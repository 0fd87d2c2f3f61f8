import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiprocessing as mp
import test_gpu_multirank as T

def run(batched):
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=T._worker, args=(r, 2, 29777 + batched, q, batched)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    [p.join() for p in procs]
    for r in res:
        print("batched", batched, "rank", r[0], "final", r[1], "iters", r[2], "term", r[3])
        for it in r[6][:8]: print("   ", it)

if __name__ == "__main__":
    run(False); run(True)
