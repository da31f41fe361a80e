"""Development aid (round 6): a graph of 200 dependent tiny kernels on stream i, alone and while stream j runs one long kernel -
the full matrix over torch's first pool streams (GPU_MAX_HW_QUEUES from Q, default 8).  Dependent dispatches on one hardware queue
slow down while certain other queues are busy (shared dispatch pipe); tools/gpu_pipe_groups.py has what that does to rollout groups."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("Q", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda", 0)
n = int(os.environ.get("NS", "12"))
pool = [torch.cuda.Stream(dev) for _ in range(32)][:n]
x = torch.zeros(1, device=dev)
g = torch.cuda.CUDAGraph()
s_cap = torch.cuda.Stream(dev)
with torch.cuda.stream(s_cap):
    x.add_(1)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s_cap):
    for _ in range(200):
        x.add_(1)
torch.cuda.synchronize()


def chain_time(sa, sb):
    torch.cuda.synchronize()
    if sb is not None:
        with torch.cuda.stream(sb):
            torch.cuda._sleep(int(2.0e9 * 0.02))
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        g.replay()
    sa.synchronize()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t * 1e3


chain_time(pool[0], pool[1])
print("rows: stream with the chain; columns: busy stream; ms (alone on the diagonal)")
print("      " + " ".join(f"{j:5d}" for j in range(n)))
for i in range(n):
    print(f"{i:5d} " + " ".join(f"{chain_time(pool[i], None if j == i else pool[j]):5.2f}" for j in range(n)), flush=True)
