import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("Q", "8")
sys.path.insert(0, "/root/repo")
import torch
from mpc_rl_for_avs_amd import engine
dev = torch.device("cuda", 0)
ch = engine.concurrent_streams(int(os.environ.get("NS", "8")), dev)
x = torch.zeros(1, device=dev)
g = torch.cuda.CUDAGraph()
# a captured chain of 200 dependent tiny kernels (like a rollout's step chain, no host in the loop)
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
            torch.cuda._sleep(int(2.0e9 * 0.03))
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        g.replay()
    sa.synchronize()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t * 1e3

for rep in range(2):
    base = chain_time(ch[0], None)
    line = f"chain of 200 dependent kernels on stream 0 alone: {base:.2f} ms; with stream j busy: "
    for j in range(1, len(ch)):
        line += f"{j}:{chain_time(ch[0], ch[j]):.2f} "
    print(line, flush=True)
