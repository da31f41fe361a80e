"""Development aid (round 6): why do two rollout groups lose 20 % in a process with 8 hardware queues?  For pairs of torch's pool
streams: whether they serialise (mpc_streams_overlap), whether a graph of 200 dependent tiny kernels on one slows down while the
other is busy (it does for some pairs: 0.35 -> 0.82 ms), and the 2 x 1024-environment rollout on the pair.  Finding: the rollout
is fast (2.24 M env-steps/s) or slow (1.81 - 1.90 M) by the pair, and neither probe predicts which - which is why rollouts stay with
the runtime's default of 4 hardware queues, where every pair is fast (profiles/r06_hw_queues_scan.txt)."""
import os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("Q", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mpc_rl_for_avs_amd import engine, rollout
dev = torch.device("cuda", 0)
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
            torch.cuda._sleep(int(2.0e9 * 0.03))
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        g.replay()
    sa.synchronize()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t * 1e3


def run(streams, label):
    torch.manual_seed(1234)
    pol = rollout.ActorCritic(1).to(dev)
    engs = [engine.MPCEngine(horizon=20, max_iter=100, device=0) for _ in range(2)]
    cols = [rollout.BatchedCollector(rollout.SyntheticIntersectionEnv(1024, device=dev, seed=gi, n_others=4, env_offset=1024 * gi), pol, engs[gi],
                                     version="v0", algorithm="ppo", n_steps=64, collision_cost=False, seed=gi, throughput=True) for gi in range(2)]
    pipe = rollout.PipelinedCollector(cols)
    if streams is not None:
        pipe.streams = list(streams)
    pipe.collect_rollouts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.collect_rollouts()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:60s} {2048 * 64 / dt / 1e6:.3f} M env-steps/s ({dt / 64 * 1e3:.3f} ms per step)", flush=True)
    for e in engs:
        e.close()


pool = [torch.cuda.Stream(dev) for _ in range(32)]
n = 10
chain_time(pool[0], pool[1])
print("chain matrix (rows: stream with the chain; columns: busy stream; ms, alone on the diagonal)")
mat = [[chain_time(pool[i], None if j == i else pool[j]) for j in range(n)] for i in range(n)]
for i in range(n):
    print(f"{i:3d} " + " ".join(f"{v:5.2f}" for v in mat[i]), flush=True)
for b_ in range(1, n):
    tag = "serialise" if max(mat[0][b_], mat[b_][0]) > 5 else ("chain slowed: " + ("0|b " if mat[0][b_] > 0.6 else "") + ("b|0" if mat[b_][0] > 0.6 else "")) if max(mat[0][b_], mat[b_][0]) > 0.6 else "clean"
    run([pool[0], pool[b_]], f"pool streams 0 and {b_} ({tag})")
for a_, b_ in ((1, 2), (2, 3), (4, 5), (6, 8)):
    tag = "serialise" if max(mat[a_][b_], mat[b_][a_]) > 5 else ("chain slowed" if max(mat[a_][b_], mat[b_][a_]) > 0.6 else "clean")
    run([pool[a_], pool[b_]], f"pool streams {a_} and {b_} ({tag})")
