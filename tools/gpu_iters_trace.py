"""Development aid: iteration counts per environment and step of a closed-loop rollout (8192 envs), to see how well the
previous step's count predicts the next one (launch-order hint)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import engine, rollout
dev = torch.device("cuda:0")
B, T = 8192, 40
pol = rollout.ActorCritic(1).to(dev)
e = engine.MPCEngine(horizon=20, max_iter=100)
env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=0, n_others=4)
col = rollout.BatchedCollector(env, pol, e, version="v0", algorithm="ppo", n_steps=T, collision_cost=False, seed=0)
its = []
inner = e.predict_batch_torch
def wrapped(*a, **k):
    out = inner(*a, **k)
    its.append(out["iters"].clone())
    return out
e.predict_batch_torch = wrapped
col.collect_rollouts()
torch.cuda.synchronize()
its = torch.stack(its).cpu().numpy()
np.save(os.path.join(ROOT, "gpurun_out", "r02_iters_trace.npy"), its)
print(its.shape, "mean", its.mean(), "max", its.max())
for t in (1, 5, 10, 20, 30):
    print(t, "corr with previous step %.3f" % np.corrcoef(its[t - 1], its[t])[0, 1], "p99", np.percentile(its[t], 99), "max", its[t].max())
