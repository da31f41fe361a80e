import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from mpc_rl_for_avs_amd import rollout, engine
dev = torch.device("cuda:0")
B = 256
env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=0, n_others=4)
eng = engine.MPCEngine(horizon=20, max_iter=100)
eng.set_diagnostics(True)
obs = env.reset()
w = torch.ones((B, 3), dtype=torch.float64, device=dev)
acc = {0: [], 1: []}
names_full = ["parse", "nearest", "ramp", "seg+cumsum", "points", "veh paths", "hit tests", "candidates", "cand loop", "export+finish"]
names_replay = ["parse", "nearest", "(skip)", "export+finish"]
for step in range(48):
    out = eng.predict_batch_torch(obs, w, None, sync=True)
    lp = eng.last_paths(B, 10)
    t = lp["agent_paths"][:, -1].reshape(B, -1)
    n = t[:, 0].astype(int); rep = t[:, 1] > 0.5
    for b in range(B):
        acc[int(rep[b])].append(t[b, 2:2 + n[b]])
    obs, rew, done, info = env.step(out["act"])
for r in (0, 1):
    rows = acc[r]
    if not rows: continue
    L = max(len(x) for x in rows)
    m = np.array([np.pad(x, (0, L - len(x))) for x in rows if len(x) == L])
    print(("REPLAY" if r else "FULL DETECTION"), "n", len(rows), "of which complete", len(m), "cycles mean per section:")
    nm = names_replay if r else names_full
    for i in range(L):
        print(f"   {(nm[i] if i < len(nm) else str(i)):14s} mean {m[:, i].mean():9.0f}  p90 {np.percentile(m[:, i], 90):9.0f}  max {m[:, i].max():9.0f}")
    print("   total mean", m.sum(axis=1).mean(), "max", m.sum(axis=1).max(), " (2.4 GHz: us)", m.sum(axis=1).mean() / 2400)
