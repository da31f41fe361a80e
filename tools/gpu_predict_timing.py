#!/usr/bin/env python3
"""Observation-level path at the headline size: B=4096 environments, 9 other vehicles, collision cost on.
Times mpc_predict_batch (device tensors) per call; run under `rocprofv3 --kernel-trace --stats` for the kernel split."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpc_rl_for_avs_amd import engine, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
eng = engine.MPCEngine(horizon=20, max_iter=100)
obs = torch.as_tensor(synth.make_obs_batch(B, 9, seed=0), device=dev)
w = torch.ones((B, 3), dtype=torch.float64, device=dev)
out = None
for cc in (False, True):
    for _ in range(3):
        out = eng.predict_batch_torch(obs, w, collision_cost=cc, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.reset_env_state()                       # same detector work every call
        out = eng.predict_batch_torch(obs, w, collision_cost=cc, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = out["status"].cpu().numpy()
    print(f"predict_batch B={B} V=9 cc={int(cc)}: {dt * 1e3:.2f} ms/call -> {B / dt:.0f} env-steps/s, converged {np.mean((st == 0) | ((st >= 5) & (st <= 7))):.4f}")
# host-pointer call (PCIe-inclusive)
o = obs.cpu().numpy()
wn = np.ones((B, 3))
eng.predict_batch(o, wn, collision_cost=True)
t0 = time.perf_counter()
for _ in range(5):
    eng.reset_env_state()
    eng.predict_batch(o, wn, collision_cost=True)
print(f"host-pointer predict_batch: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/call")
