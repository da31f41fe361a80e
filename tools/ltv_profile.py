"""rocprofv3 target for the iterative-linear path: a few launches of mpc_ltv_kernel at B = 4096 (first call of an
episode and one re-linearised call), nothing else on the GPU.  `rocprofv3 --kernel-trace --stats -- python3 tools/ltv_profile.py`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from mpc_rl_for_avs_amd import engine
from conftest import ltv_states

dev = torch.device("cuda:0")
eng = engine.MPCEngine(horizon=20, max_iter=50)
B = 4096
st = torch.as_tensor(ltv_states(B, seed=1), device=dev)
U = torch.zeros((B, 20, 2), dtype=torch.float64, device=dev)
out = eng.ltv_solve_batch_torch(st, U, sync=True)
for _ in range(10):
    U.zero_()
    eng.ltv_solve_batch_torch(st, U, out=out)          # first call of an episode
    eng.ltv_solve_batch_torch(st, U, out=out)          # linearised about that solution
torch.cuda.synchronize()
s = out["status"].cpu().numpy()
print("status", np.bincount(s, minlength=4), "iters mean", out["iters"].cpu().numpy()[s == 0].mean())
