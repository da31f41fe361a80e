"""Closed-loop check of the iterative-linear path (development aid): E synthetic environments stepped with
mpc_ltv_predict_batch on device tensors; prints status histogram, iteration statistics and step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mpc_rl_for_avs_amd import engine, rollout

E, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda:0")
env = rollout.SyntheticIntersectionEnv(E, device=dev, seed=0, n_others=3)
eng = engine.MPCEngine(horizon=20, max_iter=50)
obs = env.reset()
hist = np.zeros(5, dtype=np.int64)
its, ms = [], []
out = None
dump = os.environ.get("LTV_DUMP")          # path: save (state, stored profile) of unsolved instances for the oracle
U = torch.zeros((E, 20, 2), dtype=torch.float64, device=dev)
fail_state, fail_U = [], []
for t in range(steps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    if dump:
        o = obs.contiguous()
        yaw = torch.remainder(o[:, 0, 5].double() + np.pi, 2 * np.pi) - np.pi
        state = torch.stack([o[:, 0, 1].double(), o[:, 0, 2].double(), torch.hypot(o[:, 0, 3], o[:, 0, 4]).double(), yaw], 1).contiguous()
        U_before = U.clone()
        e0.record()
        out = eng.ltv_solve_batch_torch(state, U, out=out)
        e1.record()
        out["act"] = out["u0"]
    else:
        e0.record()
        out = eng.ltv_predict_batch_torch(obs.contiguous(), out=out)
        e1.record()
    obs, reward, done, info = env.step(out["act"])
    if dump:
        bad = out["status"] == 1
        if bool(bad.any()):
            fail_state.append(state[bad].cpu().numpy()); fail_U.append(U_before[bad].cpu().numpy())
        U[done.bool()] = 0.0
    else:
        eng.reset_env_mask_torch(done.to(torch.uint8).contiguous())
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
    st = out["status"].cpu().numpy()
    hist += np.bincount(st, minlength=5)
    its.append(out["iters"].cpu().numpy()[st == 0])
its = np.concatenate(its)
if dump and fail_state:
    np.savez(dump, state=np.concatenate(fail_state), U=np.concatenate(fail_U))
print(f"E={E} steps={steps}: status histogram {hist.tolist()} (0 solved, 1 max_iter, 2 factorization, 3 speed outside its bounds)")
print(f"iterations mean {its.mean():.2f} p99 {np.percentile(its, 99):.0f} max {its.max()}; "
      f"predict step median {np.median(ms[2:]):.3f} ms -> {E / np.median(ms[2:]) * 1e3:.0f} env-steps/s")
