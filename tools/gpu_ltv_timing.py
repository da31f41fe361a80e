"""Timing of the iterative-linear MPC path on the GPU (development aid): mpc_ltv_solve_batch on device tensors, first
call (zero profile) and a call linearised about the previous solution, plus the numpy oracle on a sample."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from mpc_rl_for_avs_amd import engine, reference_path as R
from conftest import ltv_states
import ltv_oracle as L

dev = torch.device("cuda:0")
ref = R.reference_states()
eng = engine.MPCEngine(horizon=20, max_iter=50)
for B in (1, 256, 1024, 4096, 16384, 65536):
    st = ltv_states(B, seed=1)
    t_state = torch.as_tensor(st, device=dev)
    t_U = torch.zeros((B, 20, 2), dtype=torch.float64, device=dev)
    out = eng.ltv_solve_batch_torch(t_state, t_U, sync=True)
    warmU = t_U.clone()
    for name, U0 in (("first call", torch.zeros_like(t_U)), ("about previous", warmU)):
        ts = []
        for _ in range(7):
            U = U0.clone()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); eng.ltv_solve_batch_torch(t_state, U, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        it = out["iters"].cpu().numpy(); stt = out["status"].cpu().numpy()
        print(f"B={B:6d} {name:15s}: {ms:8.3f} ms -> {B / ms * 1e3:10.0f} solves/s; iters mean {it[stt == 0].mean():.2f} "
              f"max {it.max()} status {np.bincount(stt, minlength=4)}", flush=True)
st = ltv_states(512, seed=1)
t0 = time.time(); o = L.solve_batch(ref, st, np.zeros((512, 20, 2))); t1 = time.time()
print(f"numpy oracle (dense condensed QP, batched LAPACK): 512 instances in {t1 - t0:.2f} s -> {512 / (t1 - t0):.0f} solves/s")
g = eng.ltv_solve_batch(st, np.zeros((512, 20, 2)))
ok = (g["status"] == 0) & (o["status"] == 0)
print(f"parity on them: status equal {np.array_equal(g['status'], o['status'])}, u0 max abs diff {np.abs(g['u0'] - o['u0'])[ok].max():.2e}")
