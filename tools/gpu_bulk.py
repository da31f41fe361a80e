"""Development aid: ONE launch of 65536 instances of config 3 (the bulk regime: every SIMD holds its three waves all the
time) a few times - the command of the bulk PMC passes (tools/pmc_run.sh r02_bulk python3 tools/gpu_bulk.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
B = 65536
dev = torch.device('cuda:0')
inp = synth.solver_inputs(B, 8, seed=0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
            weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
            vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
e = engine.MPCEngine(horizon=20, max_iter=int(os.environ.get("MAX_ITER", "60")))
out = e.solve_batch_torch(**args, sync=True)
ts = []
for _ in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); e.solve_batch_torch(**args, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
st = out['status'].cpu().numpy()
print(f"B={B}: {np.median(ts):.2f} ms -> {B / np.median(ts) * 1e3:.0f} solves/s, converged {((st == 0) | ((st >= 5) & (st <= 7))).mean():.4f}")
