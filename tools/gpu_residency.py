"""Development aid: what does the second round of waves cost a 4096 batch?  3072 waves are resident at once
(3 per SIMD); the same instances are timed as one batch of 4096, as the first 3072 of them, and with the straggler
instances placed first / last in launch order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
inp = synth.solver_inputs(4096, 8, seed=0)


def run(sel, max_iter=60, reps=9):
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a[sel]), dtype=dt, device=dev)
    args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
                weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
                vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
    e = engine.MPCEngine(horizon=20, max_iter=max_iter)
    out = e.solve_batch_torch(**args, sync=True)
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); e.solve_batch_torch(**args, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    it = out['iters'].cpu().numpy()
    e.close()
    return float(np.median(ts)), it


for cap in (40, 60, 100):
    ms, it = run(np.arange(4096), cap)
    order = np.argsort(-it, kind='stable')
    print(f"cap {cap}: 4096 as given {ms:.3f} ms | first 3072 {run(np.arange(3072), cap)[0]:.3f} ms | "
          f"longest first {run(order, cap)[0]:.3f} ms | longest last {run(order[::-1].copy(), cap)[0]:.3f} ms | "
          f"the 3072 longest alone {run(order[:3072].copy(), cap)[0]:.3f} ms")
