"""Development aid: one batch of BASELINE config 3 (B = 4096) at caps 40 / 60 / 100, several seeds - the figure the choice of
the mid-depth build of the solve kernel is made on (run with MPC_EXPERIMENT_LIB pointing at a candidate build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
for seed in (0, 1, 2):
    inp = synth.solver_inputs(4096, 8, seed=seed)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
                weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
                vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
    row = []
    for cap in (40, 60, 100):
        e = engine.MPCEngine(horizon=20, max_iter=cap)
        out = e.solve_batch_torch(**args, sync=True)
        ts = []
        for _ in range(9):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); e.solve_batch_torch(**args, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        row.append(f"cap {cap}: {np.median(ts):.3f} ms (min {min(ts):.3f})")
        e.close()
    print(f"seed {seed}: " + "; ".join(row), flush=True)
