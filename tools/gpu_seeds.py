"""Config 3 (B = 4096, cap 100) on the seeds 0 .. 7 = what the ranks of an 8-GPU run of bench.py solve (seed = rank): a step of
that run lasts as long as its slowest rank."""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
e = engine.MPCEngine(horizon=20, max_iter=100)
for seed in range(8):
    inp = synth.solver_inputs(4096, 8, seed=seed)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32), weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8), vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
    out = e.solve_batch_torch(**args, sync=True)
    ts = []
    for _ in range(9):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); e.solve_batch_torch(**args, out=out); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy()
    print(f"seed {seed}: {np.median(ts):.3f} ms, converged {((st==0)|(st==5)).mean():.4f}, at cap {(it>=100).sum()}, iters max {it.max()}", flush=True)
