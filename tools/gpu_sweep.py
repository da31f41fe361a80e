"""Development aid: solve time of the bench batch vs iteration cap and batch size (MI355X)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
def run(B, V, cc, max_iter, reps=5):
    inp = synth.solver_inputs(B, V, seed=0)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
                weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
                vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=bool(cc))
    e = engine.MPCEngine(horizon=20, max_iter=max_iter)
    out = e.solve_batch_torch(**args, sync=True)
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); e.solve_batch_torch(**args, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy()
    e.close()
    return np.median(ts), (st == 0).mean(), it.mean()
for mi in (40, 60, 80, 100, 200):
    ms, conv, itm = run(4096, 8, 1, mi)
    print(f"config3 B=4096 V=8 cc=1 max_iter={mi}: {ms:.2f} ms -> {4096/ms*1e3:.0f} solves/s, converged {conv:.4f}, mean iters {itm:.1f}", flush=True)
for (B, V, cc) in ((1, 8, 1), (256, 8, 1), (1024, 4, 0), (1024, 8, 1), (4096, 4, 0), (16384, 8, 1), (65536, 8, 1)):
    ms, conv, itm = run(B, V, cc, 100, reps=3)
    print(f"B={B} V={V} cc={cc} max_iter=100: {ms:.2f} ms -> {B/ms*1e3:.0f} solves/s, converged {conv:.4f}, mean iters {itm:.1f}", flush=True)
