"""What the iteration cap hides: config 3 (B = 4096, horizon 20, 8 vehicles, collision cost on) at the engine's default
cap, at the reference's own solver settings (ipopt max_iter 1000, tol 1e-6: agents/pure_mpc.py:294-295) and in between.
Per setting: batch time (median of event-timed repetitions), converged fraction, iteration distribution, and what the
instances that a cap of 60 / 100 cuts off do when they are allowed to go on.  Output: profiles/rNN_tail.txt (stdout) and
gpurun_out/tail_iters.npz (per-instance iteration counts and statuses)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
B, V = 4096, 8


def run(seed, max_iter, tol, reps=5):
    inp = synth.solver_inputs(B, V, seed=seed)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
                weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
                vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
    e = engine.MPCEngine(horizon=20, max_iter=max_iter, tol=tol)
    out = e.solve_batch_torch(**args, sync=True)
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); e.solve_batch_torch(**args, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    st = out['status'].cpu().numpy(); it = out['iters'].cpu().numpy()
    e.close()
    return float(np.median(ts)), st, it


keep = {}
print("seed max_iter tol      ms     solves/s  converged  st0    st5    st1  st2  st4  st3 | iters mean p50 p90 p99 p99.9 max")
for seed in (0, 1, 2):
    for max_iter, tol in ((60, 1e-8), (100, 1e-8), (200, 1e-8), (1000, 1e-8), (100, 1e-6), (1000, 1e-6)):
        ms, st, it = run(seed, max_iter, tol, reps=5 if max_iter <= 200 else 3)
        conv = (st == 0) | ((st >= 5) & (st <= 7))
        c = lambda s: int((st == s).sum())
        pc = np.percentile(it, [50, 90, 99, 99.9])
        print(f"{seed:4d} {max_iter:8d} {tol:.0e} {ms:7.3f} {B / ms * 1e3:10.0f}  {conv.mean():.5f}  {c(0):5d} {c(5):5d} {c(1):5d} "
              f"{c(2):4d} {c(4):4d} {c(3):4d} | {it.mean():6.2f} {pc[0]:4.0f} {pc[1]:4.0f} {pc[2]:4.0f} {pc[3]:5.0f} {it.max():5d}", flush=True)
        keep[f"s{seed}_m{max_iter}_t{tol:.0e}_status"] = st
        keep[f"s{seed}_m{max_iter}_t{tol:.0e}_iters"] = it
    # the instances a cap cuts off: what becomes of them with 1000 iterations
    for cap in (60, 100):
        st_c = keep[f"s{seed}_m{cap}_t1e-08_status"]
        cut = st_c == 1
        st_f, it_f = keep[f"s{seed}_m1000_t1e-08_status"][cut], keep[f"s{seed}_m1000_t1e-08_iters"][cut]
        print(f"     seed {seed}: {int(cut.sum())} instances at cap {cap} -> with cap 1000: "
              f"{int(((st_f == 0) | ((st_f >= 5) & (st_f <= 7))).sum())} converge (iterations {sorted(it_f[(st_f == 0) | ((st_f >= 5) & (st_f <= 7))].tolist())}), "
              f"{int((st_f == 4).sum())} stall (status 4), {int((st_f == 1).sum())} still running at 1000, {int((st_f == 2).sum())} status 2")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "tail_iters.npz"), **keep)
