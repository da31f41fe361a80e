"""Development aid (round 5): throughput with several batches of 4096 in flight (MPC_FLAG_THROUGHPUT) against the number of
streams, for a draw with cap-runners (seed 0) and one without (seed 1), and one bulk launch of the same total for comparison."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
B = 4096
for seed in (0, 1):
    inp = synth.solver_inputs(B, 8, seed=seed)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
                weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
                vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
    for cap in (100, 60):
        e = engine.MPCEngine(horizon=20, max_iter=cap, tol=1e-8)
        o1 = e.solve_batch_torch(**args, throughput=True, sync=True)
        t0 = time.perf_counter()
        for _ in range(5):
            e.solve_batch_torch(**args, out=o1, throughput=True)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 5 * 1e3
        line = f"seed {seed} cap {cap}: one batch at a time (throughput build) {one:.2f} ms |"
        for n in (2, 3, 4, 6, 8, 12):
            streams = engine.concurrent_streams(n, dev) if n <= 8 else [torch.cuda.Stream(dev) for _ in range(n)]
            outs = []
            for sq in streams:
                with torch.cuda.stream(sq):
                    outs.append(e.solve_batch_torch(**args, throughput=True))
            torch.cuda.synchronize()
            k = 12 * n
            t0 = time.perf_counter()
            for i in range(k):
                with torch.cuda.stream(streams[i % n]):
                    e.solve_batch_torch(**args, out=outs[i % n], throughput=True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            line += f" {n}: {B * k / el / 1e6:.2f} M/s"
        print(line, flush=True)
        e.close()
