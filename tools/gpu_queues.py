"""Development aid (round 6): what decides whether batches in flight on several streams overlap?  The in-flight measurement of
bench.py / tools/gpu_inflight.py repeated in one process under different ways of obtaining the streams."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
B = 4096
inp = synth.solver_inputs(B, 8, seed=0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
args = dict(state=t(inp['state'], torch.float64), ego_index=t(inp['ego_index'], torch.int32),
            weights=t(inp['weights'], torch.float64), is_collide=t(inp['is_collide'], torch.uint8),
            vref=t(inp['vref'], torch.float64), others=t(inp['others'], torch.float64), collision_cost=True)
e = engine.MPCEngine(horizon=20, max_iter=100, tol=1e-8)


def run(streams, label, k=None):
    n = len(streams)
    outs = []
    for sq in streams:
        with torch.cuda.stream(sq):
            outs.append(e.solve_batch_torch(**args, throughput=True))
    torch.cuda.synchronize()
    k = k or 12 * n
    t0 = time.perf_counter()
    for i in range(k):
        with torch.cuda.stream(streams[i % n]):
            e.solve_batch_torch(**args, out=outs[i % n], throughput=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{label:60s} {n:2d} streams, {k:3d} batches: {B * k / el / 1e6:.2f} M/s", flush=True)


def pair_time(sa, sb):
    outs = []
    for sq in (sa, sb):
        with torch.cuda.stream(sq):
            outs.append(e.solve_batch_torch(**args, throughput=True))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(3):
        for i, sq in enumerate((sa, sb)):
            with torch.cuda.stream(sq):
                e.solve_batch_torch(**args, out=outs[i], throughput=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3


pool = [torch.cuda.Stream(dev) for _ in range(32)]
for rep in range(2):
    run(pool[:8], "pool streams 0-7")
run(pool[8:16], "pool streams 8-15")
run(pool[3:11], "pool streams 3-10")
run(pool[:16], "pool streams 0-15")
run(pool[:8], "pool streams 0-7, 192 batches", k=192)
# which pool streams serialise with pool stream 0?
line = "ms for 2 batches on streams (0, j): "
for j in range(1, 17):
    line += f"{j}:{pair_time(pool[0], pool[j]):.1f} "
print(line, flush=True)
hi = [torch.cuda.Stream(dev, priority=-1) for _ in range(8)]
run(hi, "high-priority pool streams")
e2 = engine.MPCEngine(horizon=20, max_iter=100, tol=1e-8)
e2.solve_batch_torch(**args, sync=True)
e2.close()
run(pool[:8], "pool streams 0-7 after another handle came and went")
run([torch.cuda.Stream(dev) for _ in range(8)], "8 more from torch.cuda.Stream()")
run(engine.concurrent_streams(8, dev), "engine.concurrent_streams(8)")
run(engine.concurrent_streams(8, dev), "engine.concurrent_streams(8) again")
run(engine.concurrent_streams(6, dev), "engine.concurrent_streams(6)")
t0 = time.perf_counter(); engine.concurrent_streams(8, dev); print(f"picking 8 streams takes {(time.perf_counter() - t0) * 1e3:.0f} ms")
