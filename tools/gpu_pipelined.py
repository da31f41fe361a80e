"""Throughput with several batches in flight (development aid): K solves of the BASELINE config-3 batch issued round-robin
on S HIP streams, so that the straggler tail of one batch (a few lone waves, GPU mostly idle) overlaps with the bulk of
the next.  Prints solves/s for S = 1, 2, 3, 4."""
import os, sys, time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: independent streams get their own hardware queues

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mpc_rl_for_avs_amd import engine, synth

B, V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8
dev = torch.device("cuda:0")
inp = synth.solver_inputs(B, V, seed=0)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
args = dict(state=t(inp["state"], torch.float64), ego_index=t(inp["ego_index"], torch.int32),
            weights=t(inp["weights"], torch.float64), is_collide=t(inp["is_collide"], torch.uint8),
            vref=t(inp["vref"], torch.float64), others=t(inp["others"], torch.float64), collision_cost=True)
eng = engine.MPCEngine(horizon=20, max_iter=100)
ref_out = eng.solve_batch_torch(**args, sync=True)
K = 60
order = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 3, 4, 6]
for S in order:
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    outs = [None] * S
    for s in range(S):
        with torch.cuda.stream(streams[s]):
            outs[s] = eng.solve_batch_torch(**args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        with torch.cuda.stream(streams[k % S]):
            eng.solve_batch_torch(**args, out=outs[k % S])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    same = all(torch.equal(o["u0"], ref_out["u0"]) for o in outs)
    print(f"B={B} streams={S}: {K} solves in {el * 1e3:.2f} ms -> {el / K * 1e3:.3f} ms per batch, {B * K / el:.0f} solves/s; results identical {same}", flush=True)
