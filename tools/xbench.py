"""Development aid: same-box A/B of experimental builds of the engine (MPC_EXPERIMENT_LIB): for each library given on the
command line, in a process of its own - a lone wave's time per iteration (typical instance, two stragglers of seed 0), the
batch of config 3 for a few seeds, config 2.     python tools/xbench.py build/x/lib_a.so build/x/lib_b.so [--seeds 0,4]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from mpc_rl_for_avs_amd import synth, engine
dev = torch.device('cuda:0')
def run(inp, sel, cc, max_iter=100, reps=7, flags_throughput=False):
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a[sel]), dtype=dt, device=dev)
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
    return float(np.median(ts)), st, it
seeds = [int(s) for s in os.environ.get("XB_SEEDS", "0,4").split(",")]
inp = synth.solver_inputs(4096, 8, seed=0)
line = []
for name, i in (("typ", 0), ("s1037", 1037), ("s3424", 3424)):
    ms, st, it = run(inp, np.array([i]), 1, reps=5)
    line.append(f"{name} {ms*1e3/max(it[0],1):.2f} us/it ({it[0]} it)")
for sd in seeds:
    inp_s = synth.solver_inputs(4096, 8, seed=sd)
    ms, st, it = run(inp_s, np.arange(4096), 1)
    conv = (st == 0) | ((st >= 5) & (st <= 7))
    line.append(f"seed{sd} {ms:.3f} ms (conv {conv.mean():.4f} mean {it.mean():.2f} max {it.max()})")
inp2 = synth.solver_inputs(1024, 4, seed=0)
ms, st, it = run(inp2, np.arange(1024), 0)
line.append(f"config2 {ms:.3f} ms (max {it.max()})")
if os.environ.get("XB_BULK"):
    for sd in [int(q) for q in os.environ.get("XB_BULK_SEEDS", "0").split(",")]:
        inpb = synth.solver_inputs(65536, 8, seed=sd)
        ms, st, it = run(inpb, np.arange(65536), 1, reps=3)
        line.append(f"bulk65536 seed {sd} {ms:.2f} ms = {65536/ms/1e3:.3f} M/s")
print(" | ".join(line), flush=True)
''' % ROOT

libs = [a for a in sys.argv[1:] if not a.startswith("--")]
for a in sys.argv[1:]:
    if a.startswith("--seeds"):
        os.environ["XB_SEEDS"] = a.split("=", 1)[1] if "=" in a else "0,4"
    if a == "--bulk":
        os.environ["XB_BULK"] = "1"
rounds = int(os.environ.get("XB_ROUNDS", "2"))
for r in range(rounds):          # alternate the libraries: box drift shows as a difference between the rounds
    for lib in libs:
        env = dict(os.environ)
        extra = ""
        if "@" in lib:              # lib@VAR=value,VAR2=value: environment for this row
            lib, extra = lib.split("@", 1)
            for kv in extra.split(","):
                k, v = kv.split("=", 1)
                env[k] = v
        if lib != "default":
            env["MPC_EXPERIMENT_LIB"] = os.path.join(ROOT, lib)
        res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        out = res.stdout.strip().splitlines()
        print(f"[{os.path.basename(lib)}{('@' + extra) if extra else ''}] " + (out[-1] if out else "FAILED: " + res.stderr[-400:]), flush=True)
