"""Development aid: same-box A/B of mpc_ltv_solve_batch between the built library and the builds of the engine listed in
MPC_AB_LIBS (comma separated), alternating, small batches (the latency regime)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    from mpc_rl_for_avs_amd import engine
    from conftest import ltv_states
    dev = torch.device("cuda:0")
    eng = engine.MPCEngine(horizon=20, max_iter=50)
    for B in (1, 256, 1024, 2048):
        st = torch.as_tensor(ltv_states(B, seed=1), device=dev)
        U = torch.zeros((B, 20, 2), dtype=torch.float64, device=dev)
        out = eng.ltv_solve_batch_torch(st, U, sync=True)
        ts = []
        for _ in range(21):
            U.zero_()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); eng.ltv_solve_batch_torch(st, U, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"   B={B:5d}: median {np.median(ts):.4f} ms  min {np.min(ts):.4f} ms", flush=True)
    sys.exit(0)
for rep in range(2):
    for name, lib in [("built", None)] + [(os.path.basename(l), l) for l in os.environ["MPC_AB_LIBS"].split(",")]:
        env = dict(os.environ)
        if lib: env["MPC_EXPERIMENT_LIB"] = lib
        print(name, flush=True)
        subprocess.run([sys.executable, __file__, "--child"], env=env, check=True)
