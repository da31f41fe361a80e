"""One-off parity check of the solve kernel against the CPU oracle at the horizons with their own builds (16, 20) and a
runtime one (12), both builds by batch depth (B = 1024: latency build, B = 6000: 128-register build)."""
import sys, os, numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/oracle') else os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'oracle'))
import oracle_lib
from mpc_rl_for_avs_amd import engine, synth
from mpc_rl_for_avs_amd.reference_path import reference_states
ref=reference_states()
for N in (16, 12, 20):
    eng=engine.MPCEngine(horizon=N, max_iter=100)
    for V,cc,B in ((8,True,1024),(4,False,1024),(8,True,6000)):
        inp=synth.solver_inputs(B,V,seed=N+V,N=N)
        got=eng.solve_batch(inp["state"],inp["ego_index"],inp["weights"],inp["is_collide"],vref=inp["vref"],others=inp["others"],collision_cost=cc)
        want=oracle_lib.solve_batch(ref,inp["state"],inp["ego_index"],inp["weights"],inp["is_collide"],vref=inp["vref"],others=inp["others"],collision_cost=cc,max_iter=100,xy_bounds=False,N=N)
        conv=lambda s:(s==0)|(s==5)
        both=conv(got["status"])&conv(want["status"])
        err=np.abs(got["u0"]-want["u0"]).max(axis=1)/np.maximum(1,np.abs(want["u0"]).max(axis=1))
        print(f"N={N} V={V} cc={cc} B={B}: status eq {(got['status']==want['status']).mean():.4f} iters eq {(got['iters']==want['iters'])[both].mean():.4f} beyond 1e-4 {(err[both]>1e-4).sum()} p99 {np.percentile(err[both],99):.1e} conv {both.mean():.4f}", flush=True)
    eng.close()
