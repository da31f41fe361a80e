#!/usr/bin/env python3
"""MPC-in-the-loop rollout throughput (BASELINE configs 4-5): B synthetic intersection environments, torch policy,
device preamble + solve (`mpc_predict_batch`), vectorised env step, everything resident on the GPU.
Prints one JSON line per configuration: env-steps/s (= MPC solves/s) and the time split.

  python tools/bench_rollout.py --envs 256 2048                    # one GPU (config 4 and config 5's total on one GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_rollout.py \
         --envs 2048                                               # config 5: 256 envs per GPU, RCCL gather of actions
"""
import argparse
import json
import os

import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[256, 2048])
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--others", type=int, default=4)
    ap.add_argument("--version", default="v0")
    ap.add_argument("--algorithm", default="ppo", choices=("ppo", "a2c"))
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="run the step eagerly (launches from Python, MPC time measured with events) instead of replaying the "
                         "captured hipGraph, which is the collector's default on a GPU")
    ap.add_argument("--env-backend", default="auto", choices=("auto", "hip", "torch"),
                    help="fused HIP environment step (default on a GPU) or the vectorised torch ops")
    ap.add_argument("--max-iter", type=int, default=100, help="solver iteration cap (the reference: 1000)")
    ap.add_argument("--tol", type=float, default=1e-8, help="solver tolerance (the reference: 1e-6)")
    ap.add_argument("--groups", type=int, default=1,
                    help="split this rank's environments into G groups stepped on G HIP streams (PipelinedCollector)")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from mpc_rl_for_avs_amd import engine, rollout, sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    for total in a.envs:
        lo, hi = sharding.shard_range(total, rank, world)
        B = hi - lo
        G = max(1, min(a.groups, B))
        torch.manual_seed(1234)          # the same untrained policy in every run and on every rank: reproducible rows
        pol = rollout.ActorCritic(3 if a.version == "v1" else 1).to(dev)
        engs, cols = [], []
        for g in range(G):
            glo, ghi = sharding.shard_range(B, g, G)
            e_g = engine.MPCEngine(horizon=20, max_iter=a.max_iter, tol=a.tol, device=local)
            env = rollout.SyntheticIntersectionEnv(ghi - glo, device=dev, seed=rank * 97 + g, n_others=a.others,
                                                   backend=a.env_backend, env_offset=lo + glo)
            engs.append(e_g)
            cols.append(rollout.BatchedCollector(env, pol, e_g, version=a.version, algorithm=a.algorithm, n_steps=a.steps,
                                                 collision_cost=False, gather_actions=use_dist and G == 1,
                                                 seed=g, use_graph=a.graph, throughput=G > 1))
        col = cols[0] if G == 1 else rollout.PipelinedCollector(cols)
        col.collect_rollouts()                                   # warm-up (allocations, first launches)
        events = []

        def timed(inner):                                        # HIP events around the MPC call on its stream
            def call(*args, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = inner(*args, **kw)
                e1.record()
                events.append((e0, e1))
                return out
            return call

        graph = all(c._graph is not None for c in cols)          # sharded runs capture the RCCL gather inside the graph
        if not graph:                                            # a replayed graph does not pass through Python
            for e_g in engs:
                e_g.predict_batch_torch = timed(e_g.predict_batch_torch)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        stats = col.collect_rollouts()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        dist_info = None
        if use_dist and G == 1:
            assert col.gathered_actions.shape == (total, 2) and col.gathered_status.shape == (total,)
            dist_info = dict(backend=dist.get_backend(), world_size=world,
                             gathered_actions_shape=list(col.gathered_actions.shape),
                             gathered_status_shape=list(col.gathered_status.shape),
                             own_block_equals_local=bool(torch.equal(col.gathered_actions[lo:hi], col.last_mpc["act"]) and
                                                         torch.equal(col.gathered_status[lo:hi], col.last_mpc["status"])))
        dm = sum(e0.elapsed_time(e1) for e0, e1 in events) * 1e-3
        st = torch.cat([c.last_mpc["status"] for c in cols]).cpu().numpy()
        if rank == 0:
            print(json.dumps(dict(config=f"{total} envs on {world} GPU(s), {a.others} other vehicles, {a.version}/{a.algorithm}, "
                                         f"horizon 20, max_iter {a.max_iter}, tol {a.tol:g}" + (f", {G} groups on {G} streams" if G > 1 else "") + (", hipGraph step" if graph else ", eager step") + f", {cols[0].env.backend} environment",
                              envs=total, n_gpus=world, distributed=dist_info, graph_fallback_reason=cols[0].graph_fallback_reason, fused_glue=cols[0].fused_glue, groups=G, graph=bool(graph), env_backend=cols[0].env.backend,
                              steps_per_env=a.steps, env_steps_per_s=total * a.steps / dt, ms_per_step=dt / a.steps * 1e3,
                              mpc_ms_per_step=(dm / a.steps * 1e3) if events else None, episodes=stats["episodes"], crashed=stats["crashed"],
                              arrived=stats["arrived"], converged_frac=float(((st == 0) | ((st >= 5) & (st <= 7))).mean()),
                              converged_frac_rollout=1.0 - stats["mpc_unconverged"] / float(total * a.steps) * world)), flush=True)
        for e_g in engs:
            e_g.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
