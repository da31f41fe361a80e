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
        eng = engine.MPCEngine(horizon=20, max_iter=100, device=local)
        env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=rank, n_others=a.others)
        pol = rollout.ActorCritic(3 if a.version == "v1" else 1).to(dev)
        col = rollout.BatchedCollector(env, pol, eng, version=a.version, algorithm="ppo", n_steps=a.steps,
                                       collision_cost=False, gather_actions=use_dist)
        col.collect_rollouts()                                   # warm-up (allocations, first launches)
        events = []
        inner = eng.predict_batch_torch

        def timed_predict(*args, **kw):                          # HIP events around the MPC call on torch's stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = inner(*args, **kw)
            e1.record()
            events.append((e0, e1))
            return out

        eng.predict_batch_torch = timed_predict
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
            assert col.gathered_actions.shape == (total, 2)
        dm = sum(e0.elapsed_time(e1) for e0, e1 in events) * 1e-3
        st = col.last_mpc["status"].cpu().numpy()
        if rank == 0:
            print(json.dumps(dict(config=f"{total} envs on {world} GPU(s), {a.others} other vehicles, {a.version}/ppo, "
                                         f"horizon 20", envs=total, n_gpus=world,
                              steps_per_env=a.steps, env_steps_per_s=total * a.steps / dt, ms_per_step=dt / a.steps * 1e3,
                              mpc_ms_per_step=dm / a.steps * 1e3, episodes=stats["episodes"], crashed=stats["crashed"],
                              arrived=stats["arrived"], converged_frac=float((st == 0).mean()))), flush=True)
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
