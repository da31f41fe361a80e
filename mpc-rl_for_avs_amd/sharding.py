"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

MPC instances are independent (one per parallel RL env), so the batch is cut into contiguous blocks, every
rank solves its block with its own engine, and the only exchange is the all-gather of the resulting actions
(`[B/G, 2]` float64 per rank, 8 KiB at B = 4096 and G = 8 - latency-bound on xGMI) so that every learner
sees all actions, as `a2c_mpc` / `ppo_mpc` need (reference agents/ppo_mpc.py:422-432 steps every env with its
MPC action).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of `total` instances owned by `rank` (blocks differ by at most one)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_actions(local_u0, group=None):
    """Gather equally sized per-rank action blocks [b, 2] into [world*b, 2] on every rank (one collective)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world * local_u0.shape[0],) + tuple(local_u0.shape[1:]), dtype=local_u0.dtype,
                      device=local_u0.device)
    dist.all_gather_into_tensor(out, local_u0.contiguous(), group=group)
    return out


def all_gather_results(local_u0, local_status, group=None):
    """The path's one exchange as SURVEY section 8(e) states it: actions [b, 2] float64 AND solver status [b] int32 of
    every rank, in ONE collective (the status rides as a third float64 column; 24 B per instance, latency-bound).
    Returns (actions [world*b, 2] float64, status [world*b] int32) on every rank."""
    import torch
    packed = torch.cat([local_u0, local_status.to(local_u0.dtype).unsqueeze(1)], dim=1)
    full = all_gather_actions(packed, group)
    return full[:, :2], full[:, 2].to(torch.int32)


def all_gather_ragged(local_u0, total: int, group=None):
    """Gather blocks made by `shard_range` (sizes may differ by one) into the full [total, 2] array."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    width = -(-total // world)
    pad = torch.zeros((width,) + tuple(local_u0.shape[1:]), dtype=local_u0.dtype, device=local_u0.device)
    pad[: local_u0.shape[0]] = local_u0
    full = all_gather_actions(pad, group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(full[r * width: r * width + (hi - lo)])
    del rank
    return torch.cat(parts, dim=0)


def rank_devices(device, group=None):
    """Which physical device every rank of the job drives: one dict per rank (rank, local_rank, device, pci, uuid, key),
    gathered with one object collective at start-up.  `key` identifies the hardware - PCI bus id + UUID of a GPU, which
    HIP_VISIBLE_DEVICES cannot alias; host name + process id for a CPU rank of the gloo tests."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    device = torch.device(device)
    me = {"rank": dist.get_rank(group), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": str(device)}
    if device.type == "cuda":
        prop = torch.cuda.get_device_properties(device)
        me["pci"] = (f"{getattr(prop, 'pci_domain_id', 0):04x}:{getattr(prop, 'pci_bus_id', -1):02x}:"
                     f"{getattr(prop, 'pci_device_id', -1):02x}")
        me["uuid"] = str(getattr(prop, "uuid", ""))
        if getattr(prop, "pci_bus_id", None) is None and not me["uuid"]:
            # a torch / ROCm build that exposes neither a PCI id nor a UUID: every rank would get the same key and a correct
            # run would be refused - fall back to the device ordinal this process actually drives
            idx = device.index if device.index is not None else torch.cuda.current_device()
            me["key"] = f"{socket.gethostname()}/ordinal{idx}/visible={os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES', 'all'))}"
        else:
            me["key"] = f"{socket.gethostname()}/{me['pci']}/{me['uuid']}"
    else:
        me["pci"], me["uuid"] = None, None
        me["key"] = f"{socket.gethostname()}/cpu/{os.getpid()}"
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, me, group=group)
    return out


def assert_distinct_devices(ids):
    """One process per GPU: two ranks on one device would halve what a scaling run measures without failing."""
    keys = {d["key"] for d in ids}
    if len(keys) != len(ids):
        raise RuntimeError(f"{len(ids)} ranks drive {len(keys)} distinct devices: {ids}")
    return len(keys)
