#!/usr/bin/env python3
"""Cycle attribution inside the wave-cooperative solver (tools/ubench/wave_sections.hip): per-section cycles per
iteration for a lone instance (B=1: the straggler regime) and for the whole headline batch."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mpc_rl_for_avs_amd import synth
from mpc_rl_for_avs_amd.reference_path import reference_states

NAMES = ["prep", "adjoint", "dualres", "ric_assembly", "ric_operands", "ric_T", "ric_H", "ric_2x2", "ric_schur",
         "linear", "ratios", "roll_dyn", "roll_cost", "dualupd", "r_feedback", "r_clamp", "r_dyn", "r_store", "r_check", "r_proj", "r_ldsw"]
lib = ctypes.CDLL(os.environ.get("WAVE_SECTIONS_LIB") or os.path.join(ROOT, "tools", "ubench", "libwave_sections.so"))
assert lib.wave_sections_count() == len(NAMES)
dev = torch.device("cuda", 0)
ref = reference_states()
r6 = np.zeros(85 * 6)
r6[:85 * 5] = np.stack([ref[:, 0], ref[:, 1], ref[:, 3], np.sin(ref[:, 3]), np.cos(ref[:, 3])], axis=1).ravel()
r6[85 * 5:] = ref[:, 2]
d_ref = torch.as_tensor(r6, device=dev)


def run(inp, idx, cc, V, label):
    t = lambda x, dt: torch.as_tensor(np.ascontiguousarray(x[idx]), dtype=dt, device=dev)
    B = len(idx)
    a = dict(state=t(inp["state"], torch.float64), ego=t(inp["ego_index"], torch.int32), vref=t(inp["vref"], torch.float64),
             w=t(inp["weights"], torch.float64), c=t(inp["is_collide"], torch.uint8), o=t(inp["others"], torch.float64))
    u0 = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    st = torch.zeros(B, dtype=torch.int32, device=dev)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    cyc = torch.zeros((B, len(NAMES)), dtype=torch.int64, device=dev)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    for _ in range(2):
        cyc.zero_()
        rc = lib.wave_sections(B, V, int(cc), 100, p(d_ref), 85, p(a["state"]), p(a["ego"]), p(a["vref"]), p(a["w"]),
                               p(a["c"]), p(a["o"]), p(u0), p(st), p(it), p(cyc))
        assert rc == 0
    c = cyc.cpu().numpy().astype(np.float64)
    iters = it.cpu().numpy()
    tot = c.sum(axis=1)
    print(f"--- {label}: B={B}, iterations mean {iters.mean():.1f} max {iters.max()}, cycles/iteration "
          f"{(tot / np.maximum(iters, 1)).mean():.0f} (shader clock cycles)")
    share = c.sum(axis=0) / tot.sum()
    per_it = c.sum(axis=0) / iters.sum()
    for n, s, q in zip(NAMES, share, per_it):
        print(f"   {n:12s} {100 * s:5.1f} %   {q:8.1f} ticks/iteration")
    ric = share[3:9].sum()
    print(f"   riccati total {100 * ric:.1f} %, rollout total {100 * (share[11:13].sum() + share[14:].sum()):.1f} % "
          f"(the r_* rows are the parts of the rollout stages, timed in lane 0; roll_dyn is what is left of that phase)")


inp = synth.solver_inputs(4096, 8, seed=0, N=20)
run(inp, np.array([0]), True, 8, "typical instance alone")
run(inp, np.array([77]), True, 8, "straggler (instance 77) alone")
if "--batch" in sys.argv:
    run(inp, np.arange(4096), True, 8, "whole batch (config 3)")
