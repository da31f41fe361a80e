#!/usr/bin/env python3
"""Cycle attribution inside the observation preamble (tools/ubench/preamble_sections.hip: the product's preamble_env_wave with a
ticking context): a 256-environment rollout of the synthetic intersection is stepped with the engine, and before every step the
instrumented kernel runs on the same observations and on COPIES of the detector records.  Per section: mean / p90 / max cycles
over the (environment, step) pairs, apart for calls that run the full detection and calls that replay the collision memory;
and per step the slowest environment (a launch lasts as long as that one).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libpreamble_sections.so tools/ubench/preamble_sections.hip
    python tools/gpu_preamble_sections.py > profiles/rNN_preamble_sections.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mpc_rl_for_avs_amd import rollout, engine
from mpc_rl_for_avs_amd.reference_path import reference_states

NAMES = ["parse", "nearest point", "speed ramp", "arc length", "ego points", "vehicle paths", "hit tests", "collinear stretches",
         "crossings (lane j)", "candidate tests", "export", "record + profile"]
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libpreamble_sections.so"))
lib.preamble_sections_record_bytes.restype = ctypes.c_longlong
assert lib.preamble_sections_count() == len(NAMES)
dev = torch.device("cuda", 0)
B, STEPS, N = 256, int(os.environ.get("STEPS", "48")), 20
ref = reference_states()
M = len(ref)
r6 = np.zeros(M * 6)
r6[:M * 5] = np.stack([ref[:, 0], ref[:, 1], ref[:, 3], np.sin(ref[:, 3]), np.cos(ref[:, 3])], axis=1).ravel()
r6[M * 5:] = ref[:, 2]
d_ref = torch.as_tensor(r6, device=dev)
env = rollout.SyntheticIntersectionEnv(B, device=dev, seed=0, n_others=4)
eng = engine.MPCEngine(horizon=N, max_iter=100)
obs = env.reset()
w = torch.ones((B, 3), dtype=torch.float64, device=dev)
out = eng.predict_batch_torch(obs, w, None, sync=True)          # sizes the handle's per-environment records
eng.reset_env_state()
assert eng.save_env_state(1).shape[1] == lib.preamble_sections_record_bytes()
p = lambda x: ctypes.c_void_p(x.data_ptr())
rows, worst = {0: [], 1: []}, []
for step in range(STEPS):
    rec = torch.as_tensor(eng.save_env_state(B), device=dev)
    cyc = torch.zeros((B, len(NAMES)), dtype=torch.int64, device=dev)
    o = obs.contiguous()
    rc = lib.preamble_sections(B, p(o), int(o.shape[1]), p(d_ref), M, N, ctypes.c_double(0.1), p(rec), max(int(o.shape[1]) - 1, 1), 1, p(cyc))
    assert rc == 0, rc
    c = cyc.cpu().numpy().astype(np.float64)
    replay = c[:, 2] == 0                                        # no speed ramp: the detector replayed its memory
    rows[0].append(c[~replay])
    rows[1].append(c[replay])
    worst.append((c.sum(axis=1).max(), int((~replay).sum())))
    out = eng.predict_batch_torch(obs, w, None, sync=True)
    obs, rew, done, info = env.step(out["act"])
print(f"# {B} environments x {STEPS} steps, 4 other vehicles, cycles of the shader clock (2.4 GHz) per environment and call")
for r, label in ((0, "FULL DETECTION"), (1, "REPLAY of the collision memory")):
    m = np.concatenate(rows[r]) if rows[r] else np.zeros((0, len(NAMES)))
    if not len(m):
        continue
    print(f"{label}: {len(m)} calls")
    for i, n in enumerate(NAMES):
        print(f"   {n:22s} mean {m[:, i].mean():8.0f}   p90 {np.percentile(m[:, i], 90):8.0f}   max {m[:, i].max():8.0f}")
    tot = m.sum(axis=1)
    print(f"   {'total':22s} mean {tot.mean():8.0f}   p90 {np.percentile(tot, 90):8.0f}   max {tot.max():8.0f}   = {tot.mean() / 2400:.1f} / "
          f"{np.percentile(tot, 90) / 2400:.1f} / {tot.max() / 2400:.1f} us")
ws = np.array([x[0] for x in worst])
print("slowest environment of each step (what a launch waits for), us: " + " ".join(f"{x / 2400:.0f}" for x in ws))
print(f"   mean {ws.mean() / 2400:.1f} us; steps in which every environment runs the full detection: "
      f"{[i for i, x in enumerate(worst) if x[1] == B]}")
