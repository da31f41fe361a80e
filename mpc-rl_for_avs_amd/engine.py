"""ctypes binding of the C ABI (include/mpc_mi355x.h) - the only way Python reaches the HIP kernels.

There is no CPU fallback: if the shared library is missing, or no MI355X is visible, construction fails
loudly (`EngineError`).  Host numpy arrays are copied by the library; torch device tensors are passed
zero-copy as raw device pointers on the caller's stream (`solve_batch_torch`).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _build
from .reference_path import reference_states

FLAG_COLLISION_COST = 1
FLAG_DEVICE_PTRS = 2
FLAG_NO_SYNC = 4
FLAG_WARM_START = 8      # not in the reference: start from given / previous controls instead of zeros
FLAG_DETECT_ONLY = 16    # mpc_predict_batch: _check_collision alone (advance the detector records, no solve)
FLAG_DETECTED = 32       # mpc_predict_batch: _solve after such a call for the same observation
FLAG_THROUGHPUT = 64     # several batches in flight: always the build for four resident waves per SIMD
FLAG_STRICT_DISCONTINUITY = 128   # a solve that ends on the d = 1 discontinuity of the collision cost is reported unsolved (status 8)

STATUS_CONVERGED = 0
STATUS_MAX_ITER = 1
STATUS_FACTORIZATION = 2
STATUS_INFEASIBLE_START = 3
STATUS_STALLED = 4
STATUS_CONVERGED_ON_KINK = 5     # include/mpc_mi355x.h
STATUS_ACCEPTABLE = 6            # IPOPT's "Solved To Acceptable Level" (acceptable_tol 1e-6 for 15 iterations; casadi: success)
STATUS_ACCEPTABLE_ON_KINK = 7
STATUS_KINK_UNSOLVED = 8         # FLAG_STRICT_DISCONTINUITY: would be 5 or 7, counted as not solved


def converged(status):
    """Solved (MPC_STATUS_IS_SOLVED): 0; 5 on the d = 1 discontinuity of the collision cost; 6 / 7 IPOPT's acceptable level."""
    return (status == STATUS_CONVERGED) | ((status >= STATUS_CONVERGED_ON_KINK) & (status <= STATUS_ACCEPTABLE_ON_KINK))


class EngineError(RuntimeError):
    pass


class _Config(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32), ("horizon", ctypes.c_int32), ("dt", ctypes.c_double),
                ("max_iter", ctypes.c_int32), ("device", ctypes.c_int32), ("tol", ctypes.c_double),
                ("w_distance", ctypes.c_double), ("w_collision", ctypes.c_double),
                ("ltv_passes", ctypes.c_int32), ("stall_window", ctypes.c_int32)]


_EXPORTS = ["mpc_version", "mpc_last_error", "mpc_default_config", "mpc_default_config_sized", "mpc_create", "mpc_destroy",
            "mpc_set_reference", "mpc_solve_batch", "mpc_workspace_bytes", "mpc_predict_batch",
            "mpc_reset_env_state", "mpc_reset_env_mask", "mpc_get_env_state", "mpc_get_last_inputs",
            "mpc_ltv_solve_batch", "mpc_ltv_predict_batch", "mpc_env_state_bytes", "mpc_save_env_state",
            "mpc_set_env_state", "mpc_reserve_envs", "mpc_synth_env_step", "mpc_set_diagnostics", "mpc_get_last_paths", "mpc_policy_act", "mpc_rollout_record", "mpc_rollout_finish", "mpc_eval_nlp", "mpc_streams_overlap"]
ABI_VERSION = 8          # MPC_ABI_VERSION of include/mpc_mi355x.h this binding is written for
MAX_OTHERS = 16
_lib = None


def load_library(path: str | None = None):
    """dlopen libmpc_mi355x.so (building it first when hipcc is present and the sources are newer)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    if path is None and os.environ.get("MPC_EXPERIMENT_LIB"):
        path = os.environ["MPC_EXPERIMENT_LIB"]     # development aid: an experimental build of the same ABI (tools/)
        import sys
        print(f"[mpc engine] MPC_EXPERIMENT_LIB: loading {path} instead of the in-tree library", file=sys.stderr)
    if path is None:
        path = _build.LIB_PATH
        if _build.is_stale():
            try:
                _build.build()
            except RuntimeError as e:  # no hipcc on this machine: a prebuilt .so must travel with the tree
                if not os.path.exists(path):
                    raise EngineError(f"libmpc_mi355x.so is missing and cannot be built: {e}") from e
    if not os.path.exists(path):
        raise EngineError(f"{path} not found: the MI355X engine has no fallback path")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process break whichever comes
    # second.  Importing torch first makes the loader resolve our NEEDED libamdhip64.so.7 to that same copy,
    # so the engine, torch tensors/streams and RCCL all share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    vp, dp, ip = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
    lib.mpc_version.restype = ctypes.c_int
    if lib.mpc_version() != ABI_VERSION:        # whichever path the library came from: argument lists below are ABI-specific
        raise EngineError(f"{path} reports ABI {lib.mpc_version()}, this binding is written for ABI {ABI_VERSION}")
    lib.mpc_last_error.restype = ctypes.c_char_p
    lib.mpc_default_config.argtypes = [ctypes.POINTER(_Config)]
    lib.mpc_default_config.restype = None
    lib.mpc_default_config_sized.argtypes = [ctypes.POINTER(_Config), ctypes.c_int32]
    lib.mpc_default_config_sized.restype = ctypes.c_int
    lib.mpc_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(vp)]
    lib.mpc_create.restype = ctypes.c_int
    lib.mpc_destroy.argtypes = [vp]
    lib.mpc_destroy.restype = None
    lib.mpc_set_reference.argtypes = [vp, dp, ctypes.c_int32]
    lib.mpc_set_reference.restype = ctypes.c_int
    lib.mpc_solve_batch.argtypes = [vp, ctypes.c_int32, dp, ip, dp, dp, vp, dp, ctypes.c_int32, ctypes.c_uint32,
                                    dp, dp, dp, ip, ip, vp]
    lib.mpc_solve_batch.restype = ctypes.c_int
    lib.mpc_workspace_bytes.argtypes = [vp, ctypes.c_int32, ctypes.c_int32]
    lib.mpc_workspace_bytes.restype = ctypes.c_int64
    lib.mpc_predict_batch.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, dp, dp, ctypes.c_uint32, dp, ip, ip, vp]
    lib.mpc_predict_batch.restype = ctypes.c_int
    lib.mpc_ltv_solve_batch.argtypes = [vp, ctypes.c_int32, dp, ctypes.c_uint32, dp, dp, dp, ip, ip, ip, vp]
    lib.mpc_ltv_solve_batch.restype = ctypes.c_int
    lib.mpc_ltv_predict_batch.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_uint32, dp, ip, ip, vp]
    lib.mpc_ltv_predict_batch.restype = ctypes.c_int
    lib.mpc_reset_env_state.argtypes = [vp, ip, ctypes.c_int32, vp]
    lib.mpc_reset_env_state.restype = ctypes.c_int
    lib.mpc_reset_env_mask.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_uint32, vp]
    lib.mpc_reset_env_mask.restype = ctypes.c_int
    lib.mpc_get_env_state.argtypes = [vp, ctypes.c_int32, ip, ip, ip, ip, ip, dp]
    lib.mpc_get_env_state.restype = ctypes.c_int
    lib.mpc_env_state_bytes.argtypes = []
    lib.mpc_env_state_bytes.restype = ctypes.c_int64
    lib.mpc_save_env_state.argtypes = [vp, ctypes.c_int32, vp]
    lib.mpc_save_env_state.restype = ctypes.c_int
    lib.mpc_set_env_state.argtypes = [vp, ctypes.c_int32, vp]
    lib.mpc_set_env_state.restype = ctypes.c_int
    lib.mpc_reserve_envs.argtypes = [vp, ctypes.c_int32]
    lib.mpc_reserve_envs.restype = ctypes.c_int
    lib.mpc_synth_env_step.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_uint64, ctypes.c_int32, dp, ctypes.c_int32] + [vp] * 15 + [ctypes.c_int32, vp]
    lib.mpc_synth_env_step.restype = ctypes.c_int
    lib.mpc_get_last_inputs.argtypes = [vp, ctypes.c_int32, dp, ip, dp, vp, dp, ip]
    lib.mpc_get_last_inputs.restype = ctypes.c_int
    lib.mpc_set_diagnostics.argtypes = [vp, ctypes.c_int32]
    lib.mpc_set_diagnostics.restype = ctypes.c_int
    lib.mpc_get_last_paths.argtypes = [vp, ctypes.c_int32, dp, ip, vp]
    lib.mpc_get_last_paths.restype = ctypes.c_int
    lib.mpc_policy_act.argtypes = [ctypes.c_int32] * 4 + [vp] * 10 + [ctypes.c_uint64, ctypes.c_int32, vp] + [ctypes.c_int32] * 2 + [vp] * 6
    lib.mpc_policy_act.restype = ctypes.c_int
    lib.mpc_rollout_record.argtypes = [ctypes.c_int32] * 6 + [vp] * 22
    lib.mpc_rollout_record.restype = ctypes.c_int
    lib.mpc_rollout_finish.argtypes = [ctypes.c_int32] * 6 + [vp] * 4 + [ctypes.c_double] * 2 + [vp] * 3
    lib.mpc_rollout_finish.restype = ctypes.c_int
    lib.mpc_eval_nlp.argtypes = [vp, ctypes.c_int32] + [vp] * 5 + [ctypes.c_int32, ctypes.c_uint32] + [vp] * 4
    lib.mpc_eval_nlp.restype = ctypes.c_int
    lib.mpc_streams_overlap.argtypes = [ctypes.c_int32, vp, vp, vp]
    lib.mpc_streams_overlap.restype = ctypes.c_int
    _lib = lib
    return lib


def streams_overlap(a, b, device: int = 0) -> bool:
    """True if kernels enqueued on the torch streams `a` and `b` run side by side (mpc_streams_overlap: two streams that the HIP
    runtime put on one hardware queue serialise)."""
    lib = load_library()
    out = ctypes.c_int32(-1)
    rc = lib.mpc_streams_overlap(int(device), ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream), ctypes.byref(out))
    if rc != 0:
        raise EngineError(f"mpc_streams_overlap failed ({rc}): {lib.mpc_last_error().decode()}")
    return out.value == 1


def concurrent_streams(n: int, device=0, candidates: int = 32):
    """`n` torch streams on which independent batches really are in flight together: torch hands out the streams of its pool
    round robin, the runtime spreads them over GPU_MAX_HW_QUEUES hardware queues as it sees fit, and two that share a queue
    serialise (2.6 against 3.45 M solves/s with 8 batches of 4096 in flight, tools/gpu_queues.py) - so every candidate is
    probed against the streams already chosen and kept only if it overlaps with each.  Raises if fewer than `n` of the
    `candidates` qualify (set GPU_MAX_HW_QUEUES >= n before the process touches the GPU)."""
    import torch
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    chosen, seen = [], set()
    for _ in range(candidates):
        s = torch.cuda.Stream(dev)
        if s.cuda_stream in seen:
            continue
        seen.add(s.cuda_stream)
        if all(streams_overlap(s, c, idx) for c in chosen):
            chosen.append(s)
            if len(chosen) == n:
                return chosen
    raise EngineError(f"only {len(chosen)} of {len(seen)} streams run concurrently on device {idx}; "
                      f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(unset: 4)')}")


def _ptr(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


class MPCEngine:
    """One engine per (process, GPU).  `solve_batch` replaces B calls of `PureMPC_Agent._solve`."""

    def __init__(self, horizon: int = 20, dt: float = 0.1, max_iter: int = 100, tol: float = 1e-8,
                 w_distance: float = 10.0, w_collision: float = 1.0, device: int = 0, ref_table=None,
                 ltv_passes: int = 1, stall_window: int = 0):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        cfg = _Config()
        # the sized variant: a _Config that no longer matches the library's mpc_config is refused, not overrun
        rc = self._lib.mpc_default_config_sized(ctypes.byref(cfg), ctypes.sizeof(cfg))
        if rc != 0:
            raise EngineError(f"mpc_default_config_sized failed ({rc}): {self._lib.mpc_last_error().decode()}")
        cfg.horizon, cfg.dt, cfg.max_iter, cfg.tol = int(horizon), float(dt), int(max_iter), float(tol)
        cfg.w_distance, cfg.w_collision, cfg.device = float(w_distance), float(w_collision), int(device)
        cfg.stall_window = int(stall_window)  # 0 = off: see include/mpc_mi355x.h
        cfg.ltv_passes = int(ltv_passes)     # iterative-linear agent only: trip count of agents/pure_mpc_linear.py:189
        rc = self._lib.mpc_create(ctypes.byref(cfg), ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            raise EngineError(f"mpc_create failed ({rc}): {self._lib.mpc_last_error().decode()}")
        self.horizon, self.dt, self.device = int(horizon), float(dt), int(device)
        self.set_reference(reference_states(dt) if ref_table is None else ref_table)

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self._lib.mpc_last_error().decode()}")

    def set_reference(self, ref_table):
        ref = np.ascontiguousarray(ref_table, dtype=np.float64)
        if ref.ndim != 2 or ref.shape[1] != 4:
            raise ValueError(f"reference table must be [M, 4], got {ref.shape}")
        self.ref_table = ref
        self._check(self._lib.mpc_set_reference(self._h, _ptr(ref), ref.shape[0]), "mpc_set_reference")

    def workspace_bytes(self, B, V=0):
        """LDS bytes per workgroup for a batch of B (V > 0: collision-cost variant with V vehicles)."""
        return int(self._lib.mpc_workspace_bytes(self._h, int(B), int(V)))

    # ------------------------------------------------------------------ host (numpy) path
    def solve_batch(self, state, ego_index, weights, is_collide, vref=None, others=None, collision_cost=False,
                    want_trajectories=True, u_init=None, strict_discontinuity=False):
        """Solve B instances given host arrays; returns dict(u0, U, X, status, iters).
        u_init [B, N, 2] (optional, not in the reference): initial controls instead of the cold start.
        strict_discontinuity (MPC_FLAG_STRICT_DISCONTINUITY): a solve that ends on the d = 1 discontinuity of the collision
        cost is reported as not solved (status 8) with its last iterate, as the reference's IPOPT would report it."""
        N = self.horizon
        state = np.ascontiguousarray(state, dtype=np.float64)
        if state.ndim != 2 or state.shape[1] != 4:
            raise ValueError(f"state must be [B, 4], got {state.shape}")
        B = state.shape[0]
        ego_index = np.ascontiguousarray(ego_index, dtype=np.int32).reshape(B)
        weights = np.ascontiguousarray(weights, dtype=np.float64).reshape(B, 3)
        is_collide = np.ascontiguousarray(is_collide, dtype=np.uint8).reshape(B)
        if vref is not None:
            vref = np.ascontiguousarray(vref, dtype=np.float64)
            if vref.shape != (B, N + 1):
                raise ValueError(f"vref must be [B, N+1] = {(B, N + 1)}, got {vref.shape}")
        V = 0
        if others is not None:
            others = np.ascontiguousarray(others, dtype=np.float64)
            if others.ndim != 3 or others.shape[0] != B or others.shape[2] != 4:
                raise ValueError(f"others must be [B, V, 4], got {others.shape}")
            V = others.shape[1]
        u0 = np.empty((B, 2))
        U = np.empty((B, N, 2)) if want_trajectories else None
        X = np.empty((B, N + 1, 4)) if want_trajectories else None
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        flags = (FLAG_COLLISION_COST if collision_cost else 0) | (FLAG_STRICT_DISCONTINUITY if strict_discontinuity else 0)
        if u_init is not None:
            U = np.array(u_init, dtype=np.float64, order="C")       # in: initial controls, out: solution
            if U.shape != (B, N, 2):
                raise ValueError(f"u_init must be [B, N, 2] = {(B, N, 2)}, got {U.shape}")
            flags |= FLAG_WARM_START
        rc = self._lib.mpc_solve_batch(self._h, B, _ptr(state), _ptr(ego_index), _ptr(vref), _ptr(weights),
                                       _ptr(is_collide), _ptr(others), V, flags, _ptr(u0), _ptr(U), _ptr(X),
                                       _ptr(status), _ptr(iters), None)
        self._check(rc, "mpc_solve_batch")
        return dict(u0=u0, U=U, X=X, status=status, iters=iters)

    # ------------------------------------------------------------------ device (torch) path
    def solve_batch_torch(self, state, ego_index, weights, is_collide, vref=None, others=None,
                          collision_cost=False, out=None, sync=False, throughput=False, strict_discontinuity=False):
        """Zero-copy solve on torch CUDA(=HIP) tensors, enqueued on torch's current stream.

        dtypes: state/weights/vref/others float64, ego_index int32, is_collide uint8; all contiguous and on
        this engine's device.  Returns dict(u0, status, iters) of device tensors (reused when `out` given)."""
        import torch
        B = state.shape[0]
        dev = state.device
        for name, t, dt_ in (("state", state, torch.float64), ("ego_index", ego_index, torch.int32),
                             ("weights", weights, torch.float64), ("is_collide", is_collide, torch.uint8),
                             ("vref", vref, torch.float64), ("others", others, torch.float64)):
            if t is None:
                continue
            if t.dtype != dt_ or not t.is_contiguous() or t.device != dev or not t.is_cuda:
                raise ValueError(f"{name}: expected contiguous {dt_} tensor on {dev}")
        if out is None:
            out = dict(u0=torch.empty((B, 2), dtype=torch.float64, device=dev),
                       status=torch.empty(B, dtype=torch.int32, device=dev),
                       iters=torch.empty(B, dtype=torch.int32, device=dev))
        V = 0 if others is None else int(others.shape[1])
        flags = FLAG_DEVICE_PTRS | (FLAG_COLLISION_COST if collision_cost else 0) | (0 if sync else FLAG_NO_SYNC) | \
            (FLAG_THROUGHPUT if throughput else 0) | \
            (FLAG_STRICT_DISCONTINUITY if strict_discontinuity else 0)     # throughput: several batches in flight (MPC_FLAG_THROUGHPUT)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self._lib.mpc_solve_batch(self._h, B, p(state), p(ego_index), p(vref), p(weights), p(is_collide),
                                       p(others), V, flags, p(out["u0"]), p(out.get("U")), p(out.get("X")),
                                       p(out["status"]), p(out["iters"]), stream)
        self._check(rc, "mpc_solve_batch")
        return out

    # ------------------------------------------------------------------ observation-level path
    def detect_batch(self, obs):
        """`_check_collision` alone (agents/pure_mpc.py:552-676) for obs[B, vehicles_count, 8]: advances the detector
        records; `env_state` serves the result.  Follow with `predict_batch(..., detected=True)` for the same obs."""
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        if obs.ndim != 3 or obs.shape[2] != 8:
            raise ValueError(f"obs must be [B, vehicles_count, 8], got {obs.shape}")
        B, rows = obs.shape[:2]
        rc = self._lib.mpc_predict_batch(self._h, B, _ptr(obs), rows, None, None, FLAG_DETECT_ONLY, None, None, None, None)
        self._check(rc, "mpc_predict_batch")

    def predict_batch(self, obs, weights, ref_speed=None, collision_cost=False, warm_start=False, detected=False,
                      strict_discontinuity=False):
        """obs[B, vehicles_count, 8] float32 -> dict(act[B, 2], status, iters): parsing, collision detector (with the
        per-environment memory kept inside the engine), speed-profile rewrite and solve, all on the device.
        warm_start (not in the reference): each environment starts from its previous solution advanced one stage.
        detected: `detect_batch` has already advanced the records for this observation."""
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        if obs.ndim != 3 or obs.shape[2] != 8:
            raise ValueError(f"obs must be [B, vehicles_count, 8], got {obs.shape}")
        B, rows = obs.shape[:2]
        weights = np.ascontiguousarray(weights, dtype=np.float64).reshape(B, 3)
        rs = None if ref_speed is None else np.ascontiguousarray(ref_speed, dtype=np.float64).reshape(B)
        act = np.empty((B, 2))
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        flags = (FLAG_COLLISION_COST if collision_cost else 0) | (FLAG_WARM_START if warm_start else 0) | \
            (FLAG_DETECTED if detected else 0) | (FLAG_STRICT_DISCONTINUITY if strict_discontinuity else 0)
        rc = self._lib.mpc_predict_batch(self._h, B, _ptr(obs), rows, _ptr(weights), _ptr(rs), flags, _ptr(act),
                                         _ptr(status), _ptr(iters), None)
        self._check(rc, "mpc_predict_batch")
        return dict(act=act, status=status, iters=iters)

    def predict_batch_torch(self, obs, weights, ref_speed=None, collision_cost=False, out=None, sync=False,
                            warm_start=False, throughput=False, strict_discontinuity=False):
        """Zero-copy variant on torch device tensors (obs float32 [B, R, 8], weights float64 [B, 3], ref_speed float64
        [B] or None), enqueued on torch's current stream.  Returns dict(act, status, iters) of device tensors."""
        import torch
        B, rows = int(obs.shape[0]), int(obs.shape[1])
        dev = obs.device
        for name, t, dt_ in (("obs", obs, torch.float32), ("weights", weights, torch.float64),
                             ("ref_speed", ref_speed, torch.float64)):
            if t is None:
                continue
            if t.dtype != dt_ or not t.is_contiguous() or t.device != dev or not t.is_cuda:
                raise ValueError(f"{name}: expected contiguous {dt_} tensor on {dev}")
        if obs.dim() != 3 or obs.shape[2] != 8 or tuple(weights.shape) != (B, 3):
            raise ValueError("obs must be [B, vehicles_count, 8] and weights [B, 3]")
        if out is None:
            out = dict(act=torch.empty((B, 2), dtype=torch.float64, device=dev),
                       status=torch.empty(B, dtype=torch.int32, device=dev),
                       iters=torch.empty(B, dtype=torch.int32, device=dev))
        flags = FLAG_DEVICE_PTRS | (FLAG_COLLISION_COST if collision_cost else 0) | (0 if sync else FLAG_NO_SYNC) | \
            (FLAG_WARM_START if warm_start else 0) | (FLAG_THROUGHPUT if throughput else 0) | \
            (FLAG_STRICT_DISCONTINUITY if strict_discontinuity else 0)   # throughput: several groups in flight
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self._lib.mpc_predict_batch(self._h, B, p(obs), rows, p(weights), p(ref_speed), flags, p(out["act"]),
                                         p(out["status"]), p(out["iters"]), stream)
        self._check(rc, "mpc_predict_batch")
        return out

    def eval_nlp(self, ego_index, weights, is_collide, X, U, vref=None, others=None, collision_cost=False):
        """Diagnostics (mpc_eval_nlp): objective f [B] and model successors x_next [B, N, 4] of the points (X [B, N+1, 4],
        U [B, N, 2]) as the solve kernel's own code evaluates them."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        U = np.ascontiguousarray(U, dtype=np.float64)
        B, N = U.shape[0], self.horizon
        if X.shape != (B, N + 1, 4) or U.shape != (B, N, 2):
            raise ValueError(f"X must be [B, {N + 1}, 4] and U [B, {N}, 2], got {X.shape} and {U.shape}")
        ego = np.ascontiguousarray(ego_index, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        c = np.ascontiguousarray(is_collide, dtype=np.uint8)
        if ego.shape != (B,) or w.shape != (B, 3) or c.shape != (B,):
            raise ValueError("ego_index [B], weights [B, 3], is_collide [B] expected")
        vr = None if vref is None else np.ascontiguousarray(vref, dtype=np.float64)
        if vr is not None and vr.shape != (B, N + 1):
            raise ValueError(f"vref must be [B, {N + 1}]")
        oth = None if others is None else np.ascontiguousarray(others, dtype=np.float64)
        if oth is not None and (oth.ndim != 3 or oth.shape[0] != B or oth.shape[2] != 4):
            raise ValueError("others must be [B, V, 4]")
        V = 0 if oth is None else oth.shape[1]
        f, xn = np.empty(B), np.empty((B, N, 4))
        rc = self._lib.mpc_eval_nlp(self._h, B, _ptr(ego), _ptr(vr), _ptr(w), _ptr(c), _ptr(oth), V,
                                    FLAG_COLLISION_COST if collision_cost else 0, _ptr(X), _ptr(U), _ptr(f), _ptr(xn))
        self._check(rc, "mpc_eval_nlp")
        return f, xn

    # ------------------------------------------------------------------ iterative-linear MPC (pure_mpc_linear.py)
    def ltv_solve_batch(self, state, U, want_traj=False):
        """B calls of IterativeLinearMPC_Agent._solve.  state[B, 4] = x, y, v, yaw; U[B, N, 2] = stored profile (oa, od).
        Returns dict(u0, U (new profile; the old one where status != 0), status, iters, target_index[, X])."""
        state = np.ascontiguousarray(state, dtype=np.float64)
        B, N = state.shape[0], self.horizon
        if state.shape != (B, 4):
            raise ValueError(f"state must be [B, 4], got {state.shape}")
        U = np.array(U, dtype=np.float64, order="C", copy=True)
        if U.shape != (B, N, 2):
            raise ValueError(f"U must be [B, {N}, 2], got {U.shape}")
        u0 = np.empty((B, 2))
        X = np.empty((B, N + 1, 4)) if want_traj else None
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        target = np.empty(B, dtype=np.int32)
        rc = self._lib.mpc_ltv_solve_batch(self._h, B, _ptr(state), 0, _ptr(u0), _ptr(U), _ptr(X), _ptr(status),
                                           _ptr(iters), _ptr(target), None)
        self._check(rc, "mpc_ltv_solve_batch")
        out = dict(u0=u0, U=U, status=status, iters=iters, target_index=target)
        if want_traj:
            out["X"] = X
        return out

    def ltv_solve_batch_torch(self, state, U, out=None, sync=False):
        """Zero-copy variant: state float64 [B, 4] and U float64 [B, N, 2] device tensors; U is updated in place.
        Returns dict(u0, status, iters) of device tensors."""
        import torch
        B, N, dev = int(state.shape[0]), self.horizon, state.device
        for name, t, shape in (("state", state, (B, 4)), ("U", U, (B, N, 2))):
            if t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda or t.device != dev or \
                    tuple(t.shape) != shape:
                raise ValueError(f"{name}: expected contiguous float64 tensor {shape} on {dev}")
        if out is None:
            out = dict(u0=torch.empty((B, 2), dtype=torch.float64, device=dev),
                       status=torch.empty(B, dtype=torch.int32, device=dev),
                       iters=torch.empty(B, dtype=torch.int32, device=dev))
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self._lib.mpc_ltv_solve_batch(self._h, B, p(state), FLAG_DEVICE_PTRS | (0 if sync else FLAG_NO_SYNC),
                                           p(out["u0"]), p(U), None, p(out["status"]), p(out["iters"]), None, stream)
        self._check(rc, "mpc_ltv_solve_batch")
        return out

    def ltv_predict_batch(self, obs):
        """obs[B, vehicles_count, 8] float32 -> dict(act, status, iters): Agent.predict of the iterative-linear agent
        with the per-environment stored profile kept inside the engine."""
        obs = np.ascontiguousarray(obs, dtype=np.float32)
        if obs.ndim != 3 or obs.shape[2] != 8:
            raise ValueError(f"obs must be [B, vehicles_count, 8], got {obs.shape}")
        B, rows = obs.shape[:2]
        act = np.empty((B, 2))
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        rc = self._lib.mpc_ltv_predict_batch(self._h, B, _ptr(obs), rows, 0, _ptr(act), _ptr(status), _ptr(iters), None)
        self._check(rc, "mpc_ltv_predict_batch")
        return dict(act=act, status=status, iters=iters)

    def ltv_predict_batch_torch(self, obs, out=None, sync=False):
        """Zero-copy variant on a float32 device tensor [B, R, 8], enqueued on torch's current stream."""
        import torch
        if obs.dtype != torch.float32 or not obs.is_contiguous() or not obs.is_cuda or obs.dim() != 3 or obs.shape[2] != 8:
            raise ValueError("obs: expected contiguous float32 device tensor [B, vehicles_count, 8]")
        B, rows, dev = int(obs.shape[0]), int(obs.shape[1]), obs.device
        if out is None:
            out = dict(act=torch.empty((B, 2), dtype=torch.float64, device=dev),
                       status=torch.empty(B, dtype=torch.int32, device=dev),
                       iters=torch.empty(B, dtype=torch.int32, device=dev))
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self._lib.mpc_ltv_predict_batch(self._h, B, p(obs), rows, FLAG_DEVICE_PTRS | (0 if sync else FLAG_NO_SYNC),
                                             p(out["act"]), p(out["status"]), p(out["iters"]), stream)
        self._check(rc, "mpc_ltv_predict_batch")
        return out

    def reset_env_state(self, env_ids=None):
        """Episode boundaries: fresh detector state for the given environments (None = all)."""
        if env_ids is None:
            rc = self._lib.mpc_reset_env_state(self._h, None, -1, None)
        else:
            ids = np.ascontiguousarray(env_ids, dtype=np.int32).reshape(-1)
            rc = self._lib.mpc_reset_env_state(self._h, _ptr(ids), ids.size, None)
        self._check(rc, "mpc_reset_env_state")

    def reset_env_mask_torch(self, done, warm_only=False):
        """Same from a uint8 device mask [B], enqueued on torch's current stream; warm_only: forget just the warm-start
        memory of those environments and keep their detector state."""
        import torch
        if done.dtype != torch.uint8 or not done.is_contiguous() or not done.is_cuda:
            raise ValueError("done: expected contiguous uint8 device tensor")
        stream = ctypes.c_void_p(torch.cuda.current_stream(done.device).cuda_stream)
        rc = self._lib.mpc_reset_env_mask(self._h, int(done.numel()), ctypes.c_void_p(done.data_ptr()),
                                          FLAG_DEVICE_PTRS | FLAG_NO_SYNC | (FLAG_WARM_START if warm_only else 0), stream)
        self._check(rc, "mpc_reset_env_mask")

    def env_state(self, B):
        """Detector state of environments 0..B-1 (host copies): is_collide, ego_index, collision_memory, stop_index,
        conflict_index[B, 16] (-1 = none), conflict_points[B, 16, 2] (NaN = none)."""
        o = dict(is_collide=np.empty(B, np.int32), ego_index=np.empty(B, np.int32),
                 collision_memory=np.empty(B, np.int32), stop_index=np.empty(B, np.int32),
                 conflict_index=np.empty((B, MAX_OTHERS), np.int32), conflict_points=np.empty((B, MAX_OTHERS, 2)))
        rc = self._lib.mpc_get_env_state(self._h, B, _ptr(o["is_collide"]), _ptr(o["ego_index"]),
                                         _ptr(o["collision_memory"]), _ptr(o["stop_index"]), _ptr(o["conflict_index"]),
                                         _ptr(o["conflict_points"]))
        self._check(rc, "mpc_get_env_state")
        return o

    def save_env_state(self, B) -> np.ndarray:
        """Checkpoint of the detector records of environments 0..B-1: opaque bytes [B, record size]."""
        buf = np.zeros((B, int(self._lib.mpc_env_state_bytes())), dtype=np.uint8)
        self._check(self._lib.mpc_save_env_state(self._h, B, _ptr(buf)), "mpc_save_env_state")
        return buf

    def load_env_state(self, records: np.ndarray):
        """Put back what `save_env_state` returned (this or another engine)."""
        records = np.ascontiguousarray(records, dtype=np.uint8)
        if records.ndim != 2 or records.shape[1] != int(self._lib.mpc_env_state_bytes()):
            raise ValueError("records: expected [B, mpc_env_state_bytes()] uint8")
        self._check(self._lib.mpc_set_env_state(self._h, records.shape[0], _ptr(records)), "mpc_set_env_state")

    def reserve_envs(self, B):
        """Size the per-environment buffers now (required before capturing a step for B environments in a hipGraph)."""
        self._check(self._lib.mpc_reserve_envs(self._h, int(B)), "mpc_reserve_envs")

    def last_inputs(self, B, vehicles_count):
        """The problem data the last predict_batch derived from its observations (host copies)."""
        N, V = self.horizon, max(int(vehicles_count) - 1, 1)
        o = dict(state=np.empty((B, 4)), ego_index=np.empty(B, np.int32), vref=np.empty((B, N + 1)),
                 is_collide=np.empty(B, np.uint8), others=np.empty((B, V, 4)), nveh=np.empty(B, np.int32))
        rc = self._lib.mpc_get_last_inputs(self._h, B, _ptr(o["state"]), _ptr(o["ego_index"]), _ptr(o["vref"]),
                                           _ptr(o["is_collide"]), _ptr(o["others"]), _ptr(o["nveh"]))
        self._check(rc, "mpc_get_last_inputs")
        return o

    def set_diagnostics(self, on=True):
        """Keep the detector's polylines of every following predict_batch (`last_paths`); parity tests only."""
        self._check(self._lib.mpc_set_diagnostics(self._h, 1 if on else 0), "mpc_set_diagnostics")

    def last_paths(self, B, vehicles_count):
        """Polylines of the last predict_batch: ego_path[B, 31, 2], ego_len[B] (0 where the environment replayed its
        collision memory), agent_paths[B, V, 31, 2] float32."""
        V = max(int(vehicles_count) - 1, 1)
        o = dict(ego_path=np.zeros((B, 31, 2)), ego_len=np.zeros(B, np.int32), agent_paths=np.zeros((B, V, 31, 2), np.float32))
        rc = self._lib.mpc_get_last_paths(self._h, B, _ptr(o["ego_path"]), _ptr(o["ego_len"]), _ptr(o["agent_paths"]))
        self._check(rc, "mpc_get_last_paths")
        return o

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mpc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
