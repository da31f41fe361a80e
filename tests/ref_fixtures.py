"""TEST INFRASTRUCTURE - helpers shared by the CPU and GPU tests that replay the fixtures the REFERENCE's own code produced
(tests/golden/reference_random.npz, reference_sequences.npz, reference_distance_cost.npz; generator
tests/golden/make_golden.py, stand-ins for the absent third-party packages tests/golden/standins.py)."""
import os

import numpy as np

from conftest import GOLDEN

# layout of one detector record (csrc/mpc_preamble.hpp: EnvState; mpc_env_state_bytes() = 672)
ENV_DTYPE = np.dtype([("collision_memory", "<i4"), ("has_memorized", "<i4"), ("n_memorized", "<i4"), ("n_conflict", "<i4"),
                      ("is_collide", "<i4"), ("ego_index", "<i4"), ("stop_index1", "<i4"), ("last_valid_stop1", "<i4"),
                      ("conflict", "<i4", (16,)), ("memorized", "<i4", (16,)), ("conflict_pt", "<f8", (16, 2)),
                      ("memorized_pt", "<f8", (16, 2))])
assert ENV_DTYPE.itemsize == 672


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def update_ref_records(g):
    """Detector records that put the device where the reference agent stood before `_solve` in the update_ref_* cases
    (agents/pure_mpc.py:678-724: is_collide, conflict_index, collision_memory, memorized_conflict_indices,
    last_valid_stop_point set by hand; stop_point None)."""
    n = len(g["update_ref_is_collide"])
    rec = np.zeros(n, ENV_DTYPE)
    rec["collision_memory"] = g["update_ref_mem"]
    rec["has_memorized"] = g["update_ref_has_mem"]
    rec["n_memorized"] = g["update_ref_n_mem"]
    rec["n_conflict"] = g["update_ref_n_conf"]
    rec["is_collide"] = g["update_ref_is_collide"]
    rec["last_valid_stop1"] = g["update_ref_last_valid"] + 1
    rec["conflict"][:] = -1
    rec["memorized"][:] = -1
    rec["conflict"][:, :4] = g["update_ref_conflict"]
    rec["memorized"][:, :4] = g["update_ref_memorized"]
    return rec


def window(speed_col, ego_index, N):
    """vref of the solve: row min(ego_index + k, 84) of the rewritten table (agents/pure_mpc.py:129)."""
    idx = np.minimum(np.asarray(ego_index)[..., None] + np.arange(N + 1), speed_col.shape[-1] - 1)
    return np.take_along_axis(speed_col, idx, axis=-1)


def sequence_groups(g):
    """The environments of reference_sequences.npz split by what the C ABI takes per call: a reference-speed override for
    every environment of the call or for none."""
    rl = g["seq_ref_speed"]
    has = ~np.isnan(rl[0])
    assert np.array_equal(~np.isnan(rl), np.broadcast_to(has, rl.shape))
    return [np.nonzero(~has)[0], np.nonzero(has)[0]]


def sequence_weights(g, envs):
    w = g["seq_weights"][envs].copy()
    w[np.isnan(w[:, 0])] = 1.0            # cfg defaults weight_speed / control / input_diff = 1 (make_golden._REF_CFG)
    return w
