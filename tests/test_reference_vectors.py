"""Everything the REFERENCE's own Python statements produced in this container (tests/golden/make_golden.py round 4:
reference_random.npz, reference_sequences.npz, reference_distance_cost.npz; what stands in for the absent casadi / shapely
and what that leaves unpinned: tests/golden/standins.py) against
  * the numpy mirror of the preamble (tests/host_preamble.py),
  * the device preamble compiled for the host (csrc/mpc_preamble.hpp through tests/cpu_preamble_harness.cpp),
  * the oracle's restatement of the NLP (oracle/nlp_spec.py, oracle/nlp_batch.py).
CPU only; tests/test_reference_vectors_gpu.py replays the same fixtures through the C ABI on the GPU."""
import ctypes

import numpy as np
import pytest

import ref_fixtures as rf
from test_host import CFG, Env, FakeEngine
from test_preamble_cpu import DevicePreamble, load_pre


@pytest.fixture(scope="module")
def rnd():
    return rf.load("reference_random.npz")


@pytest.fixture(scope="module")
def seq():
    return rf.load("reference_sequences.npz")


@pytest.fixture(scope="module")
def pre():
    lib = load_pre()
    lib.preamble_batch_adv.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_double] + [ctypes.c_void_p] * 8 + [ctypes.c_int]
    lib.preamble_agent_path.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
    lib.preamble_agent_path.restype = None
    return lib


@pytest.fixture()
def mirror():
    from host_preamble import HostPreambleAgent
    return HostPreambleAgent(Env(), dict(CFG), engine=FakeEngine())


def test_fixture_sizes(rnd, seq):
    """>= 500 cases per function of the reference."""
    assert len(rnd["parse_ego"]) >= 500 and len(rnd["normalize_out"]) >= 500 and len(rnd["ego_future_len"]) >= 500
    assert len(rnd["agent_future_out"]) >= 500 and len(rnd["update_ref_speed_out"]) >= 500
    T, E = seq["seq_obs"].shape[:2]
    assert T * E >= 1500 and seq["seq_is_collide"].sum() > 300 and (seq["seq_is_collide"] == 0).sum() > 300
    assert seq["seq_reset"][1:].sum() >= 5                        # episode boundaries inside the sequences
    assert len(seq["nlp_f"]) >= 1000


# ---------------------------------------------------------------------------------------------------------------
# numpy mirror
# ---------------------------------------------------------------------------------------------------------------
def test_mirror_parse_and_normalize(mirror, rnd):
    for b, obs in enumerate(rnd["parse_obs"]):
        mirror._parse_obs(obs)
        e = mirror.ego_vehicle
        got = np.array([e.position[0], e.position[1], e.heading, e.speed, mirror.observed_vehicles_count], np.float64)
        assert np.array_equal(got, rnd["parse_ego"][b])
        oth = np.zeros((9, 4))
        for j, v in enumerate(mirror.agent_vehicles):
            oth[j] = (v.position[0], v.position[1], v.speed, v.heading)
        assert np.array_equal(oth, rnd["parse_others"][b])
    assert np.array_equal([mirror.normalize_angle(a) for a in rnd["normalize_in"]], rnd["normalize_out"])
    got32 = [float(mirror.normalize_angle(np.float32(a))) for a in rnd["normalize32_in"]]
    assert np.array_equal(got32, rnd["normalize32_out"])


def test_mirror_predictors(mirror, rnd):
    for obs, vref, want, n in zip(rnd["ego_future_obs"], rnd["ego_future_vref"], rnd["ego_future_out"],
                                  rnd["ego_future_len"]):
        mirror._parse_obs(obs)
        e = mirror.ego_vehicle
        fut = mirror.predict_ego_future_positions(e.position, e.speed, e.heading, e.max_acceleration, 0.1, 30, vref)
        assert len(fut) == n
        assert np.array_equal(np.asarray([np.asarray(q, np.float64) for q in fut]), want[:n])
    for obs, want in zip(rnd["agent_future_obs"], rnd["agent_future_out"]):
        mirror._parse_obs(obs)
        v = mirror.agent_vehicles[0]
        fut = mirror.predict_future_positions(np.array(v.position), v.speed, v.heading, 0.1, 30)
        assert np.array_equal(np.asarray(fut, np.float64), want)


def test_mirror_update_reference_states(mirror, rnd):
    from host_preamble import _EnvState
    tr = mirror.reference_trajectory
    for b in range(len(rnd["update_ref_is_collide"])):
        mirror._parse_obs(rnd["update_ref_obs"][b])
        st = _EnvState()
        nc, nm = int(rnd["update_ref_n_conf"][b]), int(rnd["update_ref_n_mem"][b])
        st.is_collide = bool(rnd["update_ref_is_collide"][b])
        st.conflict_index = [None if c < 0 else int(c) for c in rnd["update_ref_conflict"][b, :nc]]
        st.collision_memory = int(rnd["update_ref_mem"][b])
        st.memorized_conflict_indices = None if not rnd["update_ref_has_mem"][b] else \
            [None if c < 0 else int(c) for c in rnd["update_ref_memorized"][b, :nm]]
        lv = int(rnd["update_ref_last_valid"][b])
        st.last_valid_stop_point = None if lv < 0 else tr[lv]
        st.ego_index = mirror._nearest_ref_index(mirror.ego_vehicle.position)
        assert st.ego_index == rnd["update_ref_ego_index"][b]
        rl = rnd["update_ref_rl"][b]
        ref = mirror.update_reference_states(0, None if np.isnan(rl) else np.array([[rl]]), st, mirror.ego_vehicle.speed)
        assert np.array_equal(ref[:, 2], rnd["update_ref_speed_out"][b]), b
        stop = -1 if st.stop_point is None else int(np.argmin(np.linalg.norm(tr - st.stop_point, axis=1)))
        assert stop == rnd["update_ref_stop_out"][b], b


def _replay_sequences(seq, step, reset):
    """Feeds the recorded observation sequences to `step(t, envs, obs, ref_speed or None)` group by group and compares what
    it returns with what the reference's predict() computed; `reset(group, ids)` at episode boundaries."""
    N = 20
    T = seq["seq_obs"].shape[0]
    for gi, envs in enumerate(rf.sequence_groups(seq)):
        if envs.size == 0:
            continue
        for t in range(T):
            ids = np.nonzero(seq["seq_reset"][t, envs])[0]
            if t > 0 and ids.size:
                reset(gi, ids)
            rs = seq["seq_ref_speed"][t, envs]
            got = step(gi, t, envs, np.ascontiguousarray(seq["seq_obs"][t, envs]), None if np.isnan(rs[0]) else rs)
            want_vref = rf.window(seq["seq_speed_col"][t, envs], seq["seq_ego_index"][t, envs], N)
            tag = (gi, t)
            assert np.array_equal(got["state"], seq["seq_state"][t, envs]), tag
            assert np.array_equal(got["nveh"], seq["seq_nveh"][t, envs]), tag
            assert np.array_equal(got["ego_index"], seq["seq_ego_index"][t, envs]), tag
            assert np.array_equal(got["is_collide"], seq["seq_is_collide"][t, envs]), tag
            assert np.array_equal(got["collision_memory"], seq["seq_collision_memory"][t, envs]), tag
            assert np.array_equal(got["vref"], want_vref), tag
            assert np.array_equal(got["stop_index"], seq["seq_stop_index"][t, envs]), tag
            assert np.array_equal(got["conflict_index"][:, :9], seq["seq_conflict_index"][t, envs]), tag
            assert np.array_equal(got["conflict_points"][:, :9], seq["seq_conflict_points"][t, envs], equal_nan=True), tag
            for i in range(envs.size):
                nv = got["nveh"][i]
                assert np.array_equal(got["others"][i, :nv], seq["seq_others"][t, envs[i], :nv]), tag


def test_mirror_state_machine_on_reference_sequences(seq):
    """The mirror's `_check_collision` / `update_reference_states` against the reference's, step by step, over closed-loop
    episodes (both read the same geometry: the mirror's segment arithmetic is what serves LineString.intersection there)."""
    from host_preamble import HostPreambleAgent
    agents = {}

    def step(gi, t, envs, obs, rs):
        ag = agents.setdefault(gi, HostPreambleAgent(Env(), dict(CFG), engine=FakeEngine()))
        ag.predict_batch_host(obs, None, None if rs is None else rs[:, None])
        inp, B = ag.last_inputs, len(envs)
        out = dict(state=inp["state"], ego_index=inp["ego_index"], is_collide=inp["is_collide"], vref=inp["vref"],
                   nveh=np.array([int((o[:, 0] == 1).sum()) - 1 for o in obs], np.int32),
                   collision_memory=np.array([s.collision_memory for s in ag._states[:B]], np.int32),
                   conflict_index=np.full((B, 9), -1, np.int32), conflict_points=np.full((B, 9, 2), np.nan),
                   stop_index=np.full(B, -1, np.int32), others=np.zeros((B, 9, 4)))
        for i, s in enumerate(ag._states[:B]):
            for j, c in enumerate(s.conflict_index):
                out["conflict_index"][i, j] = -1 if c is None else c
            for j, c in enumerate(s.conflict_points):
                if c is not None:
                    out["conflict_points"][i, j] = c
            if s.stop_point is not None:
                out["stop_index"][i] = int(np.argmin(np.linalg.norm(ag.reference_trajectory - s.stop_point, axis=1)))
            e, others = ag._vehicles_from_obs(obs[i])
            for j, v in enumerate(others):
                out["others"][i, j] = (v.position[0], v.position[1], v.speed, v.heading)
        return out

    def reset(gi, ids):
        for i in ids:
            agents[gi]._states[int(i)] = type(agents[gi]._states[0])()
    _replay_sequences(seq, step, reset)


# ---------------------------------------------------------------------------------------------------------------
# the device preamble, compiled for the host
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wave", [False, True])
def test_device_code_parse_and_predictors(pre, rnd, ref_table, wave):
    dev = DevicePreamble(pre, ref_table, wave=wave)
    got = dev(rnd["parse_obs"])
    assert np.array_equal(got["state"], rnd["parse_ego"][:, :4])
    assert np.array_equal(got["nveh"], rnd["parse_ego"][:, 4].astype(np.int32))
    for b in range(len(got["nveh"])):
        nv = got["nveh"][b]
        assert np.array_equal(got["others"][b, :nv], rnd["parse_others"][b, :nv])
    ref = np.ascontiguousarray(ref_table)
    for obs, vref, want, n in zip(rnd["ego_future_obs"], rnd["ego_future_vref"], rnd["ego_future_out"],
                                  rnd["ego_future_len"]):
        sp = np.sqrt(obs[0, 3] * obs[0, 3] + obs[0, 4] * obs[0, 4])
        out = np.full((31, 2), np.nan)
        k = pre.preamble_ego_future(ref.ctypes.data, ref.shape[0], obs[0, 1], obs[0, 2], sp, float(vref), 0.1, out.ctypes.data)
        assert k == n and np.array_equal(out[:k], want[:k])
    check_agent_paths(rnd, lambda b, row: _host_agent_path(pre, row))
    if wave:
        # the wave form's own polylines (what mpc_get_last_paths exports on the GPU)
        vref = rnd["ego_future_vref"]
        for v in np.unique(vref):
            sel = np.nonzero(vref == v)[0]
            tab = ref_table.copy()
            tab[:, 2] = v
            d2 = DevicePreamble(pre, tab, wave=True)
            d2(rnd["ego_future_obs"][sel])
            assert np.array_equal(d2.paths["ego_len"], rnd["ego_future_len"][sel]), v
            for i, b in enumerate(sel):
                k = d2.paths["ego_len"][i]
                assert np.array_equal(d2.paths["ego_path"][i, :k], rnd["ego_future_out"][b, :k]), (v, b)
        d3 = DevicePreamble(pre, ref_table, wave=True)
        d3(rnd["agent_future_obs"])
        check_agent_paths(rnd, lambda b, row: d3.paths["agent_paths"][b, 0])


def _host_agent_path(pre, row):
    row = np.ascontiguousarray(row)
    out = np.zeros((31, 2), np.float32)
    pre.preamble_agent_path(row.ctypes.data, 0.1, out.ctypes.data)
    return out


def check_agent_paths(rnd, path_of):
    """predict_future_positions (agents/pure_mpc.py:529-550) is float32 arithmetic on the float32 observation, INCLUDING
    np.cos / np.sin of the float32 heading - numpy's SIMD float32 kernels, which are not correctly rounded (and whose
    dispatch depends on the CPU the reference runs on).  The device rounds the double-precision value to float32.  On the
    lane axes (what the intersection's straight lanes give) the two agree and the paths are bit-identical; elsewhere the
    step vector may differ in its last float32 bit: <= 3e-6 m over the 30 steps.  Documented deviation, bounded here."""
    axes = np.array([0.0, np.pi / 2, np.pi, -np.pi / 2], np.float32)
    exact = off = 0
    for b, (obs, want) in enumerate(zip(rnd["agent_future_obs"], rnd["agent_future_out"])):
        got = np.asarray(path_of(b, obs[1]), np.float64)
        if obs[1, 5] in axes:
            assert np.array_equal(got, want), b
            exact += 1
        else:
            assert np.abs(got - want).max() <= 3e-6, b
            off += 1
    assert exact >= 200 and off >= 200


def test_device_code_update_reference_states(pre, rnd, ref_table):
    """The device's speed-profile rewrite from detector records set like the reference agent's attributes
    (MPC_FLAG_DETECTED path: the records are not advanced)."""
    n = len(rnd["update_ref_is_collide"])
    rec = rf.update_ref_records(rnd)
    env = np.ascontiguousarray(rec.view(np.int32).reshape(n, -1))
    obs = np.ascontiguousarray(rnd["update_ref_obs"])
    rl = rnd["update_ref_rl"]
    ref = np.ascontiguousarray(ref_table)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for has_rl in (False, True):
        sel = np.nonzero(~np.isnan(rl) if has_rl else np.isnan(rl))[0]
        B = sel.size
        o = np.ascontiguousarray(obs[sel])
        e = np.ascontiguousarray(env[sel])
        rs = np.ascontiguousarray(rl[sel]) if has_rl else None
        out = dict(state=np.zeros((B, 4)), ego=np.zeros(B, np.int32), vref=np.zeros((B, 21)), col=np.zeros(B, np.uint8),
                   oth=np.zeros((B, 9, 4)), nveh=np.zeros(B, np.int32))
        rc = pre.preamble_batch_adv(B, p(o), 10, p(ref), 85, 20, 0.1, None if rs is None else p(rs), p(e), p(out["state"]),
                                    p(out["ego"]), p(out["vref"]), p(out["col"]), p(out["oth"]), p(out["nveh"]), 0)
        assert rc == 0
        assert np.array_equal(out["ego"], rnd["update_ref_ego_index"][sel])
        assert np.array_equal(out["vref"], rf.window(rnd["update_ref_speed_out"][sel], out["ego"], 20))
        after = e.view(rf.ENV_DTYPE).reshape(B)
        assert np.array_equal(after["stop_index1"] - 1, rnd["update_ref_stop_out"][sel])
        assert np.array_equal(after["last_valid_stop1"] - 1, rnd["update_ref_last_valid_out"][sel])
        assert np.array_equal(out["col"], rnd["update_ref_is_collide"][sel])


@pytest.mark.parametrize("wave", [False, True])
def test_device_code_state_machine_on_reference_sequences(pre, seq, ref_table, wave):
    devs = {}

    def step(gi, t, envs, obs, rs):
        dev = devs.setdefault(gi, DevicePreamble(pre, ref_table, wave=wave))
        o = dev(obs, rs)
        B = len(envs)
        rec = dev.env[:B].view(rf.ENV_DTYPE).reshape(B)
        nc = rec["n_conflict"]
        ci = np.where(np.arange(16)[None, :] < nc[:, None], rec["conflict"], -1)
        cp = np.where((ci >= 0)[:, :, None], rec["conflict_pt"], np.nan)
        return dict(state=o["state"], nveh=o["nveh"], ego_index=o["ego_index"], is_collide=o["is_collide"], vref=o["vref"],
                    collision_memory=rec["collision_memory"].copy(), conflict_index=ci, conflict_points=cp,
                    stop_index=rec["stop_index1"] - 1, others=o["others"])

    def reset(gi, ids):
        devs[gi].env[ids] = 0
    _replay_sequences(seq, step, reset)


# ---------------------------------------------------------------------------------------------------------------
# the NLP itself: the reference's own objective / constraint / bound / initial-guess statements, evaluated numerically
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_nlp_equals_the_references_statements(seq, ref_table):
    """agents/pure_mpc.py:128-283 executed with a numeric stand-in for casadi (values in, values out): f(z), g(z) at the
    cold start, at the oracle's solution and at random points of 400 closed-loop steps (default and RL weights incl.
    negative ones, with and without a predicted collision, with the RL speed override), lbx / ubx / lbg / ubg / x0, and
    the solver options the reference passes."""
    import nlp_batch as nb
    import nlp_spec as ns
    assert int(seq["seq_ipopt_max_iter"]) == 1000 and float(seq["seq_ipopt_tol"]) == 1e-6       # agents/pure_mpc.py:294-295
    t, e = seq["nlp_t"], seq["nlp_e"]
    n = len(t)
    w = seq["seq_weights"][e].copy()
    w[np.isnan(w[:, 0])] = 1.0
    tab = np.broadcast_to(ref_table, (n, 85, 4)).copy()
    tab[:, :, 2] = seq["seq_speed_col"][t, e]
    ego = seq["seq_ego_index"][t, e]
    col = seq["seq_is_collide"][t, e]
    state = seq["seq_state"][t, e]
    assert (w < 0).any() and col.any() and (~col.astype(bool)).any()
    worst_f = worst_g = 0.0
    for i in range(n):
        p = ns.Problem.build(20, 0.1, state[i], int(ego[i]), tab[i], w[i], bool(col[i]))
        X, U = ns.unpack(p, seq["nlp_z"][i])
        f, g = ns.cost(p, X, U), ns.constraints(p, X, U).ravel()
        worst_f = max(worst_f, abs(f - seq["nlp_f"][i]) / max(1.0, abs(f)))
        worst_g = max(worst_g, float(np.abs(g - seq["nlp_g"][i]).max()))
    assert worst_f <= 1e-14 and worst_g == 0.0, (worst_f, worst_g)
    # bounds, cold start, equality right-hand sides
    p = ns.Problem.build(20, 0.1, seq["seq_state"][0, 0], int(seq["seq_ego_index"][0, 0]), ref_table, np.ones(3), False)
    lo, hi = ns.bounds(p)
    assert np.array_equal(lo, seq["seq_lbx"]) and np.array_equal(hi, seq["seq_ubx"])
    assert not seq["seq_lbg"].any() and not seq["seq_ubg"].any() and len(seq["seq_lbg"]) == 84
    X0, U0 = ns.initial_guess(p)
    assert np.array_equal(ns.pack(X0, U0), seq["seq_x0"])
    # the batched restatement (what certifies the GPU's solutions) on the same points
    vref = rf.window(seq["seq_speed_col"][t, e], ego, 20)
    pb = nb.Batch.build(ref_table, state, ego, w, col, vref=vref)
    Xb = seq["nlp_z"][:, :84].reshape(n, 21, 4)
    Ub = seq["nlp_z"][:, 84:].reshape(n, 20, 2)
    fb = nb.cost(pb, Xb, Ub)
    cb = nb.constraints(pb, Xb, Ub).reshape(n, -1)
    assert np.max(np.abs(fb - seq["nlp_f"]) / np.maximum(1.0, np.abs(fb))) <= 1e-13
    assert np.max(np.abs(cb - seq["nlp_g"])) <= 1e-12


def nlp_points(seq, ref_table):
    """The 1200 points of reference_sequences.npz as problem data + (X, U) + the reference's f, g."""
    t, e = seq["nlp_t"], seq["nlp_e"]
    w = seq["seq_weights"][e].copy()
    w[np.isnan(w[:, 0])] = 1.0
    ego = seq["seq_ego_index"][t, e]
    n = len(t)
    return dict(ego=ego, weights=w, collide=seq["seq_is_collide"][t, e], vref=rf.window(seq["seq_speed_col"][t, e], ego, 20),
                state=seq["seq_state"][t, e], X=seq["nlp_z"][:, :84].reshape(n, 21, 4), U=seq["nlp_z"][:, 84:].reshape(n, 20, 2),
                f=seq["nlp_f"], g=seq["nlp_g"].reshape(n, 21, 4))


def check_nlp_functions(evaluate, seq, ref_table):
    """`evaluate(ego, weights, collide, X, U, vref=...) -> (f, x_next)` of the kernel source against the reference's own
    f(z) and g(z): the objective to 1e-13 relative (the sum over the stages is a DPP reduction tree on the device, a loop in
    casadi), the dynamics rows X[k+1] - x_next[k] to 5e-14 (lean sine / cosine: 1.3e-15 absolute, times speed x dt; one
    rounding of a position of ~50 m)."""
    p = nlp_points(seq, ref_table)
    f, xn = evaluate(p["ego"], p["weights"], p["collide"], p["X"], p["U"], vref=p["vref"])
    assert (p["weights"] < 0).any() and p["collide"].any() and len(f) >= 1200
    rel = np.abs(f - p["f"]) / np.maximum(1.0, np.abs(p["f"]))
    assert rel.max() <= 1e-13, rel.max()
    g = p["X"][:, 1:] - xn
    assert np.abs(g - p["g"][:, 1:]).max() <= 5e-14, np.abs(g - p["g"][:, 1:]).max()
    assert np.array_equal(p["X"][:, 0] - p["state"], p["g"][:, 0])          # the row X[0] - state, for completeness
    return float(rel.max()), float(np.abs(g - p["g"][:, 1:]).max())


def test_device_code_nlp_functions_equal_the_references_statements(seq, ref_table):
    """a7 / a9 on the kernel SOURCE (Solver::evaluate of csrc/mpc_wave.hpp = what mpc_eval_nlp runs; here compiled for the
    host): stage_terms / track / dyn_eval - the functions every line-search trial and every rollout of the solver goes
    through - against f(z), g(z) computed by the reference's statements."""
    from conftest import host_eval_nlp
    check_nlp_functions(lambda ego, w, c, X, U, vref: host_eval_nlp(ref_table, ego, w, c, X, U, vref=vref), seq, ref_table)


def distance_cost_points():
    d = rf.load("reference_distance_cost.npz")
    n = len(d["distance_cost"])
    others = np.zeros((n, 9, 4))
    nveh = np.zeros(n, np.int32)
    for b in range(n):
        o = d["obs"][b]
        nv = int((o[:, 0] == 1).sum()) - 1
        nveh[b] = nv
        for j in range(nv):
            r = o[j + 1]
            others[b, j] = (r[1], r[2], np.sqrt(r[3] * r[3] + r[4] * r[4]), r[5])
        others[b, nv:] = (1e6, 1e6, 0.0, 0.0)       # absent vehicles far away: 100 / d^2 = 5e-11, below the tolerance
    return d, others, nveh


def check_distance_cost(evaluate, ref_table):
    """a8 on the kernel source: f with the collision cost minus f without it = w_distance x the archived agent's distance
    term (agents/archive/pure_mpc.py:189-206, executed), within deviation (iv) of DESIGN.md section 3.1 (the reference walks
    the vehicles in float32): 2e-4 relative with a node inside d < 1 m, 5e-5 elsewhere - and equal to the oracle's
    restatement of the same term to rounding."""
    import nlp_spec as ns
    d, others, nveh = distance_cost_points()
    n = len(nveh)
    X = d["z"][:, :84].reshape(n, 21, 4)
    U = d["z"][:, 84:].reshape(n, 20, 2)
    ego, w, col = np.zeros(n, np.int32), np.ones((n, 3)), np.zeros(n, np.uint8)
    f1, _ = evaluate(ego, w, col, X, U, others=others, collision_cost=True)
    f0, _ = evaluate(ego, w, col, X, U, others=None, collision_cost=False)
    got = f1 - f0
    want = d["distance_cost"]
    dd = np.linalg.norm(X[:, :20, None, :2] - (others[:, None, :, :2] + np.arange(20)[None, :, None, None] * 0.1 *
                                              others[:, None, :, 2:3] * np.stack([np.cos(others[..., 3]), np.sin(others[..., 3])], -1)[:, None]), axis=-1)
    near = (dd < 1.0).any(axis=(1, 2))
    spec = np.sum(np.where(dd < 1.0, 1000.0, 100.0) / (dd + 1e-6) ** 2, axis=(1, 2))
    rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    assert near.sum() >= 100
    assert rel[near].max() <= 2e-4 and rel[~near].max() <= 5e-5, (rel[near].max(), rel[~near].max())
    # against the oracle's restatement of the same term: the difference of two objectives of size |f| carries |f| x 1e-16
    assert (np.abs(got - spec) / np.maximum(1.0, np.maximum(np.abs(spec), np.abs(f1)))).max() <= 1e-13


def test_device_code_distance_cost_equals_the_archived_agents(ref_table):
    from conftest import host_eval_nlp
    check_distance_cost(lambda ego, w, c, X, U, others, collision_cost: host_eval_nlp(
        ref_table, ego, w, c, X, U, others=others, collision_cost=collision_cost), ref_table)


def test_oracle_distance_cost_equals_the_archived_agents(ref_table):
    """the distance term of agents/archive/pure_mpc.py:189-206, evaluated by the archived agent's own
    statements.  The reference advances the other vehicles in float32 (their observation's dtype, base_agent.py:172-174);
    oracle and engine place vehicle j at stage k at p_j + k speed_j dt (cos h_j, sin h_j) in double: positions within
    2e-5 m, the cost within 2e-4 relative where a node sits inside d < 1 m (1000 / d^2 is steep there) and 5e-5 elsewhere
    - a documented deviation, bounded here."""
    import nlp_spec as ns
    d = rf.load("reference_distance_cost.npz")
    assert len(d["distance_cost"]) >= 500
    rel = np.zeros(len(d["distance_cost"]))
    near = np.zeros(len(rel), bool)
    for b in range(len(rel)):
        o = d["obs"][b]
        nv = int((o[:, 0] == 1).sum()) - 1
        oth = np.zeros((nv, 4))
        for j in range(nv):
            r = o[j + 1]
            oth[j] = (r[1], r[2], np.sqrt(r[3] * r[3] + r[4] * r[4]), r[5])
        p = ns.Problem.build(20, 0.1, np.zeros(4), 0, ref_table, np.ones(3), False, collision_cost=True, others=oth)
        X, _ = ns.unpack(p, d["z"][b])
        J = 0.0
        for k in range(20):
            dd = np.linalg.norm(X[k, :2][None, :] - p.other_positions(k), axis=1)
            J += np.sum(np.where(dd < 1.0, 1000.0, 100.0) / (dd + 1e-6) ** 2)
            near[b] |= bool((dd < 1.0).any())
        rel[b] = abs(J - d["distance_cost"][b]) / max(1.0, abs(J))
    assert near.sum() >= 100
    assert rel[near].max() <= 2e-4 and rel[~near].max() <= 5e-5, (rel[near].max(), rel[~near].max())
    # state, control and input-difference components of the archived agent are the live agent's statements with other
    # constants; its collision component is commented out there (:204-208) and reads 0
    assert not d["components"][:, 5].any()


# ---------------------------------------------------------------------------------------------------------------
# iterative-linear agent (agents/pure_mpc_linear.py)
# ---------------------------------------------------------------------------------------------------------------
def ltv_reference_slacks(u, x, dt=0.1):
    """The inequality slacks in the order the reference appends its constraints (agents/pure_mpc_linear.py:232-255)."""
    import ltv_oracle as L
    T = u.shape[0]
    s = [L.MAX_DSTEER * dt - abs(u[t + 1, 1] - u[t, 1]) for t in range(T - 1)]
    for t in range(T):
        s += [L.MAX_ACCEL - u[t, 0], u[t, 0] - L.MAX_DECEL, L.MAX_STEER - abs(u[t, 1])]
    for t in range(T + 1):
        s += [x[t, 2], L.MAX_SPEED - x[t, 2]]
    return np.array(s)


def test_ltv_oracle_equals_the_references_helpers_and_qp(ref_table):
    import ltv_oracle as L
    d = rf.load("ltv_reference_random.npz")
    assert min(len(d["nearest_out"]), len(d["linmodel_A"]), len(d["nominal_xbar"])) >= 500 and len(d["qp_cost"]) >= 250
    assert np.array_equal(L.nearest_index(d["nearest_in"][:, 0], d["nearest_in"][:, 1], ref_table), d["nearest_out"])
    A, B = L.linear_model(d["linmodel_in"][:, 0], d["linmodel_in"][:, 1], 0.1)
    assert np.array_equal(A, d["linmodel_A"]) and np.array_equal(B, d["linmodel_B"])
    xbar = L.nominal_rollout(d["nominal_x0"], d["nominal_oa"], d["nominal_od"], 0.1)
    assert np.abs(xbar - d["nominal_xbar"]).max() <= 1e-12          # math.cos / np.cos: last-bit differences at most
    ns = len(d["stage_A"])
    A, B = L.linear_model(d["nominal_xbar"][:ns, :20, 2], d["nominal_xbar"][:ns, :20, 3], 0.1)
    assert np.array_equal(A, d["stage_A"]) and np.array_equal(B, d["stage_B"])
    # the QP: objective and constraints of _linear_mpc_control, evaluated by the reference's own cvxpy statements
    n_on = 0
    for i in range(len(d["qp_cost"])):
        o = d["qp_obs"][i, 0]
        x0 = np.array([o[1], o[2], np.sqrt(o[3] * o[3] + o[4] * o[4]), o[5]], dtype=np.float64)   # make_obs_batch: |heading| <= pi
        assert np.array_equal(x0, d["qp_x"][i, 0])
        tgt = L.nearest_index(x0[:1], x0[1:2], ref_table)
        assert tgt[0] == d["qp_target"][i]
        assert np.array_equal(L.reference_window(ref_table, tgt, 20)[0], d["qp_xref"][i])
        xb = L.nominal_rollout(x0[None], d["qp_oa"][i][None], d["qp_od"][i][None], 0.1)[0]
        u, x = d["qp_u"][i], d["qp_x"][i]
        on = np.abs(d["qp_eq"][i]).max() <= 1e-9                    # the point satisfies the reference's equalities
        if on:
            n_on += 1
            assert np.abs(L.simulate_linear(x0, u, xb, 0.1) - x).max() <= 1e-9
            f = L.objective_loops(u, x0, d["qp_xref"][i], xb, 0.1)
            assert abs(f - d["qp_cost"][i]) <= 1e-10 * max(1.0, abs(f))
            want = ltv_reference_slacks(u, L.simulate_linear(x0, u, xb, 0.1))
            assert np.abs(want - d["qp_le"][i]).max() <= 1e-9
            c = L.constraint_loops(u, x0, xb, 0.1)                  # the oracle's own row order: same feasible set
            assert (c.min() >= 0) == (d["qp_le"][i].min() >= 0) or abs(c.min()) < 1e-9
        else:
            assert np.abs(ltv_reference_slacks(u, x) - d["qp_le"][i]).max() <= 1e-12
    assert n_on >= 150


# ---------------------------------------------------------------------------------------------------------------
# predict() end to end, incl. the statements behind the solver call (agents/pure_mpc.py:300-318)
# ---------------------------------------------------------------------------------------------------------------
class _StubDevice:
    """Stands where MPCEngine stands in the PRODUCT agent (mpc-rl_for_avs_amd/pure_mpc.py): hands back a given solution and
    status like mpc_predict_batch would, so that what the agent does with them can be compared with what the reference's own
    statements did with the same solution."""

    def __init__(self):
        self.next = None

    def predict_batch(self, obs, w, rs, collision_cost=False, warm_start=False, detected=False):
        u0, ok = self.next
        return dict(act=np.asarray(u0, np.float64).reshape(1, 2).copy(), status=np.array([0 if ok else 1], np.int32),
                    iters=np.array([7], np.int32))

    def env_state(self, B):
        return dict(is_collide=np.zeros(B, np.int32), ego_index=np.zeros(B, np.int32), collision_memory=np.zeros(B, np.int32),
                    stop_index=np.full(B, -1, np.int32), conflict_index=np.full((B, 16), -1, np.int32),
                    conflict_points=np.full((B, 16, 2), np.nan))


def test_product_agent_does_with_a_solution_what_the_reference_does(capsys):
    """the reference's predict() was executed END TO END with the solver stand-in handing back a given
    solution (the oracle's of that step, flagged found / not found alternately): first control as the action, `last_acc`,
    the NOTICE of a failed solve whose last iterate is used all the same (agents/pure_mpc.py:300-318).  The product agent,
    given the same solution by a stub in the engine's place, returns the same array, keeps the same `last_acc`, prints the
    notice on the same steps."""
    from mpc_rl_for_avs_amd.pure_mpc import PureMPC_Agent
    d = rf.load("reference_predict_tail.npz")
    T, E = d["success"].shape
    assert T * E >= 300 and (~d["success"]).sum() >= 50
    # the fixture itself: what the reference returned IS the first control of the solution it was handed
    assert np.array_equal(d["action"], d["U"][:, :, 0, :]) and np.array_equal(d["last_acc"], d["U"][:, :, 0, 0])
    assert np.array_equal(d["notice"], ~d["success"])
    stub = _StubDevice()
    ag = PureMPC_Agent(Env(), dict(CFG), engine=stub)
    for t in range(T):
        for e in range(E):
            stub.next = (d["U"][t, e, 0], bool(d["success"][t, e]))
            rs = None if np.isnan(d["ref_speed"][t, e]) else np.array([[d["ref_speed"][t, e]]])
            a = ag.predict(d["obs"][t, e], ref_speed=rs)
            out = capsys.readouterr().out
            assert isinstance(a, np.ndarray) and a.shape == (2,) and np.array_equal(a, d["action"][t, e])
            assert ag.last_acc == d["last_acc"][t, e]
            assert ("NOTICE: Not found solution" in out) == bool(d["notice"][t, e])
            m = ag.predict(d["obs"][t, e], return_numpy=False, ref_speed=rs)
            capsys.readouterr()
            assert (m.acceleration, m.steer) == (a[0], a[1]) and m.success == bool(d["success"][t, e])
