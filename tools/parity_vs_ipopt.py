#!/usr/bin/env python3
"""How often does the engine's algorithm return a different ACTION than the reference's solver would?  (CPU only.)

CasADi/IPOPT cannot run here, so the proxy is oracle/ipopt_restated.py (IPOPT's published algorithm on the reference's
full-space transcription) at the reference's settings (tol 1e-6, max_iter 1000: agents/pure_mpc.py:294-295); the engine's
algorithm is represented by oracle/mpc_oracle.c (GPU == oracle to 1e-9 is what the -m gpu tests establish; the GPU test
tests/test_parity_gpu.py::test_closed_loop_fixtures_vs_independent_solver repeats part (a) on the device).

(a) tests/golden/closed_loop_ipopt.npz (generator: tests/golden/make_closed_loop.py): problem data recorded from closed-loop
    runs.  Agreement rate of u0 per scenario, and for EVERY disagreement: objective values of both answers, KKT certificates
    of both (oracle/kkt_batch.py), whether each solver stays at the other's answer when started there (both are then local
    minimisers and the cold start decides the basin), and geometric class.
(b) the instances of BASELINE config 3 that the engine ends with status 5 (a vehicle held on the d = 1 discontinuity of the
    collision cost): what the proxy does there.
(c) the synthetic fixtures of round 2 (tests/golden/independent_solutions.npz) re-classified the same way.

Output: profiles/rNN_parity_vs_ipopt.txt (stdout).
"""
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "oracle")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")

import numpy as np  # noqa: E402
import mpc_rl_for_avs_amd  # noqa: E402,F401
from mpc_rl_for_avs_amd import synth  # noqa: E402
from mpc_rl_for_avs_amd.reference_path import reference_states  # noqa: E402
import nlp_batch as nb  # noqa: E402
import kkt_batch as kb  # noqa: E402
import ipopt_restated as ipr  # noqa: E402
import oracle_lib  # noqa: E402

REF = reference_states(0.1)
M, N = REF.shape[0], 20
conv = lambda st: (st == 0) | ((st >= 5) & (st <= 7))      # the engine / oracle: MPC_STATUS_IS_SOLVED
ipok = lambda st: (st == 0) | (st == 3)                     # the proxy: converged, or IPOPT's "solved to acceptable level"


def rel(a, b):
    return np.abs(a - b).max(axis=-1) / np.maximum(1.0, np.abs(b).max(axis=-1))


def _ip_from(args):
    """ipopt_restated started at a given primal point (X, U)."""
    d, cc, b, X, U = args
    p = nb.Batch.build(REF, d["state"][b:b + 1], d["ego_index"][b:b + 1], d["weights"][b:b + 1], d["is_collide"][b:b + 1],
                       vref=d["vref"][b:b + 1], others=d["others"][b:b + 1], collision_cost=cc)
    z0 = None if X is None else nb.pack(X[None], U[None])[0]
    r = ipr.solve(p, tol=1e-6, max_iter=1000, sf_min=1e-2, z_init=z0)
    return r["U"], r["X"], int(np.ravel(r["status"])[0]), int(np.ravel(r["iters"])[0])


def classify(d, b):
    tags = []
    if d["ego_index"][b] + N > M - 1:
        tags.append(f"reference window runs past the end of the 85-point table ({min(N, d['ego_index'][b] + N - (M - 1))} clamped stages)")
    if d["is_collide"][b]:
        tags.append("is_collide (stop profile, w_s = 100)")
    dth = abs(d["state"][b, 2] - REF[d["ego_index"][b], 3])
    if dth > 0.04:
        tags.append(f"heading {dth:.3f} rad off the path at {d['state'][b, 3]:.1f} m/s (steering almost free: 0.01 w_c delta^2)")
    if abs(abs(d["state"][b, 2]) - np.pi) < 1e-6:
        tags.append("theta_0 on the heading bound")
    return "; ".join(tags) if tags else "none of the listed features"


def study(name, d, cc, ip, pool, out):
    """d: problem data; ip: dict(u0, U, X, status) of the independent solver."""
    B = d["state"].shape[0]
    orc = oracle_lib.solve_batch(REF, d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"],
                                 others=d["others"], collision_cost=cc, max_iter=1000, xy_bounds=False, nthreads=8)
    p = nb.Batch.build(REF, d["state"], d["ego_index"], d["weights"], d["is_collide"], vref=d["vref"], others=d["others"],
                       collision_cost=cc)
    both = ipok(ip["status"]) & conv(orc["status"])
    err = rel(orc["u0"], ip["u0"])
    agree = both & (err <= 1e-4)
    bad = np.nonzero(both & (err > 1e-4))[0]
    pfail = ~ipok(ip["status"])                     # IPOPT's failure exits: the reference acts on the last iterate
    efail = ipok(ip["status"]) & ~conv(orc["status"])
    n_resto = int((ip["n_resto"] > 0).sum()) if "n_resto" in ip else -1
    out(f"{name}: {B} instances; independent solver converged {int((ip['status'] == 0).sum())} ({n_resto} of all went through "
        f"its restoration phase; failure exits: iteration limit {int((ip['status'] == 1).sum())}, restoration failed "
        f"{int((ip['status'] == 5).sum())}, restoration ended at a point of local infeasibility {int((ip['status'] == 6).sum())}); "
        f"engine algorithm converged {int(conv(orc['status']).sum())}; both {int(both.sum())}; u0 within 1e-4: {int(agree.sum())} = "
        f"{agree.sum() / max(both.sum(), 1):.4f}; median rel. error of the agreeing {np.median(err[agree]):.1e}")
    cls = dict(n=B, agree=int(agree.sum()), two_minima=0, unexplained=0, proxy_fails=int(pfail.sum()), engine_fails=int(efail.sum()))
    if pfail.any():
        idx = np.nonzero(pfail)[0]
        e_conv = conv(orc["status"][idx])
        th0 = np.abs(np.abs(d["state"][idx, 2]) - np.pi) < 2e-7
        out(f"    proxy fails on {idx.size}: engine converged on {int(e_conv.sum())} of them; theta_0 within 2e-7 of the heading "
            f"bound (outside the relaxed bound by float32 rounding) in {int(th0.sum())}; |u0(engine) - u0(proxy's last iterate)| "
            f"median {np.median(err[idx]):.2e}, within 1e-4 in {int((err[idx] <= 1e-4).sum())}")
    if efail.any():
        idx = np.nonzero(efail)[0]
        # the engine's algorithm ran at tol 1e-8 above; the proxy (like the reference) stops at 1e-6: the same instances at
        # the reference's tolerance (PureMPC_Agent(reference_settings=True))
        o6 = oracle_lib.solve_batch(REF, d["state"][idx], d["ego_index"][idx], d["weights"][idx], d["is_collide"][idx],
                                    vref=d["vref"][idx], others=d["others"][idx], collision_cost=cc, max_iter=1000,
                                    xy_bounds=False, nthreads=8, tol=1e-6)
        e6 = rel(o6["u0"], ip["u0"][idx])
        cls["engine_fails_at_reference_tol"] = int((~conv(o6["status"])).sum())
        out(f"    engine (tol 1e-8) fails where the proxy (tol 1e-6) converges on {idx.size}: engine statuses "
            f"{orc['status'][idx].tolist()}, iterations {orc['iters'][idx].tolist()}, proxy iterations "
            f"{np.asarray(ip.get('iters', np.zeros(B, int)))[idx].tolist()}.  The engine's algorithm at the reference's tol 1e-6: "
            f"statuses {o6['status'].tolist()}, iterations {o6['iters'].tolist()}, |du0| to the proxy {[float(f'{v:.1e}') for v in e6]}")
    if not bad.size:
        return dict(both=int(both.sum()), agree=int(agree.sum()), lower=0, higher=0, both_fixed=0, cls=cls)
    Ji, Jo = nb.cost(p, ip["X"], ip["U"]), nb.cost(p, orc["X"], orc["U"])
    ci = kb.certify(p.take(bad), ip["X"][bad], ip["U"][bad])
    co = kb.certify(p.take(bad), orc["X"][bad], orc["U"][bad])
    # the proxy stops at tol 1e-6 (scaled): its point is a KKT point to that accuracy, the engine's to 1e-8
    certified = (co["stationarity"] <= 1e-6) & (ci["stationarity"] <= 1e-3) & (co["feasibility"] <= 1e-8) & (ci["feasibility"] <= 1e-5)
    cls["two_minima"] = int(certified.sum())
    cls["unexplained"] = int((~certified).sum())
    # each solver started at the other's answer
    warm = oracle_lib.solve_batch(REF, d["state"][bad], d["ego_index"][bad], d["weights"][bad], d["is_collide"][bad],
                                  vref=d["vref"][bad], others=d["others"][bad], collision_cost=cc, max_iter=1000,
                                  xy_bounds=False, nthreads=8, u_init=ip["U"][bad])
    back = pool.map(_ip_from, [(d, cc, int(b), orc["X"][b], orc["U"][b]) for b in bad])
    lower = higher = fixed = 0
    for i, b in enumerate(bad):
        stay_o = conv(warm["status"][i:i + 1])[0] and rel(warm["u0"][i], ip["u0"][b]) <= 1e-4
        stay_i = back[i][2] == 0 and rel(back[i][0][0], orc["u0"][b]) <= 1e-4
        fixed += bool(stay_o and stay_i)
        dj = (Jo[b] - Ji[b]) / max(1.0, abs(Ji[b]))
        lower += dj < -1e-9
        higher += dj > 1e-9
        out(f"    #{b:3d} |du0| {err[b]:.2e}  u0 engine ({orc['u0'][b, 0]:+.4f}, {orc['u0'][b, 1]:+.4f}) proxy "
            f"({ip['u0'][b, 0]:+.4f}, {ip['u0'][b, 1]:+.4f})  J engine {Jo[b]:.6g} proxy {Ji[b]:.6g} ({'engine lower' if dj < -1e-9 else ('proxy lower' if dj > 1e-9 else 'equal')})"
            f"  KKT stationarity engine {co['stationarity'][i]:.1e} proxy {ci['stationarity'][i]:.1e}"
            f"  engine started at the proxy's answer {'stays' if stay_o else 'leaves'}, proxy started at the engine's "
            f"{'stays' if stay_i else 'leaves'}  [{classify(d, b)}]")
    out(f"    => {bad.size} disagreements: engine's objective lower in {lower}, proxy's lower in {higher}; in {fixed} both "
        f"answers are fixed points of both solvers (two local minimisers, the cold start decides)")
    return dict(both=int(both.sum()), agree=int(agree.sum()), lower=int(lower), higher=int(higher), both_fixed=int(fixed), cls=cls)


def main():
    lines = []

    def out(s):
        print(s, flush=True)
        lines.append(s)
    out("# action-level parity against the independent solver (oracle/ipopt_restated.py at the reference's settings: tol 1e-6, "
        "max_iter 1000)")
    out("# engine algorithm = oracle/mpc_oracle.c at tol 1e-8 (the GPU kernel matches it to 1e-9, tests/test_parity_gpu.py)")
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        out("\n## (a) closed-loop fixtures, tests/golden/closed_loop_ipopt.npz")
        g = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop_ipopt.npz"))
        tot = dict(both=0, agree=0, lower=0, higher=0, both_fixed=0)
        table = []
        for name in ("c1", "c1cc", "c4", "c4mpc", "c4cc", "c4v1"):
            d = {k: g[f"{name}_{k}"] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
            ip = {k: g[f"{name}_{k}"] for k in ("u0", "U", "X", "status", "iters", "n_resto")}
            r = study(name, d, name.endswith("cc"), ip, pool, out)
            table.append((name, r["cls"]))
            if name == "c4v1":
                neg = (d["weights"] < 0).any(axis=1)
                out(f"    (v1 input domain: {int(neg.sum())} of {len(neg)} instances have a negative cost weight, "
                    f"{int((d['weights'] < 0).all(axis=1).sum())} all three)")
                continue                                     # non-convex by construction: kept out of the convex-domain total
            for k in tot:
                tot[k] += r[k]
        out(f"closed loop, scenarios with non-negative weights: {tot['agree']} of {tot['both']} actions within 1e-4 = "
            f"{tot['agree'] / tot['both']:.4f}; of the {tot['both'] - tot['agree']} others the engine's objective is lower in "
            f"{tot['lower']}, the proxy's in {tot['higher']}; {tot['both_fixed']} are pairs of local minimisers confirmed by both solvers")
        out("\n### every fixture instance in exactly one class (none dropped)")
        out("scenario | instances | agree (<= 1e-4) | two certified minima | unexplained | proxy fails (IPOPT failure exit) | engine fails, proxy converges")
        for name, c in table:
            assert c["agree"] + c["two_minima"] + c["unexplained"] + c["proxy_fails"] + c["engine_fails"] == c["n"], (name, c)
            out(f"{name} | {c['n']} | {c['agree']} | {c['two_minima']} | {c['unexplained']} | {c['proxy_fails']} | {c['engine_fails']}")
        s_ = {k: sum(c[k] for _, c in table) for k in ("n", "agree", "two_minima", "unexplained", "proxy_fails", "engine_fails")}
        out(f"all | {s_['n']} | {s_['agree']} | {s_['two_minima']} | {s_['unexplained']} | {s_['proxy_fails']} | {s_['engine_fails']}")

        out("\n## (c) synthetic fixtures of round 2, tests/golden/independent_solutions.npz (proxy at tol 1e-8)")
        g2 = np.load(os.path.join(ROOT, "tests", "golden", "independent_solutions.npz"))
        for name, V, cc in (("c2", 4, False), ("c3", 8, True)):
            inp = synth.solver_inputs(160, V, seed=0)
            d = {k: inp[k] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
            ip = {k: g2[f"{name}_{k}"] for k in ("u0", "U", "X", "status", "iters")}
            study(f"synthetic {name}", d, cc, ip, pool, out)

        out("\n## (b) config 3 instances the engine ends with status 5 (converged on the d = 1 discontinuity)")
        inp = synth.solver_inputs(4096, 8, seed=0)
        orc = oracle_lib.solve_batch(REF, inp["state"], inp["ego_index"], inp["weights"], inp["is_collide"], vref=inp["vref"],
                                     others=inp["others"], collision_cost=True, max_iter=1000, xy_bounds=False, nthreads=8)
        k5 = np.nonzero(orc["status"] == 5)[0]
        d = {k: inp[k] for k in ("state", "ego_index", "vref", "weights", "is_collide", "others")}
        res = pool.map(_ip_from, [(d, True, int(b), None, None) for b in k5])
        st = np.array([r[2] for r in res])
        it = np.array([r[3] for r in res])
        U = np.array([r[0] for r in res])
        X = np.array([r[1] for r in res])
        e5 = rel(U[:, 0], orc["u0"][k5])
        # distance of the proxy's final iterate to the nearest vehicle over the horizon, at the pair the engine holds at d = 1
        oth = inp["others"][k5]
        k = np.arange(N + 1)[None, :, None]
        ox = oth[:, None, :, 0] + k * 0.1 * oth[:, None, :, 2] * np.cos(oth[:, None, :, 3])
        oy = oth[:, None, :, 1] + k * 0.1 * oth[:, None, :, 2] * np.sin(oth[:, None, :, 3])
        dmin = lambda XX: np.sqrt((XX[:, :, None, 0] - ox) ** 2 + (XX[:, :, None, 1] - oy) ** 2)[:, 1:N].min(axis=(1, 2))
        dm_i, dm_o = dmin(X), dmin(orc["X"][k5])
        out(f"{k5.size} of 4096 instances (seed 0).  The proxy from its cold start: converged (status 0) {int((st == 0).sum())}, "
            f"restoration failed (5) {int((st == 5).sum())}, restoration ended locally infeasible (6) {int((st == 6).sum())}, iteration limit 1000 (1) "
            f"{int((st == 1).sum())}, inertia correction failed (2) {int((st == 2).sum())}; iterations median {np.median(it):.0f} "
            f"max {it.max()}")
        out(f"  engine's answers hold a vehicle at d = {np.median(dm_o):.6f} (median of the minimum distance over the horizon, "
            f"min {dm_o.min():.6f} max {dm_o.max():.6f})")
        for code, label in ((0, "converged"), (5, "restoration failed"), (6, "locally infeasible"), (1, "iteration limit")):
            sel = st == code
            if sel.any():
                out(f"  proxy {label}: {int(sel.sum())}; its final iterate's minimum distance: median {np.median(dm_i[sel]):.4f} "
                    f"(inside d < 1: {int((dm_i[sel] < 1 - 1e-9).sum())}, within 1e-3 of d = 1: {int((np.abs(dm_i[sel] - 1) < 1e-3).sum())}, "
                    f"outside: {int((dm_i[sel] > 1 + 1e-3).sum())}); u0 within 1e-4 of the engine's: {int((e5[sel] <= 1e-4).sum())}, "
                    f"median |du0| {np.median(e5[sel]):.2e}")
    return lines


if __name__ == "__main__":
    main()
