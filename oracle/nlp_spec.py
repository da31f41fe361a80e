"""TEST INFRASTRUCTURE - numpy restatement of the reference NLP (not shipped, not measured).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

This module restates, in plain numpy float64, the nonlinear programme that
`PureMPC_Agent._solve` builds with CasADi (reference `agents/pure_mpc.py:80-318`)
and provides an *independent KKT certifier* for candidate solutions.  The
reference's arithmetic lives in casadi==3.6.6 / IPOPT, which is not installed in
this image, and the reference has no tests or golden
vectors of its own.  What pins this file (round 4): the reference's OWN objective,
constraint, bound and initial-guess statements (agents/pure_mpc.py:128-283) were
executed with a numeric stand-in for casadi's SX (tests/golden/standins.py: it holds
values, the reference's code does the arithmetic) at 1200 points of 400 closed-loop
steps -> tests/golden/reference_sequences.npz; `objective`, `constraints`, the bounds
and the cold start of this module equal them to 2e-16 relative in f and exactly in g
(tests/test_reference_vectors.py::test_oracle_nlp_equals_the_references_statements).
The SOLVER (IPOPT) remains PARITY UNPINNED by the reference: solutions are pinned by
(1) KKT certificates of this NLP, (2) an independent restatement of IPOPT's published
algorithm (oracle/ipopt_restated.py) and scipy (oracle/scipy_crosscheck.py),
(3) analytic known-answer tests.

Decision variables follow the reference layout (`pure_mpc.py:260`):
    X[k] = (x, y, theta, v), k = 0..N      U[k] = (a, delta), k = 0..N-1
"""
from __future__ import annotations

import dataclasses
import numpy as np

# ---- constants of the reference -------------------------------------------------
WHEELBASE = 2.5          # agents/utils.py:18   Vehicle.LENGTH
LR_OVER_L = 0.5          # agents/utils.py:19   LENGTH_REAR / LENGTH
STATE_COST_MULT = 10.0   # agents/pure_mpc.py:206  (literal 10, not a config weight)
X_LO = np.array([-500.0, -500.0, -np.pi, 0.0])      # agents/pure_mpc.py:273
X_HI = np.array([500.0, 500.0, np.pi, 30.0])        # agents/pure_mpc.py:274
U_LO = np.array([-5.0, -np.pi / 3])                 # agents/pure_mpc.py:279
U_HI = np.array([5.0, np.pi / 3])                   # agents/pure_mpc.py:280
COLLIDE_SPEED_WEIGHT = 100.0                        # agents/pure_mpc.py:144-147


def reference_states(dt: float = 0.1) -> np.ndarray:
    """85x4 table, columns [x, y, v, heading]  (agents/base_agent.py:118-154)."""
    rows = []
    x, y, v, heading = 2.0, 50.0, 10.0, -np.pi / 2
    for _ in range(40):                       # straight segment  :127-132
        y += v * dt * np.sin(heading)
        rows.append((x, y, v, heading))
    inc = (np.pi / 2) / 20                    # quarter turn      :135-143
    for _ in range(20):
        heading -= inc
        x += v * dt * np.cos(heading)
        y += v * dt * np.sin(heading)
        rows.append((x, y, v, heading))
    for _ in range(25):                       # exit straight     :146-152
        x += v * dt * np.cos(heading)
        rows.append((x, y, v, heading))
    return np.array(rows)


@dataclasses.dataclass
class Problem:
    """One MPC instance = the data `_solve` closes over (pure_mpc.py:95-117)."""
    N: int
    dt: float
    state: np.ndarray            # (4,) x, y, theta, v      pure_mpc.py:233-238
    ref: np.ndarray              # (N+1, 4) window rows [x, y, v, heading] = ref[min(ego_index+k, M-1)]  :129
    w_speed: float               # already 100 if is_collide  :143-147
    w_control: float
    w_input_diff: float
    # optional "collision cost on" term (archive/pure_mpc.py:189-206, dead branch pure_mpc.py:169-183)
    collision_cost: bool = False
    others: np.ndarray | None = None   # (V, 4) x, y, speed, heading
    is_collide: bool = False
    w_distance: float = 10.0           # config/cfg.yaml:105
    w_collision: float = 1.0           # config/cfg.yaml:106

    @staticmethod
    def build(N, dt, state, ego_index, ref_table, weights, is_collide,
              collision_cost=False, others=None, w_distance=10.0, w_collision=1.0):
        M = ref_table.shape[0]
        idx = np.minimum(ego_index + np.arange(N + 1), M - 1)
        ws = COLLIDE_SPEED_WEIGHT if is_collide else float(weights[0])
        return Problem(N=N, dt=dt, state=np.asarray(state, dtype=np.float64),
                       ref=np.array(ref_table[idx], dtype=np.float64), w_speed=ws,
                       w_control=float(weights[1]), w_input_diff=float(weights[2]),
                       collision_cost=collision_cost,
                       others=None if others is None else np.asarray(others, dtype=np.float64),
                       is_collide=bool(is_collide), w_distance=w_distance, w_collision=w_collision)

    # -- other vehicles: stage k sees p_j + k * speed_j*dt*(cos h_j, sin h_j)   pure_mpc.py:190-191
    def other_positions(self, k):
        o = self.others
        step = o[:, 2:3] * self.dt * np.stack([np.cos(o[:, 3]), np.sin(o[:, 3])], axis=1)
        return o[:, :2] + k * step


# ---- dynamics (agents/pure_mpc.py:220-228) ----------------------------------------
def f_dyn(x, u):
    beta = np.arctan(LR_OVER_L * np.tan(u[1]))
    return np.array([x[3] * np.cos(x[2] + beta),
                     x[3] * np.sin(x[2] + beta),
                     x[3] / WHEELBASE * np.sin(beta),
                     u[0]])


def f_jac(x, u):
    """A = df/dx (4x4), B = df/du (4x2), analytic."""
    t = np.tan(u[1])
    beta = np.arctan(LR_OVER_L * t)
    dbeta = LR_OVER_L * (1 + t * t) / (1 + (LR_OVER_L * t) ** 2)
    s, c = np.sin(x[2] + beta), np.cos(x[2] + beta)
    v = x[3]
    A = np.zeros((4, 4))
    A[0, 2] = -v * s; A[0, 3] = c
    A[1, 2] = v * c;  A[1, 3] = s
    A[2, 3] = np.sin(beta) / WHEELBASE
    B = np.zeros((4, 2))
    B[0, 1] = -v * s * dbeta
    B[1, 1] = v * c * dbeta
    B[2, 1] = v / WHEELBASE * np.cos(beta) * dbeta
    B[3, 0] = 1.0
    return A, B


# ---- objective (agents/pure_mpc.py:128-212) -------------------------------------
def cost(p: Problem, X, U) -> float:
    J_state = 0.0
    J_ctrl = 0.0
    J_diff = 0.0
    J_dist = 0.0
    J_coll = 0.0
    for k in range(p.N):
        rx, ry, rv, rh = p.ref[k]
        dx = X[k, 0] - rx
        dy = X[k, 1] - ry
        perp = dx * np.sin(rh) - dy * np.cos(rh)
        para = dx * np.cos(rh) + dy * np.sin(rh)
        J_state += 4 * perp ** 2 + 2 * para ** 2 + p.w_speed * (X[k, 3] - rv) ** 2 + 0.5 * (X[k, 2] - rh) ** 2
        J_ctrl += 0.01 * U[k, 0] ** 2 + 0.01 * U[k, 1] ** 2
        if k > 0:
            J_diff += 0.01 * ((U[k, 0] - U[k - 1, 0]) ** 2 + (U[k, 1] - U[k - 1, 1]) ** 2)
        if p.collision_cost:
            if p.others is not None and len(p.others):
                d = np.linalg.norm(X[k, :2][None, :] - p.other_positions(k), axis=1)
                J_dist += np.sum(np.where(d < 1.0, 1000.0, 100.0) / (d + 1e-6) ** 2)
            if p.is_collide:
                J_coll += 3000.0 * X[k, 3] ** 2
    J = STATE_COST_MULT * J_state + p.w_control * J_ctrl + p.w_input_diff * J_diff
    if p.collision_cost:
        J += p.w_distance * J_dist + p.w_collision * J_coll
    return float(J)


def cost_grad(p: Problem, X, U):
    gX = np.zeros_like(X)
    gU = np.zeros_like(U)
    for k in range(p.N):
        rx, ry, rv, rh = p.ref[k]
        s, c = np.sin(rh), np.cos(rh)
        dx = X[k, 0] - rx
        dy = X[k, 1] - ry
        perp = dx * s - dy * c
        para = dx * c + dy * s
        m = STATE_COST_MULT
        gX[k, 0] += m * (8 * perp * s + 4 * para * c)
        gX[k, 1] += m * (-8 * perp * c + 4 * para * s)
        gX[k, 2] += m * (X[k, 2] - rh)
        gX[k, 3] += m * 2 * p.w_speed * (X[k, 3] - rv)
        gU[k] += p.w_control * 0.02 * U[k]
        if k > 0:
            dU = U[k] - U[k - 1]
            gU[k] += p.w_input_diff * 0.02 * dU
            gU[k - 1] -= p.w_input_diff * 0.02 * dU
        if p.collision_cost:
            if p.others is not None and len(p.others):
                dp = X[k, :2][None, :] - p.other_positions(k)
                d = np.linalg.norm(dp, axis=1)
                cc = np.where(d < 1.0, 1000.0, 100.0)
                dpsi = -2 * cc / (d + 1e-6) ** 3
                gX[k, :2] += p.w_distance * np.sum((dpsi / d)[:, None] * dp, axis=0)
            if p.is_collide:
                gX[k, 3] += p.w_collision * 6000.0 * X[k, 3]
    return gX, gU


# ---- equality constraints (agents/pure_mpc.py:249-257) ----------------------------
def constraints(p: Problem, X, U):
    """g = [X0 - state ; X[k+1] - X[k] - f(X[k],U[k]) dt]  -> (N+1, 4)."""
    g = np.zeros((p.N + 1, 4))
    g[0] = X[0] - p.state
    for k in range(p.N):
        g[k + 1] = X[k + 1] - (X[k] + f_dyn(X[k], U[k]) * p.dt)
    return g


def pack(X, U):
    return np.concatenate([X.ravel(), U.ravel()])


def unpack(p: Problem, z):
    n = 4 * (p.N + 1)
    return z[:n].reshape(p.N + 1, 4), z[n:].reshape(p.N, 2)


def constraint_jac_dense(p: Problem, X, U):
    """dense d g / d z,  z = [X.ravel(), U.ravel()]  -> (4(N+1), 6N+4)."""
    N = p.N
    nX = 4 * (N + 1)
    J = np.zeros((nX, nX + 2 * N))
    J[0:4, 0:4] = np.eye(4)
    for k in range(N):
        A, B = f_jac(X[k], U[k])
        r = 4 * (k + 1)
        J[r:r + 4, 4 * (k + 1):4 * (k + 2)] = np.eye(4)
        J[r:r + 4, 4 * k:4 * (k + 1)] = -(np.eye(4) + p.dt * A)
        J[r:r + 4, nX + 2 * k:nX + 2 * k + 2] = -p.dt * B
    return J


def bounds(p: Problem):
    lo = np.concatenate([np.tile(X_LO, p.N + 1), np.tile(U_LO, p.N)])
    hi = np.concatenate([np.tile(X_HI, p.N + 1), np.tile(U_HI, p.N)])
    return lo, hi


def initial_guess(p: Problem):
    """cold start of the reference (pure_mpc.py:240-246): all states = state, controls 0."""
    return np.tile(p.state, (p.N + 1, 1)), np.zeros((p.N, 2))


# ---- KKT certificate --------------------------------------------------------------
def kkt_certificate(p: Problem, X, U, act_tol=1e-6, relax=1e-8):
    """Independent first-order optimality certificate for a candidate (X, U).

    Multipliers are NOT taken from the solver: the equality multipliers and the
    multipliers of bounds active within `act_tol` are recovered by a dense
    least-squares fit of the stationarity equations, then checked for sign.
    `relax` mirrors IPOPT's bound_relax_factor (1e-8), under which the reference's
    answer is feasible.
    Returns dict(stationarity, feasibility, bound_violation, min_bound_mult, n_active, lam).
    """
    z = pack(X, U)
    lo, hi = bounds(p)
    gX, gU = cost_grad(p, X, U)
    grad = pack(gX, gU)
    g = constraints(p, X, U).ravel()
    J = constraint_jac_dense(p, X, U)
    act_lo = np.where(z - lo <= act_tol)[0]
    act_hi = np.where(hi - z <= act_tol)[0]
    # X[0] is pinned by the equality X[0] = state: its bound multipliers are redundant with lam_0
    act_lo = act_lo[act_lo >= 4]
    act_hi = act_hi[act_hi >= 4]
    n = z.size
    E_lo = np.zeros((n, act_lo.size)); E_lo[act_lo, np.arange(act_lo.size)] = -1.0
    E_hi = np.zeros((n, act_hi.size)); E_hi[act_hi, np.arange(act_hi.size)] = 1.0
    Mmat = np.concatenate([J.T, E_lo, E_hi], axis=1)
    # multipliers of active bounds must be non-negative: bounded least squares (lam free, z >= 0)
    from scipy.optimize import lsq_linear
    nlam = J.shape[0]
    lb = np.concatenate([np.full(nlam, -np.inf), np.zeros(Mmat.shape[1] - nlam)])
    gscale = max(1.0, float(np.max(np.abs(grad))))
    sol = lsq_linear(Mmat, -grad / gscale, bounds=(lb, np.full(Mmat.shape[1], np.inf)),
                     method="bvls", tol=1e-14, max_iter=2000).x * gscale
    resid = grad + Mmat @ sol
    mult_b = sol[J.shape[0]:]
    viol = max(0.0, float(np.max(lo - z)), float(np.max(z - hi)))
    return dict(stationarity=float(np.max(np.abs(resid))),
                feasibility=float(np.max(np.abs(g))),
                bound_violation=max(0.0, viol - relax * np.pi),
                min_bound_mult=float(mult_b.min()) if mult_b.size else 0.0,
                n_active=int(act_lo.size + act_hi.size),
                lam=sol[:J.shape[0]].reshape(p.N + 1, 4))
