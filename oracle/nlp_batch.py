"""TEST INFRASTRUCTURE - batched (vectorised over instances and stages) restatement of the reference NLP.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Same programme as oracle/nlp_spec.py (reference `agents/pure_mpc.py:80-318`, collision-cost term
`agents/archive/pure_mpc.py:189-206`), written for whole batches so that every solution a GPU launch returns
can be certified, and with the second derivatives an interior-point restatement of IPOPT needs
(oracle/ipopt_restated.py).  nlp_spec.py stays the readable loop version; tests/test_oracle.py checks the
two against each other and the derivatives against finite differences.

Layout: X[B, N+1, 4] = (x, y, theta, v), U[B, N, 2] = (a, delta)   (pure_mpc.py:88-93, 260).
Equalities c[B, N+1, 4]: c_0 = X_0 - state, c_{k+1} = X_{k+1} - X_k - dt f(X_k, U_k)   (pure_mpc.py:249-257).
"""
from __future__ import annotations

import dataclasses

import numpy as np

WHEELBASE = 2.5            # agents/utils.py:18
X_LO = np.array([-500.0, -500.0, -np.pi, 0.0])      # agents/pure_mpc.py:272-274
X_HI = np.array([500.0, 500.0, np.pi, 30.0])
U_LO = np.array([-5.0, -np.pi / 3])                 # agents/pure_mpc.py:278-280
U_HI = np.array([5.0, np.pi / 3])


@dataclasses.dataclass
class Batch:
    """Problem data of B instances (what `_solve` closes over, pure_mpc.py:95-117)."""
    N: int
    dt: float
    state: np.ndarray      # [B, 4]
    ref: np.ndarray        # [B, N+1, 4]  window rows (x, y, v, heading) = table[min(ego_index + k, M-1)]   :129
    ws: np.ndarray         # [B]  speed weight, already 100 where is_collide   :143-147
    wc: np.ndarray         # [B]
    wd: np.ndarray         # [B]
    cc: bool = False       # "collision cost on" (archive/pure_mpc.py:189-206)
    others: np.ndarray | None = None   # [B, V, 4]  x, y, speed, heading
    wdist: float = 10.0    # config/cfg.yaml:105
    wcoll: np.ndarray | None = None    # [B]  w_collision * 3000 * is_collide

    @property
    def B(self):
        return self.state.shape[0]

    @staticmethod
    def build(ref_table, state, ego_index, weights, is_collide, vref=None, others=None, N=20, dt=0.1,
              collision_cost=False, w_distance=10.0, w_collision=1.0):
        state = np.asarray(state, dtype=np.float64)
        B = state.shape[0]
        M = ref_table.shape[0]
        idx = np.clip(np.asarray(ego_index)[:, None] + np.arange(N + 1)[None, :], 0, M - 1)
        ref = np.array(ref_table[idx], dtype=np.float64)
        if vref is not None:
            ref[:, :, 2] = vref
        w = np.asarray(weights, dtype=np.float64)
        col = np.asarray(is_collide).astype(bool)
        ws = np.where(col, 100.0, w[:, 0])
        return Batch(N=N, dt=dt, state=state, ref=ref, ws=ws, wc=w[:, 1].copy(), wd=w[:, 2].copy(),
                     cc=bool(collision_cost),
                     others=None if others is None else np.asarray(others, dtype=np.float64),
                     wdist=w_distance, wcoll=np.where(col, 3000.0 * w_collision, 0.0) if collision_cost else np.zeros(B))

    def take(self, sel):
        return Batch(N=self.N, dt=self.dt, state=self.state[sel], ref=self.ref[sel], ws=self.ws[sel], wc=self.wc[sel],
                     wd=self.wd[sel], cc=self.cc, others=None if self.others is None else self.others[sel],
                     wdist=self.wdist, wcoll=None if self.wcoll is None else self.wcoll[sel])

    def other_pos(self):
        """[B, N, V, 2]: vehicle j as stage k sees it, p_j + k speed_j dt (cos h_j, sin h_j)   (pure_mpc.py:190-191)."""
        o = self.others
        step = o[:, :, 2:3] * self.dt * np.stack([np.cos(o[:, :, 3]), np.sin(o[:, :, 3])], axis=-1)
        k = np.arange(self.N)[None, :, None, None]
        return o[:, None, :, :2] + k * step[:, None, :, :]


# ---------------------------------------------------------------- dynamics (pure_mpc.py:220-228)
def dyn(X, U):
    """f(X_k, U_k) for k < N and the pieces its derivatives need.  X[B, N(+1), 4], U[B, N, 2]."""
    N = U.shape[1]
    th, v = X[:, :N, 2], X[:, :N, 3]
    t = np.tan(U[:, :, 1])
    beta = np.arctan(0.5 * t)
    den = 4.0 + t * t
    bp = 2.0 * (1.0 + t * t) / den
    bpp = 12.0 * t * (1.0 + t * t) / (den * den)
    S, C = np.sin(th + beta), np.cos(th + beta)
    sb, cb = np.sin(beta), np.cos(beta)
    f = np.stack([v * C, v * S, v / WHEELBASE * sb, U[:, :, 0]], axis=-1)
    return f, dict(S=S, C=C, sb=sb, cb=cb, bp=bp, bpp=bpp, v=v)


def dyn_jac(d):
    """A[B, N, 4, 4] = df/dx, Bm[B, N, 4, 2] = df/du."""
    S, C, sb, cb, bp, v = d["S"], d["C"], d["sb"], d["cb"], d["bp"], d["v"]
    A = np.zeros(S.shape + (4, 4))
    Bm = np.zeros(S.shape + (4, 2))
    A[..., 0, 2] = -v * S
    A[..., 0, 3] = C
    A[..., 1, 2] = v * C
    A[..., 1, 3] = S
    A[..., 2, 3] = sb / WHEELBASE
    Bm[..., 0, 1] = -v * S * bp
    Bm[..., 1, 1] = v * C * bp
    Bm[..., 2, 1] = v / WHEELBASE * cb * bp
    Bm[..., 3, 0] = 1.0
    return A, Bm


def constraints(p: Batch, X, U):
    f, _ = dyn(X, U)
    c = np.zeros_like(X)
    c[:, 0] = X[:, 0] - p.state
    c[:, 1:] = X[:, 1:] - X[:, :-1] - p.dt * f
    return c


# ---------------------------------------------------------------- objective (pure_mpc.py:128-212)
def _collision_terms(p: Batch, X, want_hess):
    """value [B], gradient [B, N, 2] and Hessian [B, N, 2, 2] of w_distance * sum_kj psi(d_kj) wrt (x_k, y_k)."""
    dp = X[:, :p.N, None, :2] - p.other_pos()               # [B, N, V, 2]
    d = np.sqrt(np.sum(dp * dp, axis=-1))
    cst = np.where(d < 1.0, 1000.0, 100.0) * p.wdist         # archive/pure_mpc.py:189-196 (branch frozen in derivatives)
    de = d + 1e-6
    val = np.sum(cst / (de * de), axis=(1, 2))
    dpsi = -2.0 * cst / de ** 3
    n = dp / d[..., None]
    g = np.sum(dpsi[..., None] * n, axis=2)
    H = None
    if want_hess:
        d2psi = 6.0 * cst / de ** 4
        tt = dpsi / d
        nn = n[..., :, None] * n[..., None, :]
        H = np.sum(d2psi[..., None, None] * nn + tt[..., None, None] * (np.eye(2) - nn), axis=2)
    return val, g, H


def cost(p: Batch, X, U):
    N = p.N
    r = p.ref[:, :N]
    s, c = np.sin(r[..., 3]), np.cos(r[..., 3])
    dx, dy = X[:, :N, 0] - r[..., 0], X[:, :N, 1] - r[..., 1]
    perp = dx * s - dy * c
    para = dx * c + dy * s
    Js = np.sum(4 * perp ** 2 + 2 * para ** 2 + p.ws[:, None] * (X[:, :N, 3] - r[..., 2]) ** 2
                + 0.5 * (X[:, :N, 2] - r[..., 3]) ** 2, axis=1)
    Jc = 0.01 * np.sum(U ** 2, axis=(1, 2))
    Jd = 0.01 * np.sum((U[:, 1:] - U[:, :-1]) ** 2, axis=(1, 2))
    J = 10.0 * Js + p.wc * Jc + p.wd * Jd
    if p.cc:
        if p.others is not None and p.others.shape[1]:
            J = J + _collision_terms(p, X, False)[0]
        J = J + p.wcoll * np.sum(X[:, :N, 3] ** 2, axis=1)
    return J


def cost_grad(p: Batch, X, U):
    N = p.N
    r = p.ref[:, :N]
    s, c = np.sin(r[..., 3]), np.cos(r[..., 3])
    dx, dy = X[:, :N, 0] - r[..., 0], X[:, :N, 1] - r[..., 1]
    perp = dx * s - dy * c
    para = dx * c + dy * s
    gX = np.zeros_like(X)
    gX[:, :N, 0] = 10.0 * (8 * perp * s + 4 * para * c)
    gX[:, :N, 1] = 10.0 * (-8 * perp * c + 4 * para * s)
    gX[:, :N, 2] = 10.0 * (X[:, :N, 2] - r[..., 3])
    gX[:, :N, 3] = 20.0 * p.ws[:, None] * (X[:, :N, 3] - r[..., 2])
    gU = 0.02 * p.wc[:, None, None] * U
    dU = 0.02 * p.wd[:, None, None] * (U[:, 1:] - U[:, :-1])
    gU[:, 1:] += dU
    gU[:, :-1] -= dU
    if p.cc:
        if p.others is not None and p.others.shape[1]:
            gX[:, :N, :2] += _collision_terms(p, X, False)[1]
        gX[:, :N, 3] += 2.0 * p.wcoll[:, None] * X[:, :N, 3]
    return gX, gU


def lagrangian_hessian_blocks(p: Batch, X, U, lam, sf=1.0):
    """Blocks of  sf * d2 f + sum_k lam_k . d2 c_k  (IPOPT sign: L = f + lam'c).

    Returns Qxx[B, N+1, 4, 4] (node N is zero), Qxu[B, N, 4, 2], Quu[B, N, 2, 2] (same-stage blocks) and the
    scalar r[B] of the only cross-stage block  d2L / dU_k dU_{k-1} = -r I  (input-rate cost, pure_mpc.py:164-165).
    `sf` may be an array [B] (objective scaling)."""
    B, N = X.shape[0], p.N
    sf = np.broadcast_to(np.asarray(sf, dtype=np.float64), (B,))
    r = p.ref[:, :N]
    s, c = np.sin(r[..., 3]), np.cos(r[..., 3])
    Qxx = np.zeros((B, N + 1, 4, 4))
    Qxu = np.zeros((B, N, 4, 2))
    Quu = np.zeros((B, N, 2, 2))
    Qxx[:, :N, 0, 0] = 10.0 * (8 * s * s + 4 * c * c)
    Qxx[:, :N, 0, 1] = Qxx[:, :N, 1, 0] = 10.0 * (-4 * s * c)
    Qxx[:, :N, 1, 1] = 10.0 * (8 * c * c + 4 * s * s)
    Qxx[:, :N, 2, 2] = 10.0
    Qxx[:, :N, 3, 3] = 20.0 * p.ws[:, None]
    if p.cc:
        if p.others is not None and p.others.shape[1]:
            Qxx[:, :N, :2, :2] += _collision_terms(p, X, True)[2]
        Qxx[:, :N, 3, 3] += 2.0 * p.wcoll[:, None]
    Qxx *= sf[:, None, None, None]
    rr = 0.02 * p.wd * sf
    diag = 0.02 * p.wc[:, None] * sf[:, None] + rr[:, None] * ((np.arange(N) >= 1).astype(float) + (np.arange(N) <= N - 2))[None, :]
    Quu[:, :, 0, 0] = diag
    Quu[:, :, 1, 1] = diag
    # constraint curvature: c_{k+1} = ... - dt f(X_k, U_k)  ->  -dt * sum_i lam_{k+1,i} d2 f_i over (theta, v, delta)
    _, d = dyn(X, U)
    S, C, sb, cb, bp, bpp, v = d["S"], d["C"], d["sb"], d["cb"], d["bp"], d["bpp"], d["v"]
    l0, l1, l2 = lam[:, 1:, 0], lam[:, 1:, 1], lam[:, 1:, 2]
    m = -p.dt
    Qxx[:, :N, 2, 2] += m * (l0 * (-v * C) + l1 * (-v * S))
    tv = m * (l0 * (-S) + l1 * C)
    Qxx[:, :N, 2, 3] += tv
    Qxx[:, :N, 3, 2] += tv
    Qxu[:, :, 2, 1] = m * (l0 * (-v * C * bp) + l1 * (-v * S * bp))
    Qxu[:, :, 3, 1] = m * (l0 * (-S * bp) + l1 * (C * bp) + l2 * cb * bp / WHEELBASE)
    Quu[:, :, 1, 1] += m * (l0 * (-v * C * bp * bp - v * S * bpp) + l1 * (-v * S * bp * bp + v * C * bpp)
                            + l2 * v / WHEELBASE * (-sb * bp * bp + cb * bpp))
    return Qxx, Qxu, Quu, rr


# ---------------------------------------------------------------- dense views of one instance (for the dense solver)
def pack(X, U):
    """z = [X.ravel(), U.ravel()] per instance  (pure_mpc.py:260)."""
    B = X.shape[0]
    return np.concatenate([X.reshape(B, -1), U.reshape(B, -1)], axis=1)


def unpack(N, z):
    B = z.shape[0]
    n = 4 * (N + 1)
    return z[:, :n].reshape(B, N + 1, 4), z[:, n:].reshape(B, N, 2)


def bounds_vec(N):
    lo = np.concatenate([np.tile(X_LO, N + 1), np.tile(U_LO, N)])
    hi = np.concatenate([np.tile(X_HI, N + 1), np.tile(U_HI, N)])
    return lo, hi


def jac_dense(p: Batch, X, U):
    """dc/dz [B, 4(N+1), 6N+4]."""
    B, N = X.shape[0], p.N
    nX = 4 * (N + 1)
    _, d = dyn(X, U)
    A, Bm = dyn_jac(d)
    J = np.zeros((B, nX, nX + 2 * N))
    J[:, np.arange(nX), np.arange(nX)] = 1.0
    for k in range(N):
        r = 4 * (k + 1)
        J[:, r:r + 4, 4 * k:4 * k + 4] = -(np.eye(4) + p.dt * A[:, k])
        J[:, r:r + 4, nX + 2 * k:nX + 2 * k + 2] = -p.dt * Bm[:, k]
    return J


def hess_dense(p: Batch, X, U, lam, sf=1.0):
    B, N = X.shape[0], p.N
    nX = 4 * (N + 1)
    Qxx, Qxu, Quu, rr = lagrangian_hessian_blocks(p, X, U, lam, sf)
    W = np.zeros((B, nX + 2 * N, nX + 2 * N))
    for k in range(N + 1):
        W[:, 4 * k:4 * k + 4, 4 * k:4 * k + 4] = Qxx[:, k]
    for k in range(N):
        a = nX + 2 * k
        W[:, a:a + 2, a:a + 2] = Quu[:, k]
        W[:, 4 * k:4 * k + 4, a:a + 2] = Qxu[:, k]
        W[:, a:a + 2, 4 * k:4 * k + 4] = np.swapaxes(Qxu[:, k], 1, 2)
        if k >= 1:
            for i in range(2):
                W[:, a + i, a - 2 + i] = -rr
                W[:, a - 2 + i, a + i] = -rr
    return W
