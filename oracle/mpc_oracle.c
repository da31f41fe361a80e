/* TEST INFRASTRUCTURE - CPU float64 oracle for the batched bicycle-model MPC solve.
 *
 * Not shipped and never on the product path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library (as the checker / the
 * reported CPU baseline).  The reference (SaeedRahmani/MPC-RL_for_AVs) has no tests or
 * golden vectors and its solver stack (casadi 3.6.6 -> IPOPT/MUMPS,
 * agents/pure_mpc.py:285-300) is not installed here.  The PROBLEM this file restates is
 * pinned by the reference's own statements executed numerically (tests/golden/
 * reference_sequences.npz via oracle/nlp_spec.py, tests/test_reference_vectors.py); the
 * SOLVER is PARITY UNPINNED by the reference: this file solves to a tighter tolerance
 * (1e-8) than the reference's IPOPT call (tol 1e-6) and its answers are pinned by the
 * independent KKT certifier (oracle/kkt_batch.py), by oracle/ipopt_restated.py (IPOPT's
 * published algorithm incl. restoration phase) and by oracle/scipy_crosscheck.py.
 *
 * Problem restated (all citations relative to /root/reference):
 *   variables   X[k]=(x,y,theta,v) k=0..N, U[k]=(a,delta) k=0..N-1      agents/pure_mpc.py:88-93,260
 *   objective   10*sum_k<N [4 perp^2 + 2 para^2 + ws (v-vref)^2 + .5 (theta-h)^2]
 *               + wc*.01*sum |U_k|^2 + wd*.01*sum_{k>=1} |U_k-U_{k-1}|^2  agents/pure_mpc.py:128-212
 *               ws = 100 when is_collide                                  agents/pure_mpc.py:143-147
 *   optional    + w_distance*sum_k sum_j (d<1?1000:100)/(d+1e-6)^2
 *               + w_collision*is_collide*3000*sum_k v_k^2                 agents/archive/pure_mpc.py:189-206,
 *                                                                         agents/pure_mpc.py:169-183
 *   dynamics    explicit-Euler kinematic bicycle, beta=atan(.5 tan delta)  agents/pure_mpc.py:220-257
 *   bounds      x,y in [-500,500], theta in [-pi,pi], v in [0,30],
 *               a in [-5,5], delta in [-pi/3,pi/3]                        agents/pure_mpc.py:272-280
 *   cold start  X[k]=state, U=0                                           agents/pure_mpc.py:240-246
 *
 * Algorithm (ours; the reference delegates to IPOPT): primal-dual interior-point
 * DDP in single shooting - monotone barrier decrease with IPOPT's error measures
 * and gradient-based objective scaling, exact Lagrangian Hessian (a stage whose
 * control block is not positive definite falls back to its Gauss-Newton terms;
 * if that fails the whole sweep uses the Gauss-Newton model, then a growing
 * diagonal shift; a Levenberg-Marquardt term adapted by the line search is kept
 * across iterations), stage-wise Riccati factorisation of the KKT system (state
 * augmented with the previous control to carry the input-rate cost), nonlinear
 * feedback rollouts with an Armijo line search on the barrier objective,
 * fraction-to-the-boundary rule, split dual step.  The dynamics hold exactly at
 * every iterate; X[0] is pinned by X[0] = state.  Round 2: a rollout that would take theta or v of the
 * next node out of its bounds gets the control that decides it pulled back (PROJ_KEEP); a vehicle that a
 * rejected trial took across the d = 1 discontinuity of the collision cost becomes a wall constraint
 * |p - o|^2 >= 1 of that node (status 5 when it ends with a multiplier); four line-search trials.
 * Round 5 (the iteration tail): IPOPT's inertia correction on the EXACT Hessian when the
 * Gauss-Newton fallback stagnates; Newton steps below the resolution of the feedback law applied open loop; the projection
 * floor lowered in the end-game; IPOPT's acceptable-level termination (status 6); the Levenberg-Marquardt term capped.
 * oracle_last_work() reports the iterations / sweeps / rollouts of the last call and their flops.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NMAX 64
#define VMAX 16
#define WHEELBASE 2.5
#define PI 3.14159265358979323846

typedef struct {
    int N, V, cc;
    double dt, x0[4];
    double rx[NMAX + 1], ry[NMAX + 1], rv[NMAX + 1], rh[NMAX + 1], rs[NMAX + 1], rc[NMAX + 1];
    double ws, wc, wd;
    double ox[VMAX], oy[VMAX], osx[VMAX], osy[VMAX];
    double wdist, wcoll; /* wcoll already multiplied by 3000*is_collide */
    int i0;              /* first bounded state component: 0 = all (reference NLP), 2 = theta, v only */
    double sf;           /* objective scale factor (IPOPT-style gradient-based scaling), 1 = unscaled */
} prob_t;

static const double XLO[4] = {-500.0, -500.0, -PI, 0.0};
static const double XHI[4] = {500.0, 500.0, PI, 30.0};
static const double ULO[2] = {-5.0, -PI / 3.0};
static const double UHI[2] = {5.0, PI / 3.0};

typedef struct {
    double f[4];                   /* f(x,u) */
    double S, C, sb, cb, bp, bpp;  /* sin/cos(theta+beta), sin/cos beta, beta', beta'' */
} dyn_t;

static void dyn_eval(const double *x, const double *u, dyn_t *d) {
    double t = tan(u[1]);
    double beta = atan(0.5 * t);
    double den = 4.0 + t * t;
    d->bp = 2.0 * (1.0 + t * t) / den;
    d->bpp = 12.0 * t * (1.0 + t * t) / (den * den);
    d->sb = sin(beta);
    d->cb = cos(beta);
    d->S = sin(x[2] + beta);
    d->C = cos(x[2] + beta);
    d->f[0] = x[3] * d->C;
    d->f[1] = x[3] * d->S;
    d->f[2] = x[3] / WHEELBASE * d->sb;
    d->f[3] = u[0];
}

/* stage state cost (node k, 1<=k<=N-1), gradient lx[4] and Hessian Q[4][4] */
static double stage_cost_raw(const prob_t *p, int k, const double *x, double *lx, double (*Q)[4], double (*Qg)[4]);
static double stage_cost(const prob_t *p, int k, const double *x, double *lx, double (*Q)[4], double (*Qg)[4]) {
    double J = stage_cost_raw(p, k, x, lx, Q, Qg);
    if (p->sf != 1.0) {
        if (lx) for (int i = 0; i < 4; ++i) lx[i] *= p->sf;
        if (Q) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Q[i][j] *= p->sf;
        if (Qg) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Qg[i][j] *= p->sf;
    }
    return J * p->sf;
}
static double stage_cost_raw(const prob_t *p, int k, const double *x, double *lx, double (*Q)[4], double (*Qg)[4]) {
    double s = p->rs[k], c = p->rc[k];
    double dx = x[0] - p->rx[k], dy = x[1] - p->ry[k];
    double perp = dx * s - dy * c, para = dx * c + dy * s;
    double dv = x[3] - p->rv[k], dth = x[2] - p->rh[k];
    double J = 10.0 * (4 * perp * perp + 2 * para * para + p->ws * dv * dv + 0.5 * dth * dth);
    if (lx) {
        lx[0] = 10.0 * (8 * perp * s + 4 * para * c);
        lx[1] = 10.0 * (-8 * perp * c + 4 * para * s);
        lx[2] = 10.0 * dth;
        lx[3] = 20.0 * p->ws * dv;
    }
    if (Q) {
        memset(Q, 0, 16 * sizeof(double));
        Q[0][0] = 10.0 * (8 * s * s + 4 * c * c);
        Q[0][1] = Q[1][0] = 10.0 * (-8 * s * c + 4 * c * s);
        Q[1][1] = 10.0 * (8 * c * c + 4 * s * s);
        Q[2][2] = 10.0;
        Q[3][3] = 20.0 * p->ws;
        if (Qg) memcpy(Qg, Q, 16 * sizeof(double));
    }
    if (p->cc) {
        for (int j = 0; j < p->V; ++j) {
            double px = x[0] - (p->ox[j] + k * p->osx[j]);
            double py = x[1] - (p->oy[j] + k * p->osy[j]);
            double d = sqrt(px * px + py * py);
            double cst = (d < 1.0 ? 1000.0 : 100.0) * p->wdist;
            double de = d + 1e-6;
            J += cst / (de * de);
            if (lx || Q) {
                double dpsi = -2.0 * cst / (de * de * de);
                double nx = px / d, ny = py / d;
                if (lx) {
                    lx[0] += dpsi * nx;
                    lx[1] += dpsi * ny;
                }
                if (Q) {
                    double d2psi = 6.0 * cst / (de * de * de * de);
                    double tt = dpsi / d;
                    Q[0][0] += d2psi * nx * nx + tt * (1 - nx * nx);
                    Q[0][1] += (d2psi - tt) * nx * ny;
                    Q[1][0] += (d2psi - tt) * nx * ny;
                    Q[1][1] += d2psi * ny * ny + tt * (1 - ny * ny);
                    if (Qg) { /* convex (Gauss-Newton-like) part: radial curvature only */
                        Qg[0][0] += d2psi * nx * nx;
                        Qg[0][1] += d2psi * nx * ny;
                        Qg[1][0] += d2psi * nx * ny;
                        Qg[1][1] += d2psi * ny * ny;
                    }
                }
            }
        }
        J += p->wcoll * x[3] * x[3];
        if (lx) lx[3] += 2.0 * p->wcoll * x[3];
        if (Q) Q[3][3] += 2.0 * p->wcoll;
        if (Q && Qg) Qg[3][3] += 2.0 * p->wcoll;
    }
    return J;
}

/* optional work counters (ORACLE_COUNT=1, single thread): iterations, backward sweeps, rollouts */
static long g_cnt_iter, g_cnt_sweep, g_cnt_roll, g_cnt_solve;
/* per-thread tallies of the instance being solved (no shared counters inside solve_one: with 32+ threads the atomics on
 * three global words were a measurable part of the run); summed by the OpenMP reduction of the batch loop */
static _Thread_local long t_cnt_iter, t_cnt_sweep, t_cnt_roll;
/* experiment knobs of tools/portfolio_study.py (never changed by tests or the bench): barrier start and backtracking factor */
static double g_exp_mu_init = 0.1, g_exp_btf = 0.25;
static double g_exp_warm_mu = 0.0; /* > 0 (warm-start study): barrier parameter a warm start begins with, multipliers mu / slack */
static int g_stall_window; /* mpc_config.stall_window of the engine under test (0 = off); oracle_set_stall_window */
static int g_trace, g_trace2; /* ORACLE_TRACE / ORACLE_TRACE2, read once per batch call */
static int g_cnt_N, g_cnt_V, g_cnt_cc;

typedef struct {
    double tol, mu_init;
    int max_iter;
} opts_t;

typedef struct {
    double x[NMAX + 1][4], u[NMAX][2];
    double zxl[NMAX + 1][4], zxu[NMAX + 1][4], zul[NMAX][2], zuu[NMAX][2];
    double lam[NMAX + 1][4]; /* equality multipliers (IPOPT sign: L = f + lam'g), filled at exit */
    /* d = 1 discontinuity of the collision cost (archive/pure_mpc.py:189-196: 100/d^2 outside, 1000/d^2 inside): per
     * node the nearest vehicle on the outer branch (-1: none) is kept outside by the constraint |p - o|^2 - 1 >= 0 with
     * multiplier zw - the cost jumps upwards when it is crossed inwards, so a minimiser pressed against d = 1 is a
     * constrained stationary point of the outer branch (status 5), which no smooth method reaches otherwise */
    double zw[NMAX + 1];
    int wj[NMAX + 1];
} iter_t;

/* bounds relaxed like IPOPT's bound_relax_factor = 1e-8 so that a strict interior always exists */
static double xlo_r(int i) { return XLO[i] - 1e-8 * fmax(1.0, fabs(XLO[i])); }
static double xhi_r(int i) { return XHI[i] + 1e-8 * fmax(1.0, fabs(XHI[i])); }
static double ulo_r(int i) { return ULO[i] - 1e-8 * fmax(1.0, fabs(ULO[i])); }
static double uhi_r(int i) { return UHI[i] + 1e-8 * fmax(1.0, fabs(UHI[i])); }

/* slack |p_k - o_jk|^2 - 1 of the wall constraint of node k and vehicle j */
static double wall_slack(const prob_t *p, int k, const double *x, int j) {
    double px = x[0] - (p->ox[j] + k * p->osx[j]), py = x[1] - (p->oy[j] + k * p->osy[j]);
    return px * px + py * py - 1.0;
}
/* barrier objective of a dynamically feasible trajectory (node-0 state cost is a constant and dropped) */
static double barrier_objective(const prob_t *p, const iter_t *it, double mu) {
    int N = p->N;
    double J = 0.0, bar = 0.0;
    for (int k = 0; k < N; ++k) {
        const double *u = it->u[k];
        if (k >= 1) J += stage_cost(p, k, it->x[k], NULL, NULL, NULL);
        J += 0.01 * p->sf * p->wc * (u[0] * u[0] + u[1] * u[1]);
        if (k >= 1) {
            double d0 = u[0] - it->u[k - 1][0], d1 = u[1] - it->u[k - 1][1];
            J += 0.01 * p->sf * p->wd * (d0 * d0 + d1 * d1);
        }
        for (int i = 0; i < 2; ++i) bar -= log(u[i] - ulo_r(i)) + log(uhi_r(i) - u[i]);
    }
    for (int k = 1; k <= N; ++k)
        for (int i = p->i0; i < 4; ++i) bar -= log(it->x[k][i] - xlo_r(i)) + log(xhi_r(i) - it->x[k][i]);
    for (int k = 1; k < N; ++k)
        if (it->wj[k] >= 0) bar -= log(wall_slack(p, k, it->x[k], it->wj[k]));
    return J + mu * bar;
}

/* solve one instance.
 * status: 0 converged, 1 max_iter reached, 2 factorisation failure, 3 start not strictly feasible,
 *         4 stalled (no acceptable step in three consecutive iterations),
 *         5 converged with a vehicle held at the d = 1 discontinuity of the collision cost,
 *         6 IPOPT's "solved to acceptable level" (scaled KKT error <= 1e-6 in 15 consecutive iterations while tol is tighter),
 *         7 the same with a vehicle held at d = 1 */
static int solve_one(prob_t *p, const opts_t *o, iter_t *it, const double *uinit, int *iters_out, double *kkt_out) {
    const int N = p->N;
    const double dt = p->dt;
    double rd_full = 0.02 * p->wd, rc = 0.02 * p->wc;
    double mu = o->mu_init;
    const double mu_min = o->tol / 10.0;
    int status = 1, iter = 0, nfail = 0;
    int i_mark = 0; /* progress guard: iteration at which the KKT error last fell below half of its value at the previous mark */
    double e_mark = INFINITY;
    const double KSIG = 1e10; /* IPOPT kappa_Sigma */
    const int MAXLS = 4;      /* line-search trials per iteration */
    const double BTF = g_exp_btf; /* backtracking factor, 0.25 (tools/portfolio_study.py may set another) */
    const double kap_eps = 30.0, kap_mu = 0.2; /* kappa_epsilon: 10 in IPOPT and here until round 3 */
    /* Levenberg-Marquardt term kept across iterations: added to the diagonal of the stage Hessians like delta_w.  Two or
     * more backtracks (or no acceptable step) multiply it by 4 (from 1e-3), a full first trial divides it by 4 (to 0
     * below 1e-3).  Without it instances on the nonconvex side of the heading wrap crawl with 1/64-steps for the whole
     * iteration budget. */
    /* REG_MAX: 1e6 until round 4.  An instance that needs more than the curvature scale of the scaled problem (1) to get a
     * step accepted crawls - alpha = 1/4 with the term at 16 for hundreds of iterations on BASELINE config 3's worst
     * instances (368 iterations to converge; IPOPT's restated algorithm gives up on the same instance after 1000) - so the
     * term stops at 1 and the line-search-failure exit (status 4) ends what cannot move. */
    const double REG_MIN = 1e-3, REG_FACTOR = 4.0, REG_MAX = 1.0;
    /* the multiple of the identity the last inertia correction ended with (0: none yet in this solve).  Until round 3 every
     * correction climbed 1e-8, 1e-6, ... x 100 from scratch: with NEGATIVE cost weights (the v1 input domain: the RL action
     * clipped to [-1, 1], agents/ppo_mpc.py:407-417) the control block needs ~1e-2 in every iteration, the ladder ended at 1
     * - five wasted sweeps per iteration and steps a hundred times shorter than the curvature warrants: 121 iterations on
     * average where IPOPT's rule takes 45 (tests/golden/closed_loop_ipopt.npz, scenario c4v1). */
    double dw_last = 0.0;
    /* a rollout that would take theta or v of the next node out of its bounds gets the one control that decides it
     * (delta resp. a) pulled back so that the node keeps PROJ_KEEP of its slack: the Newton direction of a single
     * shooting method knows the state bounds only to first order, and without this every trial of an instance that
     * runs along theta = -pi (the reference heading of the exit straight IS the bound, base_agent.py:146-152 vs
     * pure_mpc.py:273) is infeasible until the step is tiny (2048 instances of config 2: 66 -> 10 with 40 iterations
     * or more, none left at the cap) */
    const double PROJ_KEEP0 = 0.1;
    /* ... and PROJ_KEEP_END of it once the barrier parameter is at most PROJ_END_MU: in the end-game a slack has to shrink by
     * orders of magnitude to meet its multiplier, a projection that keeps 10 % per iteration turns the full Newton step into
     * something the Armijo test rejects (the barrier objective RISES by 40 x the predicted decrease) and the solve closes with
     * 15 - 20 steps of alpha = 1/4, the error falling by exactly 0.75 per iteration (round 4's trace of BASELINE config 2's
     * slowest instances, egos on the exit straight with the heading next to its bound): 46 -> 34 iterations there */
    const double PROJ_KEEP_END = 1e-3, PROJ_END_MU = 1e-6;
    /* no trial may bring a control or a bounded state nearer to its bound than this: late in a solve tau = 1 - mu lets a
     * slack shrink by the factor mu ~ 1e-9 per iteration, two such steps take a control at -5 below one ulp of its bound -
     * slack exactly 0, 1 / slack infinite, a NaN in the sweep that no regularisation repairs (status 2; seen on the GPU
     * about once in 4000 instances).  IPOPT has the same safeguard (Waechter & Biegler 2006, section 3.5: slacks that
     * become too small are corrected). */
    const double MIN_SLACK = 1e-14;
    double reg = 0.0;
    /* (vii) IPOPT's inertia correction on the exact Hessian when the Gauss-Newton fallback stagnates.  The Gauss-Newton model
     * is convex: near a saddle point of the barrier problem (a stationary point with an indefinite reduced Hessian) its steps
     * are attracted to the saddle, every one is accepted at full length with a predicted decrease of 1e-6, and the iterate
     * drifts away along the unstable direction by 8 % per iteration - half of round 4's instances at the iteration cap.  A
     * streak of iterations that needed the whole-sweep fallback is therefore watched: every GN_WATCH iterations of it the scaled
     * KKT error must have halved; if it has not, the next GN_SKIP iterations of the streak go from the failed exact sweep straight
     * to the exact Hessian + delta_w I ladder (Waechter & Biegler 2006, section 3.1, with its memory dw_last), whose step along a
     * direction of negative curvature is long.  (Always skipping the Gauss-Newton model is much worse: mean 19.8 iterations.) */
    const int GN_WATCH = 4, GN_SKIP = 4;
    const double GN_END_MU = 1e-2; /* (xi) below */
    int gn_streak = 0, gn_skip = 0;
    /* (xi) a negative cost weight (the v1 action domain, agents/ppo_mpc.py:407-420: weights anywhere in [-1, 1]^3) makes the stage
     * cost itself non-convex, so the "convex" Gauss-Newton model is not convex either: it only removes the constraint curvature,
     * which far from the solution is still the better model (fewer iterations), but in the end game it costs the quadratic
     * convergence and the iteration dithers between the two models at 1e-6 (9 of the 160 c4v1 fixture states ended at the cap;
     * tools/v1_study.py).  From the second barrier decrease on such an instance goes from the exact Hessian straight to the
     * inertia ladder. */
    const int nonconvex_cost = (p->ws < 0.0 || p->wc < 0.0 || p->wd < 0.0);
    /* ... and its control blocks are indefinite in nearly every iteration (sweeps per iteration 2.0 on the c4v1 fixture states):
     * an iteration that follows one which needed the ladder starts at the ladder's first rung directly (a third of the value that
     * worked, IPOPT's kappa_w^-), in the same model; every DW_PROBE-th iteration tries delta_w = 0 first as before, so an iterate
     * that has reached a region where the exact Hessian is positive definite is found out (1.99 -> 1.50 sweeps per iteration). */
    const int DW_PROBE = 4;
    int prev_needed = 0, prev_gn = 0;
    double e_streak = INFINITY;
    /* (viii) a Newton step smaller than OPEN_LOOP_STEP in the states and the previous control of a stage is applied OPEN LOOP
     * there (alpha times the linearised control step) instead of through the feedback law.  The feedback acts on the difference
     * of two rolled-out trajectories, which carries the rounding of positions ~50 m (7e-15); times gains of 10 - 100 where a
     * control and the state bound it decides are both active that is 5e-13 of noise in a control whose slack is 1e-9, and the
     * dual infeasibility dithers at 1e-6 - 1e-5 for the rest of the iteration budget (round 4's cap-runners with a standing ego).
     * The linearised step is a recursion on small numbers and keeps its relative precision; the two differ by O(step^2). */
    const double OPEN_LOOP_STEP = 1e-9;
    /* (ix) IPOPT's acceptable-level termination with IPOPT's defaults (acceptable_tol 1e-6, acceptable_iter 15; the
     * acceptable_dual_inf / constr_viol / compl_inf tolerances 1e10 / 1e-2 / 1e-2 cannot bind once the scaled error is below
     * 1e-6, acceptable_obj_change_tol 1e20 is off): live in the reference, which sets only max_iter, tol and print options
     * (agents/pure_mpc.py:291-296).  It can end a solve only when tol < acceptable_tol, i.e. never at the reference's tol 1e-6. */
    const double ACCEPTABLE_TOL = 1e-6;
    const int ACCEPTABLE_ITER = 15;
    int n_acceptable = 0;

    static _Thread_local double A[NMAX][4][4], Bm[NMAX][4][2];
    static _Thread_local double lxs[NMAX + 1][4], Qs[NMAX + 1][4][4], Qgs[NMAX + 1][4][4], lus[NMAX][2], lps[NMAX][2];
    static _Thread_local double Wtt[NMAX], Wtv[NMAX], Wtd[NMAX], Wvd[NMAX], Wdd[NMAX];
    static _Thread_local double Kx[NMAX][2][4], Kp[NMAX][2][2], kf[NMAX][2], yv[NMAX + 2][4];
    static _Thread_local iter_t trial;
    static _Thread_local double gw[NMAX + 1], wn[NMAX + 1][2]; /* wall slack and its gradient at the current iterate */

    /* ---- start: cold like the reference (pure_mpc.py:240-246: controls 0; a standing vehicle gets a_0 > 0 so that
     *      v_1.. are strictly inside v >= 0), or - opt-in, not in the reference - from given controls clamped 0.1 % of
     *      the range inside their bounds; a warm start whose rollout leaves the state bounds falls back to the cold one */
    int warm = uinit != NULL;
    for (;;) {
        memset(it, 0, sizeof(*it));
        for (int i = 0; i < 4; ++i) it->x[0][i] = p->x0[i];
        if (warm) {
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < 2; ++i) {
                    const double m = 1e-3 * (UHI[i] - ULO[i]);
                    it->u[k][i] = fmin(fmax(uinit[2 * k + i], ULO[i] + m), UHI[i] - m);
                }
        } else if (p->x0[3] < 0.01) {
            it->u[0][0] = (0.01 - p->x0[3]) / dt;
        }
        for (int k = 0; k < N; ++k) {
            dyn_t d;
            dyn_eval(it->x[k], it->u[k], &d);
            double n2 = it->x[k][2] + dt * d.f[2];
            /* cold start only: a node whose heading would sit on or outside its bound (theta_0 = -pi to float32 rounding
             * is 5.6e-8 outside the relaxed bound, and zero controls keep every node there) gets the steering angle that
             * puts it INIT_PUSH inside, less for a slow vehicle - IPOPT's bound_push on its starting point; the 5.6e-8
             * at node 0 are below its constraint tolerance */
            if (!warm && (!(n2 > xlo_r(2)) || !(n2 < xhi_r(2))) && it->x[k][3] > 1e-6) {
                const double INIT_PUSH = 1e-2;
                const double reach = dt * it->x[k][3] / WHEELBASE;
                const double push = fmin(INIT_PUSH, 0.25 * reach);
                const double target = !(n2 > xlo_r(2)) ? xlo_r(2) + push : xhi_r(2) - push;
                const double sreq = (target - it->x[k][2]) / reach;
                if (fabs(sreq) < 0.9) {
                    const double m = 1e-3 * (UHI[1] - ULO[1]);
                    it->u[k][1] = fmin(fmax(atan(2.0 * sreq / sqrt(1.0 - sreq * sreq)), ULO[1] + m), UHI[1] - m);
                    dyn_eval(it->x[k], it->u[k], &d);
                }
            }
            for (int i = 0; i < 4; ++i) it->x[k + 1][i] = it->x[k][i] + dt * d.f[i];
            for (int i = 0; i < 2; ++i) it->zul[k][i] = it->zuu[k][i] = 1.0;
            for (int i = p->i0; i < 4; ++i) it->zxl[k + 1][i] = it->zxu[k + 1][i] = 1.0;
        }
        int feasible = 1;
        for (int k = 1; k <= N; ++k)
            for (int i = p->i0; i < 4; ++i)
                if (!(it->x[k][i] > xlo_r(i)) || !(it->x[k][i] < xhi_r(i))) feasible = 0;
        for (int i = 0; i < 2; ++i)
            if (!(it->u[0][i] > ulo_r(i)) || !(it->u[0][i] < uhi_r(i))) feasible = 0;
        for (int k = 0; k <= N; ++k) {
            it->wj[k] = -1;
            it->zw[k] = 0.0;
        }
        if (feasible) break;
        if (!warm) {
            *iters_out = 0;
            if (kkt_out) *kkt_out = INFINITY;
            return 3;
        }
        warm = 0;
    }

    /* ---- objective scaling like IPOPT's gradient-based scaling (nlp_scaling_max_gradient = 100):
     *      sf = 100 / max(100, |grad f|_inf at the starting trajectory) */
    {
        double gmax = 0.0, lx[4];
        p->sf = 1.0;
        for (int k = 1; k < N; ++k) {
            stage_cost(p, k, it->x[k], lx, NULL, NULL);
            for (int i = 0; i < 4; ++i) gmax = fmax(gmax, fabs(lx[i]));
        }
        gmax = fmax(gmax, 0.02 * (p->wc + p->wd) * fabs(it->u[0][0]));
        p->sf = 100.0 / fmin(fmax(100.0, gmax), 1e4); /* capped: a start that grazes another vehicle must not loosen the tolerance */
        rd_full *= p->sf;
        rc *= p->sf;
    }

    if (uinit && warm && g_exp_warm_mu > 0.0) {
        /* warm-start study: start on the central path of a smaller barrier parameter at the warm point */
        mu = g_exp_warm_mu;
        for (int k = 0; k < N; ++k) {
            for (int i = 0; i < 2; ++i) {
                it->zul[k][i] = mu / (it->u[k][i] - ulo_r(i));
                it->zuu[k][i] = mu / (uhi_r(i) - it->u[k][i]);
            }
            for (int i = p->i0; i < 4; ++i) {
                it->zxl[k + 1][i] = mu / (it->x[k + 1][i] - xlo_r(i));
                it->zxu[k + 1][i] = mu / (xhi_r(i) - it->x[k + 1][i]);
            }
        }
    }
    for (iter = 0; iter <= o->max_iter; ++iter) {
        /* ---------------- stage derivatives along the current (feasible) trajectory ---------------- */
        for (int k = 0; k < N; ++k) {
            dyn_t d;
            const double *x = it->x[k], *u = it->u[k];
            dyn_eval(x, u, &d);
            memset(A[k], 0, sizeof(A[k]));
            memset(Bm[k], 0, sizeof(Bm[k]));
            for (int i = 0; i < 4; ++i) A[k][i][i] = 1.0;
            A[k][0][2] = -dt * x[3] * d.S;
            A[k][0][3] = dt * d.C;
            A[k][1][2] = dt * x[3] * d.C;
            A[k][1][3] = dt * d.S;
            A[k][2][3] = dt * d.sb / WHEELBASE;
            Bm[k][0][1] = -dt * x[3] * d.S * d.bp;
            Bm[k][1][1] = dt * x[3] * d.C * d.bp;
            Bm[k][2][1] = dt * x[3] / WHEELBASE * d.cb * d.bp;
            Bm[k][3][0] = dt;
            if (k >= 1) stage_cost(p, k, x, lxs[k], Qs[k], Qgs[k]);
            if (it->wj[k] >= 0) {
                int j = it->wj[k];
                gw[k] = wall_slack(p, k, x, j);
                wn[k][0] = 2.0 * (x[0] - (p->ox[j] + k * p->osx[j]));
                wn[k][1] = 2.0 * (x[1] - (p->oy[j] + k * p->osy[j]));
            }
            double rdk = (k >= 1) ? rd_full : 0.0;
            for (int i = 0; i < 2; ++i) {
                double dprev = (k >= 1) ? (u[i] - it->u[k - 1][i]) : 0.0;
                lus[k][i] = rc * u[i] + rdk * dprev;
                lps[k][i] = -rdk * dprev;
            }
        }
        memset(lxs[N], 0, sizeof(lxs[N]));
        memset(Qs[N], 0, sizeof(Qs[N]));
        memset(Qgs[N], 0, sizeof(Qgs[N]));

        /* ---------------- adjoint sweep: y_k = dL/dx_k with the current bound multipliers -------------
         * (y = -lam in IPOPT's sign convention); dual residual r_u and the constraint curvature terms */
        double err_d = 0.0, sum_lam = 0.0, sum_z = 0.0;
        int dbg_k = 0, dbg_i = 0; /* where the dual infeasibility is largest (ORACLE_TRACE) */
        for (int i = 0; i < 4; ++i) {
            yv[N][i] = -it->zxl[N][i] + it->zxu[N][i];
            yv[N + 1][i] = 0.0;
        }
        for (int k = N - 1; k >= 0; --k) {
            const double *y = yv[k + 1];
            for (int i = 0; i < 2; ++i) {
                double r = lus[k][i] - it->zul[k][i] + it->zuu[k][i] + ((k + 1 < N) ? lps[k + 1][i] : 0.0);
                for (int j = 0; j < 4; ++j) r += Bm[k][j][i] * y[j];
                if (fabs(r) > err_d) dbg_k = k, dbg_i = i;
                err_d = fmax(err_d, fabs(r));
                sum_z += it->zul[k][i] + it->zuu[k][i];
            }
            for (int i = 0; i < 4; ++i) {
                sum_lam += fabs(y[i]);
                if (i >= p->i0) sum_z += it->zxl[k + 1][i] + it->zxu[k + 1][i];
            }
            if (it->wj[k] >= 0) sum_z += it->zw[k];
            if (k >= 1)
                for (int i = 0; i < 4; ++i) {
                    double s = lxs[k][i] - it->zxl[k][i] + it->zxu[k][i];
                    if (i < 2 && it->wj[k] >= 0) s -= it->zw[k] * wn[k][i];
                    for (int j = 0; j < 4; ++j) s += A[k][j][i] * y[j];
                    yv[k][i] = s;
                }
            /* exact curvature  sum_i y_i * d2(dt f_i)  over (theta, v, delta) */
            dyn_t d;
            dyn_eval(it->x[k], it->u[k], &d);
            double v = it->x[k][3];
            double g = -(y[0] * d.C + y[1] * d.S);
            double h = -(y[0] * d.S - y[1] * d.C);
            Wtt[k] = dt * v * g;
            Wtv[k] = dt * h;
            Wtd[k] = dt * v * g * d.bp;
            Wvd[k] = dt * h * d.bp + dt * y[2] * d.cb * d.bp / WHEELBASE;
            Wdd[k] = dt * v * (g * d.bp * d.bp + h * d.bpp) +
                     dt * y[2] * v / WHEELBASE * (-d.sb * d.bp * d.bp + d.cb * d.bpp);
        }
        const int nvar = 6 * N, ncon = 4 * N; /* counts of the reference NLP, independent of i0 */
        double s_d = fmax(100.0, (sum_lam + sum_z) / (nvar + ncon)) / 100.0;
        double s_c = fmax(100.0, sum_z / nvar) / 100.0;
        double err_c0 = 0.0;
        for (;;) {
            double ec = 0.0;
            err_c0 = 0.0;
            for (int k = 1; k <= N; ++k)
                for (int i = p->i0; i < 4; ++i) {
                    double cl = (it->x[k][i] - xlo_r(i)) * it->zxl[k][i], cu = (xhi_r(i) - it->x[k][i]) * it->zxu[k][i];
                    ec = fmax(ec, fmax(fabs(cl - mu), fabs(cu - mu)));
                    err_c0 = fmax(err_c0, fmax(cl, cu));
                }
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < 2; ++i) {
                    double cl = (it->u[k][i] - ulo_r(i)) * it->zul[k][i], cu = (uhi_r(i) - it->u[k][i]) * it->zuu[k][i];
                    ec = fmax(ec, fmax(fabs(cl - mu), fabs(cu - mu)));
                    err_c0 = fmax(err_c0, fmax(cl, cu));
                }
            for (int k = 1; k < N; ++k)
                if (it->wj[k] >= 0) {
                    double cw = gw[k] * it->zw[k];
                    ec = fmax(ec, fabs(cw - mu));
                    err_c0 = fmax(err_c0, cw);
                }
            double E_mu = fmax(err_d / s_d, ec / s_c);
            if (E_mu <= kap_eps * mu && mu > mu_min) {
                mu = fmax(mu_min, fmin(kap_mu * mu, pow(mu, 1.5)));
                continue;
            }
            break;
        }
        double E0 = fmax(err_d / s_d, err_c0 / s_c);
        if (kkt_out) *kkt_out = E0;
        if (E0 <= o->tol) {
            status = 0;
            for (int k = 1; k < N; ++k)
                if (it->wj[k] >= 0 && it->zw[k] > 1e-6 * p->sf) status = 5;
            break;
        }
        if (o->tol < ACCEPTABLE_TOL) {
            n_acceptable = (E0 <= ACCEPTABLE_TOL) ? n_acceptable + 1 : 0;
            if (n_acceptable >= ACCEPTABLE_ITER) {
                status = 6;
                for (int k = 1; k < N; ++k)
                    if (it->wj[k] >= 0 && it->zw[k] > 1e-6 * p->sf) status = 7;
                break;
            }
        }
        if (iter == o->max_iter) break;
        if (g_stall_window > 0) {
            if (E0 < 0.5 * e_mark) {
                e_mark = E0;
                i_mark = iter;
            } else if (iter - i_mark >= g_stall_window) {
                status = 4;
                break;
            }
        }

        /* ---------------- backward (Riccati / DDP) sweep with the exact Lagrangian Hessian ------------------
         * If a control block Huu_k is not positive definite the sweep is repeated with the convex Gauss-Newton model
         * (no constraint curvature, radial part of the collision potential); should that fail numerically too, a
         * multiple of the identity is added. */
        double dV1 = 0.0, delta_w = reg;
        int nmod = 0, ok = 0, gn = 0, skipped_gn = 0;
        if (nonconvex_cost && prev_needed && iter % DW_PROBE != 0) {
            /* (xi) the last iteration needed the inertia ladder and the cost is non-convex by its weights: start where the
             * ladder would start, in the mode it ended in, instead of finding out again that delta_w = 0 does not do */
            delta_w = fmax(reg, (1.0 / 3.0) * dw_last);
            gn = prev_gn && !(mu <= GN_END_MU);
        }
        ++t_cnt_iter;
        for (int attempt = 0; attempt < 60 && !ok; ++attempt) {
            ++t_cnt_sweep;
            ok = 1;
            dV1 = 0.0;
            double Pxx[4][4], Pxp[4][2], Ppp[2][2] = {{0, 0}, {0, 0}}, px[4], pp[2] = {0, 0};
            memset(Pxx, 0, sizeof(Pxx));
            memset(Pxp, 0, sizeof(Pxp));
            for (int i = 0; i < 4; ++i) {
                double sl = it->x[N][i] - xlo_r(i), su = xhi_r(i) - it->x[N][i];
                Pxx[i][i] = (i >= p->i0 ? it->zxl[N][i] / sl + it->zxu[N][i] / su : 0.0) + delta_w;
                px[i] = (i >= p->i0) ? -mu / sl + mu / su : 0.0;
            }
            for (int k = N - 1; k >= 0; --k) {
                /* a stage whose control block is not positive definite with the exact Hessian is redone with the
                 * Gauss-Newton terms of that stage alone; only if that fails too the whole sweep is repeated */
                int lgn = gn;
            retry_stage:;
                const double rdk = (k >= 1) ? rd_full : 0.0;
                double Lxx[4][4], Lxu[4][2], Luu[2][2], lx[4], lu[2];
                memset(Lxx, 0, sizeof(Lxx));
                memset(Lxu, 0, sizeof(Lxu));
                memset(lx, 0, sizeof(lx));
                if (k >= 1) {
                    for (int i = 0; i < 4; ++i) {
                        for (int j = 0; j < 4; ++j) Lxx[i][j] = lgn ? Qgs[k][i][j] : Qs[k][i][j];
                        double sl = it->x[k][i] - xlo_r(i), su = xhi_r(i) - it->x[k][i];
                        Lxx[i][i] += (i >= p->i0 ? it->zxl[k][i] / sl + it->zxu[k][i] / su : 0.0) + delta_w;
                        lx[i] = lxs[k][i] + (i >= p->i0 ? -mu / sl + mu / su : 0.0);
                    }
                    if (it->wj[k] >= 0) {
                        const double sg = it->zw[k] / gw[k];
                        for (int i = 0; i < 2; ++i) {
                            for (int j = 0; j < 2; ++j) Lxx[i][j] += sg * wn[k][i] * wn[k][j];
                            if (!lgn) Lxx[i][i] -= 2.0 * it->zw[k];
                            lx[i] -= mu / gw[k] * wn[k][i];
                        }
                    }
                    if (!lgn) {
                        Lxx[2][2] += Wtt[k];
                        Lxx[2][3] += Wtv[k];
                        Lxx[3][2] += Wtv[k];
                        Lxu[2][1] = Wtd[k];
                        Lxu[3][1] = Wvd[k];
                    }
                }
                for (int i = 0; i < 2; ++i) {
                    double sl = it->u[k][i] - ulo_r(i), su = uhi_r(i) - it->u[k][i];
                    Luu[i][i] = rc + rdk + it->zul[k][i] / sl + it->zuu[k][i] / su + delta_w;
                    lu[i] = lus[k][i] - mu / sl + mu / su;
                }
                Luu[0][1] = Luu[1][0] = 0.0;
                if (!lgn) Luu[1][1] += Wdd[k];
                double PA[4][4], PB[4][2];
                for (int i = 0; i < 4; ++i) {
                    for (int j = 0; j < 4; ++j) {
                        double s = 0;
                        for (int m = 0; m < 4; ++m) s += Pxx[i][m] * A[k][m][j];
                        PA[i][j] = s;
                    }
                    for (int j = 0; j < 2; ++j) {
                        double s = 0;
                        for (int m = 0; m < 4; ++m) s += Pxx[i][m] * Bm[k][m][j];
                        PB[i][j] = s;
                    }
                }
                double Hxx[4][4], Hxu[4][2], Huu[2][2], hx[4], hu[2];
                for (int i = 0; i < 4; ++i) {
                    for (int j = 0; j < 4; ++j) {
                        double s = Lxx[i][j];
                        for (int m = 0; m < 4; ++m) s += A[k][m][i] * PA[m][j];
                        Hxx[i][j] = s;
                    }
                    for (int j = 0; j < 2; ++j) {
                        double s = Lxu[i][j];
                        for (int m = 0; m < 4; ++m) s += A[k][m][i] * (PB[m][j] + Pxp[m][j]);
                        Hxu[i][j] = s;
                    }
                    double s = lx[i];
                    for (int m = 0; m < 4; ++m) s += A[k][m][i] * px[m];
                    hx[i] = s;
                }
                for (int i = 0; i < 2; ++i) {
                    for (int j = 0; j < 2; ++j) {
                        double s = Luu[i][j] + Ppp[i][j];
                        for (int m = 0; m < 4; ++m)
                            s += Bm[k][m][i] * (PB[m][j] + Pxp[m][j]) + Pxp[m][i] * Bm[k][m][j];
                        Huu[i][j] = s;
                    }
                    double s = lu[i] + pp[i];
                    for (int m = 0; m < 4; ++m) s += Bm[k][m][i] * px[m];
                    hu[i] = s;
                }
                /* 2x2 control block: positive definite? */
                double ha = Huu[0][0], hb = 0.5 * (Huu[0][1] + Huu[1][0]), hc = Huu[1][1];
                if (!(ha > 0.0) || !(hc > 0.0) || !(ha * hc - hb * hb > 1e-12 * ha * hc)) {
                    if (!lgn) {
                        lgn = 1;
                        goto retry_stage;
                    }
                    ok = 0;
                    break;
                }
                double det = ha * hc - hb * hb;
                double Hi[2][2] = {{hc / det, -hb / det}, {-hb / det, ha / det}};
                for (int i = 0; i < 2; ++i) {
                    for (int j = 0; j < 4; ++j) Kx[k][i][j] = -(Hi[i][0] * Hxu[j][0] + Hi[i][1] * Hxu[j][1]);
                    for (int j = 0; j < 2; ++j) Kp[k][i][j] = rdk * Hi[i][j]; /* -Huu^-1 Hup with Hup = -rd I */
                    kf[k][i] = -(Hi[i][0] * hu[0] + Hi[i][1] * hu[1]);
                }
                dV1 += 0.5 * (kf[k][0] * hu[0] + kf[k][1] * hu[1]);
                double nPxx[4][4], nPpp[2][2], nppv[2];
                for (int i = 0; i < 4; ++i) {
                    for (int j = 0; j < 4; ++j) nPxx[i][j] = Hxx[i][j] + Hxu[i][0] * Kx[k][0][j] + Hxu[i][1] * Kx[k][1][j];
                    for (int j = 0; j < 2; ++j) Pxp[i][j] = Hxu[i][0] * Kp[k][0][j] + Hxu[i][1] * Kp[k][1][j];
                    px[i] = hx[i] + Hxu[i][0] * kf[k][0] + Hxu[i][1] * kf[k][1];
                }
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) Pxx[i][j] = 0.5 * (nPxx[i][j] + nPxx[j][i]);
                for (int i = 0; i < 2; ++i) {
                    for (int j = 0; j < 2; ++j) nPpp[i][j] = (i == j ? rdk : 0.0) - rdk * Kp[k][i][j];
                    nppv[i] = lps[k][i] - rdk * kf[k][i];
                }
                memcpy(Ppp, nPpp, sizeof(Ppp));
                memcpy(pp, nppv, sizeof(pp));
            }
            if (!ok) {
                /* exact Hessian not positive definite on the null space: use the convex Gauss-Newton model for this
                 * iteration; if even that is numerically singular, add a small multiple of the identity */
                ++nmod;
                if (gn_skip > 0) skipped_gn = 1;
                if (!gn && gn_skip == 0 && !(nonconvex_cost && mu <= GN_END_MU)) {
                    gn = 1;
                } else if (dw_last == 0.0) {
                    /* first inertia correction of this solve (IPOPT's algorithm IC: delta_w^0 = 1e-4, then x 100) */
                    delta_w = (delta_w < 1e-4) ? 1e-4 : 100.0 * delta_w;
                } else {
                    /* later ones start at a third of the value that worked last and grow by 8 (kappa_w^- = 1/3,
                     * kappa_w^+ = 8): the term stays within an order of magnitude of the curvature it has to cover */
                    const double third = (1.0 / 3.0) * dw_last;
                    delta_w = (delta_w < third) ? third : 8.0 * delta_w;
                }
                if (delta_w > 1e40) break;
            }
        }
        if (!ok) {
            status = 2;
            break;
        }
        if (delta_w > reg) dw_last = delta_w; /* the ladder was needed: remember where it ended */
        prev_needed = delta_w > reg;
        prev_gn = gn;
        if (nmod > 0) {
            if (gn_streak == 0) e_streak = E0;
            ++gn_streak;
            if (gn_skip > 0) {
                --gn_skip;
            } else if (gn_streak % GN_WATCH == 0) {
                if (E0 > 0.5 * e_streak) gn_skip = GN_SKIP;
                e_streak = E0;
            }
        } else {
            gn_streak = 0;
            gn_skip = 0;
        }

        /* ---------------- linear forward sweep: full primal-dual Newton step, step-length limits -------- */
        const double tau = fmax(0.99, 1.0 - mu);
        static _Thread_local double dxl[NMAX + 1][4], dul[NMAX][2];
        static _Thread_local double dzxl[NMAX + 1][4], dzxu[NMAX + 1][4], dzul[NMAX][2], dzuu[NMAX][2], dzw[NMAX + 1];
        /* The trial step lengths are 1, 1/4, 1/16, 1/64: they do not depend on this linearised step (anchoring the ladder at its
         * fraction-to-the-boundary length was measured no better: 17.96 against 17.50 iterations on BASELINE config 3), the
         * trials' own state-bound tests decide what is feasible - which lets the kernel run this recursion inside its
         * rollout loop.  The step is needed for the dual step and its length a_du. */
        const double a_pr = 1.0;
        double a_du = 1.0;
        memset(dxl[0], 0, sizeof(dxl[0]));
        for (int k = 0; k < N; ++k) {
            for (int i = 0; i < 2; ++i) {
                double s = kf[k][i];
                for (int j = 0; j < 4; ++j) s += Kx[k][i][j] * dxl[k][j];
                if (k >= 1)
                    for (int j = 0; j < 2; ++j) s += Kp[k][i][j] * dul[k - 1][j];
                dul[k][i] = s;
            }
            for (int i = 0; i < 4; ++i) {
                double s = 0.0;
                for (int j = 0; j < 4; ++j) s += A[k][i][j] * dxl[k][j];
                for (int j = 0; j < 2; ++j) s += Bm[k][i][j] * dul[k][j];
                dxl[k + 1][i] = s;
            }
            for (int i = 0; i < 2; ++i) {
                double sl = it->u[k][i] - ulo_r(i), su = uhi_r(i) - it->u[k][i], d = dul[k][i];
                dzul[k][i] = (mu - it->zul[k][i] * d) / sl - it->zul[k][i];
                dzuu[k][i] = (mu + it->zuu[k][i] * d) / su - it->zuu[k][i];
                if (dzul[k][i] < 0) a_du = fmin(a_du, -tau * it->zul[k][i] / dzul[k][i]);
                if (dzuu[k][i] < 0) a_du = fmin(a_du, -tau * it->zuu[k][i] / dzuu[k][i]);
            }
            for (int i = p->i0; i < 4; ++i) {
                double sl = it->x[k + 1][i] - xlo_r(i), su = xhi_r(i) - it->x[k + 1][i], d = dxl[k + 1][i];
                dzxl[k + 1][i] = (mu - it->zxl[k + 1][i] * d) / sl - it->zxl[k + 1][i];
                dzxu[k + 1][i] = (mu + it->zxu[k + 1][i] * d) / su - it->zxu[k + 1][i];
                if (dzxl[k + 1][i] < 0) a_du = fmin(a_du, -tau * it->zxl[k + 1][i] / dzxl[k + 1][i]);
                if (dzxu[k + 1][i] < 0) a_du = fmin(a_du, -tau * it->zxu[k + 1][i] / dzxu[k + 1][i]);
            }
            if (k + 1 < N && it->wj[k + 1] >= 0) {
                double d = wn[k + 1][0] * dxl[k + 1][0] + wn[k + 1][1] * dxl[k + 1][1];
                dzw[k + 1] = (mu - it->zw[k + 1] * d) / gw[k + 1] - it->zw[k + 1];
                if (dzw[k + 1] < 0) a_du = fmin(a_du, -tau * it->zw[k + 1] / dzw[k + 1]);
            }
        }

        /* ---------------- nonlinear rollout with feedback, Armijo on the barrier objective ------------- */
        const double PROJ_KEEP = (mu <= PROJ_END_MU) ? PROJ_KEEP_END : PROJ_KEEP0;
        double phi0 = barrier_objective(p, it, mu), alpha = a_pr, phi1 = phi0;
        int accepted = 0, nls = 0;
        int cross[NMAX + 1]; /* vehicle that a rejected trial took across d = 1 inwards (smallest slack), per node */
        for (int k = 0; k <= N; ++k) cross[k] = -1;
        for (nls = 0; nls < MAXLS; ++nls, alpha *= BTF) {
            int feas = 1;
            ++t_cnt_roll;
            trial = *it;
            for (int k = 0; k < N && feas; ++k) {
                double dmax = 0.0; /* size of the linearised step that reaches this stage */
                for (int j = 0; j < 4; ++j) dmax = fmax(dmax, fabs(dxl[k][j]));
                if (k >= 1)
                    for (int j = 0; j < 2; ++j) dmax = fmax(dmax, fabs(dul[k - 1][j]));
                for (int i = 0; i < 2; ++i) {
                    double s = alpha * kf[k][i];
                    if (dmax < OPEN_LOOP_STEP) {
                        s = alpha * dul[k][i];
                    } else {
                        /* the order the kernel's row-cooperative rollout adds the terms in (what arrives last on its
                         * dependency chain - the positions - is added last): kf, Kp, Kx[v], Kx[theta], Kx[x], Kx[y] */
                        static const int order[4] = {3, 2, 0, 1};
                        if (k >= 1)
                            for (int j = 0; j < 2; ++j) s += Kp[k][i][j] * (trial.u[k - 1][j] - it->u[k - 1][j]);
                        for (int jj = 0; jj < 4; ++jj) s += Kx[k][i][order[jj]] * (trial.x[k][order[jj]] - it->x[k][order[jj]]);
                    }
                    trial.u[k][i] = it->u[k][i] + s;
                    /* control bounds: clamp each component to the fraction-to-the-boundary box instead of
                     * shortening the whole step (saturated accelerations would otherwise jam every iteration) */
                    double lo_c = ulo_r(i) + fmax((1.0 - tau) * (it->u[k][i] - ulo_r(i)), MIN_SLACK);
                    double hi_c = uhi_r(i) - fmax((1.0 - tau) * (uhi_r(i) - it->u[k][i]), MIN_SLACK);
                    trial.u[k][i] = fmin(fmax(trial.u[k][i], lo_c), hi_c);
                }
                {
                    /* v of node k+1 is decided by a_k alone: keep it inside the node's box */
                    const double *xo = it->x[k + 1];
                    double vlo = xlo_r(3) + PROJ_KEEP * (xo[3] - xlo_r(3)), vhi = xhi_r(3) - PROJ_KEEP * (xhi_r(3) - xo[3]);
                    double a = fmin(fmax(trial.u[k][0], (vlo - trial.x[k][3]) / dt), (vhi - trial.x[k][3]) / dt);
                    double lo_c = ulo_r(0) + fmax((1.0 - tau) * (it->u[k][0] - ulo_r(0)), MIN_SLACK),
                           hi_c = uhi_r(0) - fmax((1.0 - tau) * (uhi_r(0) - it->u[k][0]), MIN_SLACK);
                    trial.u[k][0] = fmin(fmax(a, lo_c), hi_c);
                }
                dyn_t d;
                dyn_eval(trial.x[k], trial.u[k], &d);
                {
                    /* theta of node k+1 is decided by delta_k alone (theta + dt v/L sin beta(delta)): if it leaves the
                     * node's box, take the delta that puts it on the edge of the box */
                    const double *xo = it->x[k + 1];
                    const double tlo = xlo_r(2) + PROJ_KEEP * (xo[2] - xlo_r(2)), thi = xhi_r(2) - PROJ_KEEP * (xhi_r(2) - xo[2]);
                    const double th1 = trial.x[k][2] + dt * d.f[2], vk = trial.x[k][3];
                    if ((th1 < tlo || th1 > thi) && vk > 1e-6) {
                        double sreq = ((th1 < tlo ? tlo : thi) - trial.x[k][2]) * WHEELBASE / (dt * vk);
                        if (fabs(sreq) < 0.9) {
                            double del = atan(2.0 * sreq / sqrt(1.0 - sreq * sreq));
                            double lo_d = ulo_r(1) + fmax((1.0 - tau) * (it->u[k][1] - ulo_r(1)), MIN_SLACK),
                                   hi_d = uhi_r(1) - fmax((1.0 - tau) * (uhi_r(1) - it->u[k][1]), MIN_SLACK);
                            trial.u[k][1] = fmin(fmax(del, lo_d), hi_d);
                            dyn_eval(trial.x[k], trial.u[k], &d);
                        }
                    }
                }
                for (int i = 0; i < 4; ++i) {
                    trial.x[k + 1][i] = trial.x[k][i] + dt * d.f[i];
                    if (i < p->i0) continue;
                    if (trial.x[k + 1][i] - xlo_r(i) < fmax(0.5 * (1.0 - tau) * (it->x[k + 1][i] - xlo_r(i)), MIN_SLACK) ||
                        xhi_r(i) - trial.x[k + 1][i] < fmax(0.5 * (1.0 - tau) * (xhi_r(i) - it->x[k + 1][i]), MIN_SLACK))
                        feas = 0;
                }
                if (k + 1 < N && it->wj[k + 1] >= 0 &&
                    wall_slack(p, k + 1, trial.x[k + 1], it->wj[k + 1]) < 0.5 * (1.0 - tau) * gw[k + 1])
                    feas = 0;
            }
            if (g_trace2) fprintf(stderr, "   trial %d alpha %.3e feas %d\n", nls, alpha, feas);
            if (!feas) continue;
            phi1 = barrier_objective(p, &trial, mu);
            if (g_trace2) fprintf(stderr, "      phi1-phi0 %.6e need %.6e\n", phi1 - phi0, 1e-4 * alpha * 2.0 * dV1);
            if (phi1 <= phi0 + 1e-4 * alpha * 2.0 * dV1 + 1e-12 * fabs(phi0)) {
                accepted = 1;
                break;
            }
            if (p->cc)
                for (int k = 1; k < N; ++k)
                    for (int j = 0; j < p->V; ++j) {
                        double g0 = wall_slack(p, k, it->x[k], j);
                        if (g0 > 0.0 && wall_slack(p, k, trial.x[k], j) < 0.0 &&
                            (cross[k] < 0 || g0 < wall_slack(p, k, it->x[k], cross[k])))
                            cross[k] = j;
                    }
        }
        if (g_trace)
            fprintf(stderr, "   Ed at stage %d control %d: u %.12g zl %.3e zu %.3e wall %d zw %.3e\n", dbg_k, dbg_i, it->u[dbg_k][dbg_i],
                    it->zul[dbg_k][dbg_i], it->zuu[dbg_k][dbg_i], it->wj[dbg_k], it->wj[dbg_k] >= 0 ? it->zw[dbg_k] : 0.0);
        if (g_trace)
            fprintf(stderr, "it %3d nmod %d gn %d mu %.2e dw %.1e Ed %.3e Ec %.3e E0 %.3e a_pr %.3e alpha %.3e a_du %.3e nls %d dV1 %.3e phi0 %.8e phi1 %.8e acc %d\n",
                    iter, nmod, gn, mu, delta_w, err_d, err_c0, E0, a_pr, alpha, a_du, nls, dV1, phi0, phi1, accepted);
        if (!accepted) trial = *it; /* keep the primal point; the dual step below still moves z */
        if (accepted && nls == 0) {
            reg = reg / REG_FACTOR;
            if (reg < REG_MIN) reg = 0.0;
        } else if (!accepted || nls >= 2) {
            reg = (reg == 0.0) ? REG_MIN : fmin(REG_FACTOR * reg, REG_MAX);
        }
        nfail = accepted ? 0 : nfail + 1;
        if (!accepted && skipped_gn) {
            /* (vii) the step of the regularised exact Hessian was not acceptable at any length: the next iteration starts with
             * the Gauss-Newton fallback again, and this one does not count towards the line-search-failure exit */
            gn_skip = 0;
            --nfail;
        }
        /* ---------------- dual step: multipliers that shrink share one fraction-to-the-boundary length, multipliers
         *                  that grow (no positivity issue) take the full Newton step ---------------- */
        for (int k = 1; k <= N; ++k)
            for (int i = p->i0; i < 4; ++i) {
                double sln = trial.x[k][i] - xlo_r(i), sun = xhi_r(i) - trial.x[k][i];
                double zl = it->zxl[k][i] + (dzxl[k][i] > 0 ? 1.0 : a_du) * dzxl[k][i], zu = it->zxu[k][i] + (dzxu[k][i] > 0 ? 1.0 : a_du) * dzxu[k][i];
                trial.zxl[k][i] = fmax(fmin(zl, KSIG * mu / sln), mu / (KSIG * sln));
                trial.zxu[k][i] = fmax(fmin(zu, KSIG * mu / sun), mu / (KSIG * sun));
            }
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < 2; ++i) {
                double sln = trial.u[k][i] - ulo_r(i), sun = uhi_r(i) - trial.u[k][i];
                double zl = it->zul[k][i] + (dzul[k][i] > 0 ? 1.0 : a_du) * dzul[k][i], zu = it->zuu[k][i] + (dzuu[k][i] > 0 ? 1.0 : a_du) * dzuu[k][i];
                trial.zul[k][i] = fmax(fmin(zl, KSIG * mu / sln), mu / (KSIG * sln));
                trial.zuu[k][i] = fmax(fmin(zu, KSIG * mu / sun), mu / (KSIG * sun));
            }
        int newwall = 0;
        for (int k = 1; k < N; ++k) {
            if (it->wj[k] >= 0) {
                double gn = wall_slack(p, k, trial.x[k], it->wj[k]);
                double z = it->zw[k] + (dzw[k] > 0 ? 1.0 : a_du) * dzw[k];
                trial.zw[k] = fmax(fmin(z, KSIG * mu / gn), mu / (KSIG * gn));
            }
            if (cross[k] >= 0 && cross[k] != it->wj[k]) {
                double gc = wall_slack(p, k, trial.x[k], cross[k]);
                if (gc > 0.0 && (it->wj[k] < 0 || gc < wall_slack(p, k, trial.x[k], it->wj[k]))) {
                    trial.wj[k] = cross[k];
                    trial.zw[k] = mu / gc;
                    newwall = 1;
                }
            }
        }
        if (newwall && !accepted) nfail = 0;
        *it = trial;
        /* three consecutive iterations without an acceptable step: the primal point cannot move any more (a kink of
         * the collision cost at d = 1, or numerical stationarity) - stop instead of burning the iteration budget */
        if (nfail >= 3) {
            status = 4;
            ++iter;
            break;
        }
    }
    for (int k = 1; k <= N; ++k)
        for (int i = 0; i < 4; ++i) it->lam[k][i] = -yv[k][i] / p->sf;
    *iters_out = iter;
    return status;
}

/* ------------------------------------------------------------------------------------------
 * batch entry point (same argument meaning as include/mpc_mi355x.h : mpc_solve_batch)
 *   ref_table [M,4] x,y,v,heading; state [B,4]; ego_index [B]; vref [B,N+1] or NULL (table speeds);
 *   weights [B,3] speed,control,input_diff; is_collide [B]; others [B,V,4] x,y,speed,heading.
 *   flags bit0: collision-cost term on; bit1: drop the (never active) |x|,|y| <= 500 bounds, as the GPU kernel does.
 * outputs: u0 [B,2]; U [B,N,2] / X [B,N+1,4] / lam [B,N+1,4] optional (NULL to skip);
 *   status [B]; iters [B]; kkt [B] optional.
 * ------------------------------------------------------------------------------------------ */
int oracle_solve_batch_warm(int B, int N, double dt, const double *ref_table, int M, const double *state,
                            const int32_t *ego_index, const double *vref, const double *weights,
                            const uint8_t *is_collide, const double *others, int V, uint32_t flags,
                            double w_distance, double w_collision, double tol, int max_iter, const double *u_init,
                            double *u0, double *U, double *X, double *lam, int32_t *status, int32_t *iters,
                            double *kkt, int nthreads);

int oracle_solve_batch(int B, int N, double dt, const double *ref_table, int M, const double *state,
                       const int32_t *ego_index, const double *vref, const double *weights,
                       const uint8_t *is_collide, const double *others, int V, uint32_t flags,
                       double w_distance, double w_collision, double tol, int max_iter, double *u0,
                       double *U, double *X, double *lam, int32_t *status, int32_t *iters, double *kkt,
                       int nthreads) {
    return oracle_solve_batch_warm(B, N, dt, ref_table, M, state, ego_index, vref, weights, is_collide, others, V, flags,
                                   w_distance, w_collision, tol, max_iter, NULL, u0, U, X, lam, status, iters, kkt,
                                   nthreads);
}

/* same with initial controls u_init [B,N,2] (NULL = cold start of the reference) */
int oracle_solve_batch_warm(int B, int N, double dt, const double *ref_table, int M, const double *state,
                            const int32_t *ego_index, const double *vref, const double *weights,
                            const uint8_t *is_collide, const double *others, int V, uint32_t flags,
                            double w_distance, double w_collision, double tol, int max_iter, const double *u_init,
                            double *u0, double *U, double *X, double *lam, int32_t *status, int32_t *iters,
                            double *kkt, int nthreads) {
    if (N < 1 || N > NMAX || V < 0 || V > VMAX || M < 1) return -1;
    g_cnt_iter = g_cnt_sweep = g_cnt_roll = 0;
    g_cnt_solve = B;
    g_cnt_N = N;
    g_cnt_V = (flags & 1u) ? V : 0;
    g_cnt_cc = (flags & 1u) ? 1 : 0;
    opts_t o = {tol, g_exp_mu_init, max_iter};
    g_trace = getenv("ORACLE_TRACE") != NULL;
    g_trace2 = getenv("ORACLE_TRACE2") != NULL;
    long c_iter = 0, c_sweep = 0, c_roll = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_iter, c_sweep, c_roll) proc_bind(close)
    for (int b = 0; b < B; ++b) {
        t_cnt_iter = t_cnt_sweep = t_cnt_roll = 0;
        prob_t p;
        memset(&p, 0, sizeof(p));
        p.N = N;
        p.dt = dt;
        p.V = (flags & 1u) ? V : 0;
        p.cc = (flags & 1u) ? 1 : 0;
        for (int i = 0; i < 4; ++i) p.x0[i] = state[4 * b + i];
        for (int k = 0; k <= N; ++k) {
            int idx = ego_index[b] + k;
            if (idx > M - 1) idx = M - 1;
            if (idx < 0) idx = 0;
            p.rx[k] = ref_table[4 * idx + 0];
            p.ry[k] = ref_table[4 * idx + 1];
            p.rv[k] = vref ? vref[(size_t)b * (N + 1) + k] : ref_table[4 * idx + 2];
            p.rh[k] = ref_table[4 * idx + 3];
            p.rs[k] = sin(p.rh[k]);
            p.rc[k] = cos(p.rh[k]);
        }
        p.ws = is_collide[b] ? 100.0 : weights[3 * b + 0];
        p.wc = weights[3 * b + 1];
        p.wd = weights[3 * b + 2];
        for (int j = 0; j < p.V; ++j) {
            const double *ov = others + ((size_t)b * V + j) * 4;
            p.ox[j] = ov[0];
            p.oy[j] = ov[1];
            p.osx[j] = ov[2] * dt * cos(ov[3]);
            p.osy[j] = ov[2] * dt * sin(ov[3]);
        }
        p.i0 = (flags & 2u) ? 2 : 0;
        p.wdist = w_distance;
        p.wcoll = (p.cc && is_collide[b]) ? 3000.0 * w_collision : 0.0;
        iter_t *it = (iter_t *)malloc(sizeof(iter_t));
        int its = 0;
        double e = 0.0;
        int st = solve_one(&p, &o, it, u_init ? u_init + (size_t)b * N * 2 : NULL, &its, &e);
        u0[2 * b + 0] = it->u[0][0];
        u0[2 * b + 1] = it->u[0][1];
        if (U)
            for (int k = 0; k < N; ++k) {
                U[((size_t)b * N + k) * 2 + 0] = it->u[k][0];
                U[((size_t)b * N + k) * 2 + 1] = it->u[k][1];
            }
        if (X)
            for (int k = 0; k <= N; ++k)
                for (int i = 0; i < 4; ++i) X[((size_t)b * (N + 1) + k) * 4 + i] = it->x[k][i];
        if (lam)
            for (int k = 0; k <= N; ++k)
                for (int i = 0; i < 4; ++i) lam[((size_t)b * (N + 1) + k) * 4 + i] = it->lam[k][i];
        status[b] = st;
        iters[b] = its;
        if (kkt) kkt[b] = e;
        free(it);
        c_iter += t_cnt_iter;
        c_sweep += t_cnt_sweep;
        c_roll += t_cnt_roll;
    }
    g_cnt_iter = c_iter;
    g_cnt_sweep = c_sweep;
    g_cnt_roll = c_roll;
    if (getenv("ORACLE_COUNT"))
        fprintf(stderr, "oracle work: %ld iterations, %ld backward sweeps, %ld rollouts\n", g_cnt_iter, g_cnt_sweep, g_cnt_roll);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Work of the last oracle_solve_batch* call ("the CPU oracle must carry an exact flop counter so
 * the figure is measured, not estimated").  Counted exactly: interior-point iterations, backward sweeps (an iteration
 * repeats its sweep when a fallback is needed) and line-search rollouts, over all instances.  Converted to floating
 * point operations with the per-stage operation counts of the statements of solve_one above (+ - * / sqrt fmin fmax
 * fabs and comparisons count 1; sin cos tan atan log are counted apart as transcendentals):
 *   derivatives of a stage   dyn_eval 18 + A, B entries 22 + tracking cost/gradient/Hessian 41 + control terms 8
 *                            (+ per vehicle 53 + 6 with the collision cost), 6 transcendentals
 *   adjoint + dual residual  22 + 8 + 40 + curvature terms 18 + 40, 6 transcendentals (dyn_eval again)
 *   complementarity          50
 *   backward sweep stage     820 (PA 128, PB 64, Hxx 128, Hxu 96, hx 32, Huu 84, hu 18, stage Hessian 52, 2x2 and gains
 *                            58, value function 160)
 *   linear forward stage     128
 *   rollout stage            113 + barrier objective 29 (+ 13 per vehicle + 3), 6 + 8 transcendentals
 *   dual update stage        64
 * out[0] = iterations, out[1] = sweeps, out[2] = rollouts, out[3] = solves, out[4] = arithmetic flops,
 * out[5] = transcendentals.
 * ------------------------------------------------------------------------------------------ */
/* the engine's optional progress guard (include/mpc_mi355x.h: mpc_config.stall_window); applies to the calls that follow */
void oracle_set_stall_window(int w) { g_stall_window = w > 0 ? w : 0; }

/* tools/portfolio_study.py only: variants of the globalisation (what a portfolio of solver settings would race) */
void oracle_set_experiment(double mu_init, double btf) {
    g_exp_mu_init = mu_init > 0.0 ? mu_init : 0.1;
    g_exp_btf = (btf > 0.0 && btf < 1.0) ? btf : 0.25;
}

/* warm-start study (tools/warm_start_study.py): 0 = off */
void oracle_set_warm_experiment(double warm_mu) { g_exp_warm_mu = warm_mu > 0.0 ? warm_mu : 0.0; }

void oracle_last_work(double out[6]) {
    const double N = g_cnt_N, V = g_cnt_V, cc = g_cnt_cc;
    const double per_iter = N * ((18 + 22 + 41 + 8 + cc * (V * 53 + 6)) + (22 + 8 + 40 + 18 + 40) + 50 + 128 + 64);
    const double per_iter_t = N * (6 + 6);
    const double per_sweep = N * 820;
    const double per_roll = N * (113 + 29 + cc * (V * 13 + 3));
    const double per_roll_t = N * (6 + 8);
    const double start = N * (26 + 27 + cc * V * 23), start_t = N * 6;   /* first rollout + objective scaling */
    out[0] = (double)g_cnt_iter;
    out[1] = (double)g_cnt_sweep;
    out[2] = (double)g_cnt_roll;
    out[3] = (double)g_cnt_solve;
    out[4] = g_cnt_solve * start + g_cnt_iter * per_iter + g_cnt_sweep * per_sweep + g_cnt_roll * per_roll;
    out[5] = g_cnt_solve * start_t + g_cnt_iter * per_iter_t + g_cnt_roll * per_roll_t;
}
