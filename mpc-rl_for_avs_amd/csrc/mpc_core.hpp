// mpc_core.hpp - per-instance interior-point DDP solver, written for one GPU lane per MPC instance.
//
// This is the arithmetic of the hot path: the NLP of PureMPC_Agent._solve (reference
// agents/pure_mpc.py:80-318) solved by a primal-dual interior-point method whose Newton systems are
// factorised stage by stage (Riccati / DDP backward sweep) and whose iterates are kept dynamically
// feasible by nonlinear feedback rollouts.  One wave64 lane owns one instance; every per-stage quantity
// lives in a structure-of-arrays workspace indexed [slot][stage][instance] (LDS on the GPU), so the lanes of
// a wave touch consecutive doubles.  All small matrices are scalarised (no runtime-indexed arrays, which
// hipcc would spill to scratch) and the sparsity of the bicycle-model Jacobians
//     A = I + dt*df/dx = [1 0 a02 a03; 0 1 a12 a13; 0 0 1 a23; 0 0 0 1],   B = [0 b01; 0 b11; 0 b21; dt 0]
// is exploited by hand.
//
// The same header compiles for the host (tests/cpu_core_harness.cpp) so the kernel logic can be run
// under sanitizers without a GPU; the product never takes that path.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif

namespace mpc {

// ---------------------------------------------------------------------------------------------------
// workspace slot map (doubles per stage).  The map is kept small on purpose: 46 slots (54 with the
// collision-cost term) x (N+1) stages x 8 B = 7.7 KB (9.1 KB) per instance at N = 20, so that 16 instances
// fit the 160 KB LDS of one CU.  What is cheap to recompute is not stored: the reference geometry comes
// from the shared path table by index, beta'/beta'' from sin/cos(beta), the quadratic tracking cost and its
// derivatives from the state (only the collision-cost variant caches them), the linearised step is
// parked in the adjoint slots for the dual update, and the never-active |x|,|y| <= 500 bounds carry no multipliers.
// Two trajectory buffers (current / trial) are swapped on accept.
// ---------------------------------------------------------------------------------------------------
enum : int {
    B_X = 0,    // 4  x, y, theta, v                       (node k)
    B_U = 4,    // 2  a, delta                             (node k < N)
    B_DYN = 6,  // 4  sin/cos(theta+beta), sin/cos(beta)   of (x_k, u_k)
    BUF_SLOTS = 10,
    S_BUF0 = 0,
    S_BUF1 = BUF_SLOTS,
    S_ZXL = 2 * BUF_SLOTS,  // 2 lower-bound multipliers of theta_k, v_k
    S_ZXU = S_ZXL + 2,      // 2
    S_ZUL = S_ZXU + 2,      // 2
    S_ZUU = S_ZUL + 2,      // 2
    S_Y = S_ZUU + 2,        // 4 adjoint dL/dx_k
    S_KX = S_Y + 4,         // 8 feedback gain on dx (row-major 2x4)
    S_KP = S_KX + 8,        // 3 feedback gain on the previous control (symmetric 2x2: 00 01 11)
    S_KF = S_KP + 3,        // 2 feed-forward
    S_RV = S_KF + 2,        // 1 reference speed of stage k
    STAGE_SLOTS = S_RV + 1,  // 46
    // collision-cost variant only: cached stage-cost derivatives of the last completed rollout
    S_LX = STAGE_SLOTS,         // 2 scaled gradient of the distance potential wrt x, y
    S_Q = S_LX + 2,             // 3 its scaled Hessian q00 q01 q11
    S_QG = S_Q + 3,             // 3 convex (radial) part of q00 q01 q11
    STAGE_SLOTS_CC = S_QG + 3   // 54
};

// reference-table columns served by WS::ref(k, c)
enum : int { R_X = 0, R_Y = 1, R_H = 2, R_SIN = 3, R_COS = 4, REF_COLS = 5 };

struct SolveParams {
    int N;         // horizon
    int V;         // other vehicles used by the collision-cost term (0 when the term is off)
    int max_iter;
    double dt;
    double tol;
    double mu_init;
    double w_distance;  // weight_distance
};

// single v_max_f64 / v_min_f64 on the device (a compare + two v_cndmask otherwise)
MPC_HD double fmax2(double a, double b) { return __builtin_fmax(a, b); }
MPC_HD double fmin2(double a, double b) { return __builtin_fmin(a, b); }

// bounds of the reference NLP (agents/pure_mpc.py:272-280), relaxed like IPOPT's bound_relax_factor 1e-8.
// State bounds: index 0 = theta in [-pi, pi], 1 = v in [0, 30]  (|x|,|y| <= 500 can never be active here).
#define MPC_PI 3.14159265358979323846
MPC_HD double xlo_r(int i) { return i == 0 ? -MPC_PI - 1e-8 * MPC_PI : -1e-8; }
MPC_HD double xhi_r(int i) { return i == 0 ? MPC_PI + 1e-8 * MPC_PI : 30.0 + 30e-8; }
MPC_HD double ulo_r(int i) { return i == 0 ? -5.0 - 5e-8 : -(MPC_PI / 3.0) - 1e-8 * (MPC_PI / 3.0); }
MPC_HD double uhi_r(int i) { return i == 0 ? 5.0 + 5e-8 : (MPC_PI / 3.0) + 1e-8 * (MPC_PI / 3.0); }

constexpr double kInvWheelbase = 1.0 / 2.5;  // Vehicle.LENGTH, agents/utils.py:18

// warm start (opt-in, not in the reference): given initial controls are moved 0.1 % of the range inside their bounds
MPC_HD double warm_clamp(double u, int i) {
    const double lo = i == 0 ? -5.0 : -(MPC_PI / 3.0), hi = -lo, m = 1e-3 * (hi - lo);
    return fmin2(fmax2(u, lo + m), hi - m);
}

// ---- lean FP64 math -------------------------------------------------------------------------------
// The solver is bound by instruction issue / instruction-cache footprint, not by memory, so the generic
// libm expansions (large-argument trig reduction, IEEE division with denormal fix-ups) are replaced by
// bounded-range versions.  All arguments here are bounded by the NLP itself: |theta| <= pi(1+1e-8),
// |delta| <= pi/3(1+1e-8), slacks and determinants are positive normal numbers.
MPC_HD double frcp(double x) {  // 1/x for positive normal x
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rcp_f64 is good to 2^-24 (measured, tools/ubench/fp64_latency.hip): one third-order step gives 2^-72
    const double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, fma(e, e, e), y);
#else
    return 1.0 / x;
#endif
}
MPC_HD double frsqrt(double x) {  // 1/sqrt(x) for positive normal x
#if defined(__HIP_DEVICE_COMPILE__)
    // v_rsq_f64 is good to 2^-24 (measured): one third-order step gives 2^-72
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);     // 1 - x y^2
    return fma(y * e, fma(0.375, e, 0.5), y);
#else
    return 1.0 / sqrt(x);
#endif
}
// sin and cos of |x| <= ~2 pi: two-constant Cody-Waite reduction to |r| <= pi/4, fdlibm kernel polynomials
MPC_HD void sincos_b(double x, double &s, double &c) {
    const double n = rint(x * 6.36619772367581382433e-01);  // x * 2/pi
    double r = fma(-n, 1.57079632679489655800e+00, x);
    r = fma(-n, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
                                              2.75573137070700676789e-06),
                                       -1.98412698298579493134e-04),
                                8.33333333332248946124e-03),
                         -1.66666666666666324348e-01);
    const double sr = fma(z * r, ps, r);
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
                                              -2.75573143513906633035e-07),
                                       2.48015872894767294178e-05),
                                -1.38888888888741095749e-03),
                         4.16666666666666019037e-02);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = ((int)n) & 3;
    const double sa = (q & 1) ? cr : sr, ca = (q & 1) ? sr : cr;
    s = (q & 2) ? -sa : sa;
    c = ((q + 1) & 2) ? -ca : ca;
}

// kinematic bicycle model (agents/pure_mpc.py:220-228): beta = atan(LENGTH_REAR/LENGTH * tan(delta)).
// tan and atan are eliminated algebraically: with q = (4 cos^2 delta + sin^2 delta)^-1/2,
//   cos(beta) = 2 cos(delta) q,  sin(beta) = sin(delta) q,  sin/cos(theta+beta) by the addition theorems.
MPC_HD void dyn_eval(double theta, double delta, double &S, double &C, double &sb, double &cb) {
    double sd, cd, st, ct;
    sincos_b(delta, sd, cd);
    sincos_b(theta, st, ct);
    const double q = frsqrt(fma(3.0 * cd, cd, 1.0));
    cb = 2.0 * cd * q;
    sb = sd * q;
    S = fma(st, cb, ct * sb);
    C = fma(ct, cb, -st * sb);
}
// beta' = 2 q^2 and beta'' = 12 sin(delta) cos(delta) q^4 expressed through sin/cos(beta): q^2 = sb^2 + cb^2/4
MPC_HD void beta_derivs(double sb, double cb, double &bp, double &bpp) {
    const double q2 = fma(sb, sb, 0.25 * cb * cb);
    bp = 2.0 * q2;
    bpp = 6.0 * sb * cb * q2;
}

// Quadratic tracking part of the stage cost at node k (agents/pure_mpc.py:134-156, multiplier 10 of :206):
//   10*[4 perp^2 + 2 para^2 + ws (v-vref)^2 + .5 (theta-h)^2],  returned UNSCALED; g = gradient (x,y,theta,v)
template <class WS>
MPC_HD double track_cost(const WS &w, int k, double ws_, double x0, double x1, double x2, double x3, double *g) {
    const double s = w.ref(k, R_SIN), c = w.ref(k, R_COS);
    const double dx = x0 - w.ref(k, R_X), dy = x1 - w.ref(k, R_Y);
    const double perp = dx * s - dy * c, para = dx * c + dy * s;
    const double dv = x3 - w.ld(S_RV, k), dth = x2 - w.ref(k, R_H);
    if (g) {
        g[0] = 10.0 * (8.0 * perp * s + 4.0 * para * c);
        g[1] = 10.0 * (-8.0 * perp * c + 4.0 * para * s);
        g[2] = 10.0 * dth;
        g[3] = 20.0 * ws_ * dv;
    }
    return 10.0 * (4.0 * perp * perp + 2.0 * para * para + ws_ * dv * dv + 0.5 * dth * dth);
}
template <class WS>
MPC_HD void track_hess(const WS &w, int k, double &h00, double &h01, double &h11) {
    const double s = w.ref(k, R_SIN), c = w.ref(k, R_COS);
    h00 = 10.0 * (8.0 * s * s + 4.0 * c * c);
    h01 = 10.0 * (-8.0 * s * c + 4.0 * c * s);
    h11 = 10.0 * (8.0 * c * c + 4.0 * s * s);
}

// Distance potential of the optional collision-cost term (agents/archive/pure_mpc.py:189-196):
//   w_distance * sum_j (d<1 ? 1000 : 100)/(d+1e-6)^2, d = |p - (p_j + k*step_j)|;  UNSCALED.
// d[0..1] gradient, d[2..4] Hessian 00 01 11, d[5..7] its convex (radial) part.
template <class WS>
MPC_HD double dist_cost(const SolveParams &P, const WS &w, int k, double x0, double x1, double *d8) {
    double J = 0.0, g0 = 0, g1 = 0, h00 = 0, h01 = 0, h11 = 0, c00 = 0, c01 = 0, c11 = 0;
    for (int j = 0; j < P.V; ++j) {
        const double px = x0 - (w.oth(j, 0) + k * w.oth(j, 2));
        const double py = x1 - (w.oth(j, 1) + k * w.oth(j, 3));
        const double d2 = fma(px, px, py * py);
        const double rd = frsqrt(d2), d = d2 * rd;
        const double cst = (d < 1.0 ? 1000.0 : 100.0) * P.w_distance;
        const double rde = frcp(d + 1e-6);
        const double inv2 = rde * rde;
        J += cst * inv2;
        if (d8) {
            const double dpsi = -2.0 * cst * inv2 * rde;
            const double nx = px * rd, ny = py * rd;
            const double d2psi = 6.0 * cst * inv2 * inv2;
            const double tt = dpsi * rd;
            g0 += dpsi * nx;
            g1 += dpsi * ny;
            h00 += d2psi * nx * nx + tt * (1.0 - nx * nx);
            h01 += (d2psi - tt) * nx * ny;
            h11 += d2psi * ny * ny + tt * (1.0 - ny * ny);
            c00 += d2psi * nx * nx;
            c01 += d2psi * nx * ny;
            c11 += d2psi * ny * ny;
        }
    }
    if (d8) {
        d8[0] = g0; d8[1] = g1; d8[2] = h00; d8[3] = h01; d8[4] = h11; d8[5] = c00; d8[6] = c01; d8[7] = c11;
    }
    return J;
}

// collision-cost variant: derivatives of the distance potential at node k of trajectory buffer CB -> cache
template <class WS>
MPC_HD void cache_dist_derivs(const SolveParams &P, WS &w, int CB, int k, double sf) {
    double d8[8];
    dist_cost(P, w, k, w.ld(CB + B_X + 0, k), w.ld(CB + B_X + 1, k), d8);
    w.st(S_LX + 0, k, sf * d8[0]);
    w.st(S_LX + 1, k, sf * d8[1]);
    w.st(S_Q + 0, k, sf * d8[2]);
    w.st(S_Q + 1, k, sf * d8[3]);
    w.st(S_Q + 2, k, sf * d8[4]);
    w.st(S_QG + 0, k, sf * d8[5]);
    w.st(S_QG + 1, k, sf * d8[6]);
    w.st(S_QG + 2, k, sf * d8[7]);
}

// scaled stage-cost gradient at node k (1 <= k < N) of the current trajectory
template <bool CC, class WS>
MPC_HD void cost_grad(const WS &w, int CB, int k, double sf, double ws_, double wcoll, double *lx) {
    const double x0 = w.ld(CB + B_X + 0, k), x1 = w.ld(CB + B_X + 1, k), x2 = w.ld(CB + B_X + 2, k),
                 x3 = w.ld(CB + B_X + 3, k);
    double g[4];
    track_cost(w, k, ws_, x0, x1, x2, x3, g);
    lx[0] = sf * g[0];
    lx[1] = sf * g[1];
    lx[2] = sf * g[2];
    lx[3] = sf * g[3];
    if (CC) {
        lx[0] += w.ld(S_LX + 0, k);
        lx[1] += w.ld(S_LX + 1, k);
        lx[3] += sf * 2.0 * wcoll * x3;
    }
}

// One forward rollout from x0 with controls  u_k = ucur_k + alpha*kf_k + Kx_k (x_k - xcur_k) + Kp_k (u_{k-1} - ucur_{k-1})
// written into buffer `tb` (reading the current iterate from buffer `cb`); with first==true the controls of
// `tb` are taken as they are (cold start).  Returns false when a bound would be crossed
// (fraction-to-the-boundary rule with parameter `frac`).  J / bar receive the scaled objective and the
// log-barrier sum of the new trajectory; STORE = false evaluates a trial without writing anything (used by the
// replica lanes of the parallel line search).  (Trial rollouts need only the VALUE of the distance potential; its
// derivatives are computed once per iteration, in the adjoint sweep of the accepted trajectory.)
template <bool CC, bool STORE, class WS>
MPC_HD bool rollout(const SolveParams &P, WS &w, int cb, int tb, bool first, double alpha, double frac, double sf,
                    double ws_, double wc_, double wd_, double wcoll, const double *x0, double &Jout, double &barout) {
    const int N = WS::kN > 0 ? WS::kN : P.N;  // compile-time horizon turns LDS offsets into immediates
    const double dt = P.dt;
    const int CB = cb * BUF_SLOTS, TB = tb * BUF_SLOTS;
    double x_0 = x0[0], x_1 = x0[1], x_2 = x0[2], x_3 = x0[3];
    double J = 0.0, bar = 0.0, slack_acc = 1.0;
    double up0 = 0.0, up1 = 0.0;    // previous new control
    double dup0 = 0.0, dup1 = 0.0;  // previous control change
    const double fracu = 2.0 * frac;  // = 1 - tau for the controls; the states keep half of that as slack
    for (int k = 0; k < N; ++k) {
        double u0, u1;
        if (first) {
            u0 = w.ld(TB + B_U + 0, k);
            u1 = w.ld(TB + B_U + 1, k);
        } else {
            const double e0 = x_0 - w.ld(CB + B_X + 0, k), e1 = x_1 - w.ld(CB + B_X + 1, k);
            const double e2 = x_2 - w.ld(CB + B_X + 2, k), e3 = x_3 - w.ld(CB + B_X + 3, k);
            const double c0 = w.ld(CB + B_U + 0, k), c1 = w.ld(CB + B_U + 1, k);
            double s0 = alpha * w.ld(S_KF + 0, k) + w.ld(S_KX + 0, k) * e0 + w.ld(S_KX + 1, k) * e1 +
                        w.ld(S_KX + 2, k) * e2 + w.ld(S_KX + 3, k) * e3;
            double s1 = alpha * w.ld(S_KF + 1, k) + w.ld(S_KX + 4, k) * e0 + w.ld(S_KX + 5, k) * e1 +
                        w.ld(S_KX + 6, k) * e2 + w.ld(S_KX + 7, k) * e3;
            if (k >= 1) {
                const double kp00 = w.ld(S_KP + 0, k), kp01 = w.ld(S_KP + 1, k), kp11 = w.ld(S_KP + 2, k);
                s0 += kp00 * dup0 + kp01 * dup1;
                s1 += kp01 * dup0 + kp11 * dup1;
            }
            // control bounds: clamp each component to the fraction-to-the-boundary box instead of shortening
            // the whole step (saturated accelerations would otherwise jam every iteration)
            u0 = fmin2(fmax2(c0 + s0, ulo_r(0) + fracu * (c0 - ulo_r(0))), uhi_r(0) - fracu * (uhi_r(0) - c0));
            u1 = fmin2(fmax2(c1 + s1, ulo_r(1) + fracu * (c1 - ulo_r(1))), uhi_r(1) - fracu * (uhi_r(1) - c1));
            dup0 = u0 - c0;
            dup1 = u1 - c1;
            if (STORE) w.st(TB + B_U + 0, k, u0);
            if (STORE) w.st(TB + B_U + 1, k, u1);
        }
        if (STORE) w.st(TB + B_X + 0, k, x_0);
        if (STORE) w.st(TB + B_X + 1, k, x_1);
        if (STORE) w.st(TB + B_X + 2, k, x_2);
        if (STORE) w.st(TB + B_X + 3, k, x_3);
        // control costs  (agents/pure_mpc.py:161-165)
        J += 0.01 * sf * wc_ * (u0 * u0 + u1 * u1);
        if (k >= 1) {
            const double d0 = u0 - up0, d1 = u1 - up1;
            J += 0.01 * sf * wd_ * (d0 * d0 + d1 * d1);
        }
        const double slack_u = ((u0 - ulo_r(0)) * (uhi_r(0) - u0)) * ((u1 - ulo_r(1)) * (uhi_r(1) - u1));
        up0 = u0;
        up1 = u1;
        double S, C, sb, cbeta;
        dyn_eval(x_2, u1, S, C, sb, cbeta);
        if (STORE) w.st(TB + B_DYN + 0, k, S);
        if (STORE) w.st(TB + B_DYN + 1, k, C);
        if (STORE) w.st(TB + B_DYN + 2, k, sb);
        if (STORE) w.st(TB + B_DYN + 3, k, cbeta);
        const double n0 = x_0 + dt * (x_3 * C);
        const double n1 = x_1 + dt * (x_3 * S);
        const double n2 = x_2 + dt * (x_3 * kInvWheelbase * sb);
        const double n3 = x_3 + dt * u0;
        if (!first) {
            const double o2 = w.ld(CB + B_X + 2, k + 1), o3 = w.ld(CB + B_X + 3, k + 1);
            if (n2 - xlo_r(0) < frac * (o2 - xlo_r(0)) || xhi_r(0) - n2 < frac * (xhi_r(0) - o2) ||
                n3 - xlo_r(1) < frac * (o3 - xlo_r(1)) || xhi_r(1) - n3 < frac * (xhi_r(1) - o3))
                return false;
        } else {
            if (!(n2 > xlo_r(0)) || !(n2 < xhi_r(0)) || !(n3 > xlo_r(1)) || !(n3 < xhi_r(1))) return false;
        }
        x_0 = n0;
        x_1 = n1;
        x_2 = n2;
        x_3 = n3;
        // one log per TWO stages: 16 slacks, each within 1e-12 .. 1e2, cannot under/overflow a double
        slack_acc *= slack_u * (((x_2 - xlo_r(0)) * (xhi_r(0) - x_2)) * ((x_3 - xlo_r(1)) * (xhi_r(1) - x_3)));
        if ((k & 1) || k + 1 == N) {
            bar -= log(slack_acc);
            slack_acc = 1.0;
        }
        if (k + 1 < N) {
            J += sf * track_cost(w, k + 1, ws_, x_0, x_1, x_2, x_3, (double *)nullptr);
            if (CC) J += sf * (dist_cost(P, w, k + 1, x_0, x_1, (double *)nullptr) + wcoll * x_3 * x_3);
        }
    }
    if (STORE) w.st(TB + B_X + 0, N, x_0);
    if (STORE) w.st(TB + B_X + 1, N, x_1);
    if (STORE) w.st(TB + B_X + 2, N, x_2);
    if (STORE) w.st(TB + B_X + 3, N, x_3);
    Jout = J;
    barout = bar;
    return true;
}

// Linearised forward sweep of the Newton step along the current trajectory: step-length limits a_pr (primal,
// state bounds only - controls are clamped in the rollout) and a_du (dual) by the fraction-to-the-boundary
// rule.  The bounded components of the step (du, d theta, d v) are parked in the adjoint slots S_Y, which are
// free until the next iteration, for the dual update.
template <class WS>
MPC_HD void linear_sweep(const SolveParams &P, WS &w, int cb, double mu, double tau, double &a_pr, double &a_du) {
    const int N = WS::kN > 0 ? WS::kN : P.N;  // compile-time horizon turns LDS offsets into immediates
    const double dt = P.dt;
    const int CB = cb * BUF_SLOTS;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, dp0 = 0, dp1 = 0;
    // step limits without divisions: primal ratio rp = max |d|/s ; dual ratio (-dz)/z kept as a fraction
    double rp = 0.0, rdn = 0.0, rdd = 1.0;
    for (int k = 0; k < N; ++k) {
        double du0 = w.ld(S_KF + 0, k) + w.ld(S_KX + 0, k) * d0 + w.ld(S_KX + 1, k) * d1 + w.ld(S_KX + 2, k) * d2 +
                     w.ld(S_KX + 3, k) * d3;
        double du1 = w.ld(S_KF + 1, k) + w.ld(S_KX + 4, k) * d0 + w.ld(S_KX + 5, k) * d1 + w.ld(S_KX + 6, k) * d2 +
                     w.ld(S_KX + 7, k) * d3;
        if (k >= 1) {
            const double kp00 = w.ld(S_KP + 0, k), kp01 = w.ld(S_KP + 1, k), kp11 = w.ld(S_KP + 2, k);
            du0 += kp00 * dp0 + kp01 * dp1;
            du1 += kp01 * dp0 + kp11 * dp1;
        }
        const double v = w.ld(CB + B_X + 3, k);
        const double S = w.ld(CB + B_DYN + 0, k), C = w.ld(CB + B_DYN + 1, k);
        const double sb = w.ld(CB + B_DYN + 2, k), cbeta = w.ld(CB + B_DYN + 3, k);
        double bp, bpp;
        beta_derivs(sb, cbeta, bp, bpp);
        const double a02 = -dt * v * S, a03 = dt * C, a12 = dt * v * C, a13 = dt * S, a23 = dt * sb * kInvWheelbase;
        const double b01 = -dt * v * S * bp, b11 = dt * v * C * bp, b21 = dt * v * kInvWheelbase * cbeta * bp;
        const double n0 = d0 + a02 * d2 + a03 * d3 + b01 * du1;
        const double n1 = d1 + a12 * d2 + a13 * d3 + b11 * du1;
        const double n2 = d2 + a23 * d3 + b21 * du1;
        const double n3 = d3 + dt * du0;
        d0 = n0; d1 = n1; d2 = n2; d3 = n3;
        dp0 = du0; dp1 = du1;
        w.st(S_Y + 0, k, du0);
        w.st(S_Y + 1, k, du1);
        w.st(S_Y + 2, k + 1, d2);
        w.st(S_Y + 3, k + 1, d3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // i = 0,1: controls of node k;  i = 2,3: theta, v of node k+1
            const bool isu = i < 2;
            const int j = isu ? i : i - 2;
            const double lo = isu ? ulo_r(j) : xlo_r(j), hi = isu ? uhi_r(j) : xhi_r(j);
            const int kk = isu ? k : k + 1;
            const int sv = isu ? (CB + B_U + j) : (CB + B_X + 2 + j);
            const int szl = isu ? (S_ZUL + j) : (S_ZXL + j), szu = isu ? (S_ZUU + j) : (S_ZXU + j);
            const double val = w.ld(sv, kk), d = (i == 0) ? du0 : (i == 1) ? du1 : (i == 2) ? d2 : d3;
            const double rsl = frcp(val - lo), rsu = frcp(hi - val);
            const double zl = w.ld(szl, kk), zu = w.ld(szu, kk);
            const double dzl = (mu - zl * d) * rsl - zl, dzu = (mu + zu * d) * rsu - zu;
            if (!isu) rp = fmax2(rp, fmax2(-d * rsl, d * rsu));
            if (-dzl * rdd > rdn * zl) { rdn = -dzl; rdd = zl; }
            if (-dzu * rdd > rdn * zu) { rdn = -dzu; rdd = zu; }
        }
    }
    a_pr = (rp > tau) ? tau / rp : 1.0;
    a_du = (rdn > tau * rdd) ? tau * rdd / rdn : 1.0;
}

// Dual step  z += a*dz  (dz from the parked Newton step): multipliers that shrink share the
// fraction-to-the-boundary length a_du, growing ones take the full step; clamped like IPOPT (kappa_Sigma 1e10)
// around mu / (new slack) with the slacks of buffer `nb`.
template <class WS>
MPC_HD void dual_update(const SolveParams &P, WS &w, int cb, int nb, double mu, double a_du) {
    const int N = WS::kN > 0 ? WS::kN : P.N;  // compile-time horizon turns LDS offsets into immediates
    const int CB = cb * BUF_SLOTS, NB = nb * BUF_SLOTS;
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool isu = i < 2;
            const int j = isu ? i : i - 2;
            const double lo = isu ? ulo_r(j) : xlo_r(j), hi = isu ? uhi_r(j) : xhi_r(j);
            const int kk = isu ? k : k + 1;
            const int sv = isu ? (B_U + j) : (B_X + 2 + j);
            const int szl = isu ? (S_ZUL + j) : (S_ZXL + j), szu = isu ? (S_ZUU + j) : (S_ZXU + j);
            const double val = w.ld(CB + sv, kk), d = w.ld(S_Y + i, kk);
            const double zl = w.ld(szl, kk), zu = w.ld(szu, kk);
            const double dzl = (mu - zl * d) * frcp(val - lo) - zl, dzu = (mu + zu * d) * frcp(hi - val) - zu;
            const double vn = w.ld(NB + sv, kk);
            const double ml = mu * frcp(vn - lo), mh = mu * frcp(hi - vn);
            w.st(szl, kk, fmax2(fmin2(zl + (dzl > 0.0 ? 1.0 : a_du) * dzl, 1e10 * ml), 1e-10 * ml));
            w.st(szu, kk, fmax2(fmin2(zu + (dzu > 0.0 ? 1.0 : a_du) * dzu, 1e10 * mh), 1e-10 * mh));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// solve one instance.  On return the solution sits in trajectory buffer `cur_out`.
//   ws_/wc_/wd_ : weight_speed (100 if is_collide), weight_control, weight_input_diff
//   wcoll       : 3000 * weight_collision when the collision-cost term is on and is_collide, else 0
// CC selects the variant with the distance/collision terms of agents/archive/pure_mpc.py:189-206.
// ---------------------------------------------------------------------------------------------------
template <bool CC, class WS>
MPC_HD void solve_instance(const SolveParams &P, WS &w, const double *x0, double ws_, double wc_, double wd_,
                           double wcoll, int &status_out, int &iters_out, int &cur_out, double &kkt_out) {
    const int N = WS::kN > 0 ? WS::kN : P.N;  // compile-time horizon turns LDS offsets into immediates
    const double dt = P.dt;
    int cur = 0;
    status_out = 1;
    iters_out = 0;
    cur_out = 0;
    kkt_out = INFINITY;

    // ---- cold start of the reference (agents/pure_mpc.py:240-246: controls 0) rolled out through the
    //      dynamics; a standing vehicle gets a_0 > 0 so that v_1.. are strictly inside v >= 0
    for (int k = 0; k < N; ++k) {
        w.st(S_BUF0 + B_U + 0, k, 0.0);
        w.st(S_BUF0 + B_U + 1, k, 0.0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            w.st(S_ZUL + i, k, 1.0);
            w.st(S_ZUU + i, k, 1.0);
            w.st(S_ZXL + i, k + 1, 1.0);
            w.st(S_ZXU + i, k + 1, 1.0);
        }
    }
    if (x0[3] < 0.01) w.st(S_BUF0 + B_U + 0, 0, (0.01 - x0[3]) / dt);
    double sf = 1.0, Jcur = 0.0, barcur = 0.0;
    if (!rollout<CC, true>(P, w, 0, 0, true, 0.0, 0.0, 1.0, ws_, wc_, wd_, wcoll, x0, Jcur, barcur)) {
        status_out = 3;
        return;
    }
    // ---- objective scaling like IPOPT's gradient-based scaling: sf = 100 / clamp(|grad f|_inf, 100, 1e4)
    {
        double gmax = 0.0;
        for (int k = 1; k < N; ++k) {
            double lx[4];
            if (CC) cache_dist_derivs(P, w, 0, k, 1.0);
            cost_grad<CC>(w, 0, k, 1.0, ws_, wcoll, lx);
            gmax = fmax2(gmax, fmax2(fmax2(fabs(lx[0]), fabs(lx[1])), fmax2(fabs(lx[2]), fabs(lx[3]))));
        }
        gmax = fmax2(gmax, 0.02 * (wc_ + wd_) * fabs(w.ld(S_BUF0 + B_U + 0, 0)));
        sf = 100.0 / fmin2(fmax2(100.0, gmax), 1e4);
        Jcur *= sf;
    }
    const double rd_full = 0.02 * sf * wd_, rc = 0.02 * sf * wc_, qtt = 10.0 * sf;
    const double q33 = sf * (20.0 * ws_ + (CC ? 2.0 * wcoll : 0.0));
    double mu = P.mu_init;
    const double mu_min = P.tol / 10.0;
    int iter = 0, nfail = 0;

    for (iter = 0; iter <= P.max_iter; ++iter) {
        const int CB = cur * BUF_SLOTS;
        // =========================== adjoint sweep: dual residual, complementarity =====================
        double err_d = 0.0, sum_lam = 0.0, sum_z = 0.0, cmax = 0.0, cmin = INFINITY;
        {
            double y0 = 0.0, y1 = 0.0;  // y_{k+1}
            double y2 = -w.ld(S_ZXL + 0, N) + w.ld(S_ZXU + 0, N);
            double y3 = -w.ld(S_ZXL + 1, N) + w.ld(S_ZXU + 1, N);
            double un0 = 0.0, un1 = 0.0;  // u_{k+1}
            for (int k = N - 1; k >= 0; --k) {
                const double u0 = w.ld(CB + B_U + 0, k), u1 = w.ld(CB + B_U + 1, k);
                double um0 = 0.0, um1 = 0.0;
                if (k >= 1) {
                    um0 = w.ld(CB + B_U + 0, k - 1);
                    um1 = w.ld(CB + B_U + 1, k - 1);
                }
                const double rdk = (k >= 1) ? rd_full : 0.0;
                const double v = w.ld(CB + B_X + 3, k);
                const double S = w.ld(CB + B_DYN + 0, k), C = w.ld(CB + B_DYN + 1, k);
                const double sb = w.ld(CB + B_DYN + 2, k), cbeta = w.ld(CB + B_DYN + 3, k);
                double bp, bpp;
                beta_derivs(sb, cbeta, bp, bpp);
                const double a02 = -dt * v * S, a03 = dt * C, a12 = dt * v * C, a13 = dt * S,
                             a23 = dt * sb * kInvWheelbase;
                const double b01 = -dt * v * S * bp, b11 = dt * v * C * bp, b21 = dt * v * kInvWheelbase * cbeta * bp;
                // keep y_{k+1} for the curvature terms of the factorisation sweep
                w.st(S_Y + 0, k + 1, y0);
                w.st(S_Y + 1, k + 1, y1);
                w.st(S_Y + 2, k + 1, y2);
                w.st(S_Y + 3, k + 1, y3);
                const double zul0 = w.ld(S_ZUL + 0, k), zul1 = w.ld(S_ZUL + 1, k);
                const double zuu0 = w.ld(S_ZUU + 0, k), zuu1 = w.ld(S_ZUU + 1, k);
                double r0 = rc * u0 + rdk * (u0 - um0) - zul0 + zuu0;
                double r1 = rc * u1 + rdk * (u1 - um1) - zul1 + zuu1;
                if (k + 1 < N) {
                    r0 -= rd_full * (un0 - u0);
                    r1 -= rd_full * (un1 - u1);
                }
                r0 += dt * y3;
                r1 += b01 * y0 + b11 * y1 + b21 * y2;
                err_d = fmax2(err_d, fmax2(fabs(r0), fabs(r1)));
                sum_z += zul0 + zul1 + zuu0 + zuu1;
                sum_lam += fabs(y0) + fabs(y1) + fabs(y2) + fabs(y3);
                // complementarity products of u_k and (theta, v)_{k+1}
                {
                    const double c0 = (u0 - ulo_r(0)) * zul0, c1 = (uhi_r(0) - u0) * zuu0;
                    const double c2 = (u1 - ulo_r(1)) * zul1, c3 = (uhi_r(1) - u1) * zuu1;
                    cmax = fmax2(cmax, fmax2(fmax2(c0, c1), fmax2(c2, c3)));
                    cmin = fmin2(cmin, fmin2(fmin2(c0, c1), fmin2(c2, c3)));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const double xi = w.ld(CB + B_X + 2 + i, k + 1);
                    const double zl = w.ld(S_ZXL + i, k + 1), zu = w.ld(S_ZXU + i, k + 1);
                    const double c0 = (xi - xlo_r(i)) * zl, c1 = (xhi_r(i) - xi) * zu;
                    cmax = fmax2(cmax, fmax2(c0, c1));
                    cmin = fmin2(cmin, fmin2(c0, c1));
                    sum_z += zl + zu;
                }
                if (k >= 1) {
                    double lx[4];
                    if (CC) cache_dist_derivs(P, w, CB, k, sf);
                    cost_grad<CC>(w, CB, k, sf, ws_, wcoll, lx);
                    const double t0 = lx[0] + y0;
                    const double t1 = lx[1] + y1;
                    const double t2 = lx[2] - w.ld(S_ZXL + 0, k) + w.ld(S_ZXU + 0, k) + a02 * y0 + a12 * y1 + y2;
                    const double t3 =
                        lx[3] - w.ld(S_ZXL + 1, k) + w.ld(S_ZXU + 1, k) + a03 * y0 + a13 * y1 + a23 * y2 + y3;
                    y0 = t0;
                    y1 = t1;
                    y2 = t2;
                    y3 = t3;
                }
                un0 = u0;
                un1 = u1;
            }
        }
        const double s_d = fmax2(100.0, (sum_lam + sum_z) / (10.0 * N)) / 100.0;
        const double s_c = fmax2(100.0, sum_z / (6.0 * N)) / 100.0;
        // monotone barrier update (IPOPT: kappa_eps 10, kappa_mu 0.2, theta_mu 1.5)
        for (;;) {
            const double ec = fmax2(cmax - mu, mu - cmin);
            const double E_mu = fmax2(err_d / s_d, ec / s_c);
            if (E_mu <= 10.0 * mu && mu > mu_min) {
                mu = fmax2(mu_min, fmin2(0.2 * mu, mu * sqrt(mu)));
                continue;
            }
            break;
        }
        const double E0 = fmax2(err_d / s_d, cmax / s_c);
        kkt_out = E0;
        if (E0 <= P.tol) {
            status_out = 0;
            break;
        }
        if (iter == P.max_iter) break;

        // =========================== Riccati / DDP factorisation sweep ================================
        // exact Lagrangian Hessian first; if a control block is not positive definite the sweep is redone with
        // the convex Gauss-Newton model (at most one retry in practice).
        double dV1 = 0.0, delta_w = 0.0;
        bool ok = false, gn = false;
        for (int attempt = 0; attempt < 16 && !ok; ++attempt) {
            ok = true;
            dV1 = 0.0;
            // value function of node k+1: Pxx (sym 4x4), Pxp (4x2), Ppp (sym 2x2), px, pp
            double p00 = delta_w, p01 = 0, p02 = 0, p03 = 0, p11 = delta_w, p12 = 0, p13 = 0, p22, p23 = 0, p33;
            double e00 = 0, e01 = 0, e10 = 0, e11 = 0, e20 = 0, e21 = 0, e30 = 0, e31 = 0;
            double pp00 = 0, pp01 = 0, pp11 = 0, px0 = 0.0, px1 = 0.0, px2, px3, ppv0 = 0, ppv1 = 0;
            {
                double xi = w.ld(CB + B_X + 2, N);
                double rl = frcp(xi - xlo_r(0)), ru = frcp(xhi_r(0) - xi);
                p22 = w.ld(S_ZXL + 0, N) * rl + w.ld(S_ZXU + 0, N) * ru + delta_w;
                px2 = mu * (ru - rl);
                xi = w.ld(CB + B_X + 3, N);
                rl = frcp(xi - xlo_r(1));
                ru = frcp(xhi_r(1) - xi);
                p33 = w.ld(S_ZXL + 1, N) * rl + w.ld(S_ZXU + 1, N) * ru + delta_w;
                px3 = mu * (ru - rl);
            }
            for (int k = N - 1; k >= 0; --k) {
                const double rdk = (k >= 1) ? rd_full : 0.0;
                const double v = w.ld(CB + B_X + 3, k);
                const double S = w.ld(CB + B_DYN + 0, k), C = w.ld(CB + B_DYN + 1, k);
                const double sb = w.ld(CB + B_DYN + 2, k), cbeta = w.ld(CB + B_DYN + 3, k);
                double bp, bpp;
                beta_derivs(sb, cbeta, bp, bpp);
                const double a02 = -dt * v * S, a03 = dt * C, a12 = dt * v * C, a13 = dt * S,
                             a23 = dt * sb * kInvWheelbase;
                const double b01 = -dt * v * S * bp, b11 = dt * v * C * bp, b21 = dt * v * kInvWheelbase * cbeta * bp;
                const double u0 = w.ld(CB + B_U + 0, k), u1 = w.ld(CB + B_U + 1, k);
                // ---- stage Hessian / gradient (cost + barrier + constraint curvature)
                double l00 = 0, l01 = 0, l11 = 0, l22 = 0, l23 = 0, l33 = 0, lxu21 = 0, lxu31 = 0;
                double lx0 = 0, lx1 = 0, lx2 = 0, lx3 = 0;
                double wdd = 0.0;
                if (!gn) {
                    const double yy0 = w.ld(S_Y + 0, k + 1), yy1 = w.ld(S_Y + 1, k + 1), yy2 = w.ld(S_Y + 2, k + 1);
                    const double g = -(yy0 * C + yy1 * S), h = -(yy0 * S - yy1 * C);
                    const double wtt = dt * v * g, wtv = dt * h;
                    const double wtd = dt * v * g * bp;
                    const double wvd = dt * h * bp + dt * yy2 * cbeta * bp * kInvWheelbase;
                    wdd = dt * v * (g * bp * bp + h * bpp) + dt * yy2 * v * kInvWheelbase * (-sb * bp * bp + cbeta * bpp);
                    if (k >= 1) {
                        l22 = wtt;
                        l23 = wtv;
                        lxu21 = wtd;
                        lxu31 = wvd;
                    }
                }
                if (k >= 1) {
                    double lx[4];
                    cost_grad<CC>(w, CB, k, sf, ws_, wcoll, lx);
                    double t00, t01, t11;
                    track_hess(w, k, t00, t01, t11);
                    l00 = sf * t00 + delta_w;
                    l01 = sf * t01;
                    l11 = sf * t11 + delta_w;
                    if (CC) {
                        const int QS = gn ? S_QG : S_Q;
                        l00 += w.ld(QS + 0, k);
                        l01 += w.ld(QS + 1, k);
                        l11 += w.ld(QS + 2, k);
                    }
                    lx0 = lx[0];
                    lx1 = lx[1];
                    const double xi = w.ld(CB + B_X + 2, k);
                    double rl = frcp(xi - xlo_r(0)), ru = frcp(xhi_r(0) - xi);
                    l22 += qtt + w.ld(S_ZXL + 0, k) * rl + w.ld(S_ZXU + 0, k) * ru + delta_w;
                    lx2 = lx[2] + mu * (ru - rl);
                    rl = frcp(v - xlo_r(1));
                    ru = frcp(xhi_r(1) - v);
                    l33 = q33 + w.ld(S_ZXL + 1, k) * rl + w.ld(S_ZXU + 1, k) * ru + delta_w;
                    lx3 = lx[3] + mu * (ru - rl);
                }
                double luu00, luu11, lu0, lu1, lp0 = 0.0, lp1 = 0.0;
                {
                    double um0 = 0.0, um1 = 0.0;
                    if (k >= 1) {
                        um0 = w.ld(CB + B_U + 0, k - 1);
                        um1 = w.ld(CB + B_U + 1, k - 1);
                    }
                    double rl = frcp(u0 - ulo_r(0)), ru = frcp(uhi_r(0) - u0);
                    luu00 = rc + rdk + w.ld(S_ZUL + 0, k) * rl + w.ld(S_ZUU + 0, k) * ru + delta_w;
                    lu0 = rc * u0 + rdk * (u0 - um0) + mu * (ru - rl);
                    rl = frcp(u1 - ulo_r(1));
                    ru = frcp(uhi_r(1) - u1);
                    luu11 = rc + rdk + w.ld(S_ZUL + 1, k) * rl + w.ld(S_ZUU + 1, k) * ru + delta_w + wdd;
                    lu1 = rc * u1 + rdk * (u1 - um1) + mu * (ru - rl);
                    lp0 = -rdk * (u0 - um0);
                    lp1 = -rdk * (u1 - um1);
                }
                // ---- M = Pxx A (4x4; columns 0,1 are those of Pxx), G = Pxx B + Pxp (4x2)
                const double m02 = p00 * a02 + p01 * a12 + p02, m03 = p00 * a03 + p01 * a13 + p02 * a23 + p03;
                const double m12 = p01 * a02 + p11 * a12 + p12, m13 = p01 * a03 + p11 * a13 + p12 * a23 + p13;
                const double m22 = p02 * a02 + p12 * a12 + p22, m23 = p02 * a03 + p12 * a13 + p22 * a23 + p23;
                const double m32 = p03 * a02 + p13 * a12 + p23, m33 = p03 * a03 + p13 * a13 + p23 * a23 + p33;
                const double g00 = dt * p03 + e00, g01 = p00 * b01 + p01 * b11 + p02 * b21 + e01;
                const double g10 = dt * p13 + e10, g11 = p01 * b01 + p11 * b11 + p12 * b21 + e11;
                const double g20 = dt * p23 + e20, g21 = p02 * b01 + p12 * b11 + p22 * b21 + e21;
                const double g30 = dt * p33 + e30, g31 = p03 * b01 + p13 * b11 + p23 * b21 + e31;
                // ---- Hxx = Lxx + A' M (symmetric)
                const double h00 = l00 + p00, h01 = l01 + p01, h11 = l11 + p11;
                const double h02 = m02, h03 = m03, h12 = m12, h13 = m13;
                const double h22 = l22 + a02 * m02 + a12 * m12 + m22;
                const double h23 = l23 + a02 * m03 + a12 * m13 + m23;
                const double h33 = l33 + a03 * m03 + a13 * m13 + a23 * m23 + m33;
                (void)m32;
                // ---- Hxu = Lxu + A' G
                const double hxu00 = g00, hxu01 = g01, hxu10 = g10, hxu11 = g11;
                const double hxu20 = a02 * g00 + a12 * g10 + g20, hxu21 = lxu21 + a02 * g01 + a12 * g11 + g21;
                const double hxu30 = a03 * g00 + a13 * g10 + a23 * g20 + g30;
                const double hxu31 = lxu31 + a03 * g01 + a13 * g11 + a23 * g21 + g31;
                // ---- hx = lx + A' px
                const double hx0 = lx0 + px0, hx1 = lx1 + px1, hx2 = lx2 + a02 * px0 + a12 * px1 + px2;
                const double hx3 = lx3 + a03 * px0 + a13 * px1 + a23 * px2 + px3;
                // ---- Huu = Luu + Ppp + B' G + Pxp' B
                const double huu00 = luu00 + pp00 + dt * g30 + dt * e30;
                const double huu01a = pp01 + dt * g31 + (e00 * b01 + e10 * b11 + e20 * b21);
                const double huu10a = pp01 + (b01 * g00 + b11 * g10 + b21 * g20) + dt * e31;
                const double huu11 =
                    luu11 + pp11 + (b01 * g01 + b11 * g11 + b21 * g21) + (e01 * b01 + e11 * b11 + e21 * b21);
                const double hu0 = lu0 + ppv0 + dt * px3;
                const double hu1 = lu1 + ppv1 + b01 * px0 + b11 * px1 + b21 * px2;
                const double ha = huu00, hb = 0.5 * (huu01a + huu10a), hc = huu11;
                const double det = ha * hc - hb * hb;
                if (!(ha > 0.0) || !(hc > 0.0) || !(det > 1e-12 * ha * hc)) {
                    ok = false;
                    break;
                }
                const double idet = frcp(det);
                const double i00 = hc * idet, i01 = -hb * idet, i11 = ha * idet;
                // gains
                const double kx00 = -(i00 * hxu00 + i01 * hxu01), kx01 = -(i00 * hxu10 + i01 * hxu11);
                const double kx02 = -(i00 * hxu20 + i01 * hxu21), kx03 = -(i00 * hxu30 + i01 * hxu31);
                const double kx10 = -(i01 * hxu00 + i11 * hxu01), kx11 = -(i01 * hxu10 + i11 * hxu11);
                const double kx12 = -(i01 * hxu20 + i11 * hxu21), kx13 = -(i01 * hxu30 + i11 * hxu31);
                const double kp00 = rdk * i00, kp01 = rdk * i01, kp11 = rdk * i11;
                const double kf0 = -(i00 * hu0 + i01 * hu1), kf1 = -(i01 * hu0 + i11 * hu1);
                w.st(S_KX + 0, k, kx00); w.st(S_KX + 1, k, kx01); w.st(S_KX + 2, k, kx02); w.st(S_KX + 3, k, kx03);
                w.st(S_KX + 4, k, kx10); w.st(S_KX + 5, k, kx11); w.st(S_KX + 6, k, kx12); w.st(S_KX + 7, k, kx13);
                w.st(S_KP + 0, k, kp00); w.st(S_KP + 1, k, kp01); w.st(S_KP + 2, k, kp11);
                w.st(S_KF + 0, k, kf0); w.st(S_KF + 1, k, kf1);
                dV1 += 0.5 * (kf0 * hu0 + kf1 * hu1);
                // value function of node k:  Pxx = sym(Hxx + Hxu Kx), Pxp = Hxu Kp, Ppp = rd I - rd Kp
                const double n00 = h00 + hxu00 * kx00 + hxu01 * kx10;
                const double n01 = 0.5 * ((h01 + hxu00 * kx01 + hxu01 * kx11) + (h01 + hxu10 * kx00 + hxu11 * kx10));
                const double n02 = 0.5 * ((h02 + hxu00 * kx02 + hxu01 * kx12) + (h02 + hxu20 * kx00 + hxu21 * kx10));
                const double n03 = 0.5 * ((h03 + hxu00 * kx03 + hxu01 * kx13) + (h03 + hxu30 * kx00 + hxu31 * kx10));
                const double n11 = h11 + hxu10 * kx01 + hxu11 * kx11;
                const double n12 = 0.5 * ((h12 + hxu10 * kx02 + hxu11 * kx12) + (h12 + hxu20 * kx01 + hxu21 * kx11));
                const double n13 = 0.5 * ((h13 + hxu10 * kx03 + hxu11 * kx13) + (h13 + hxu30 * kx01 + hxu31 * kx11));
                const double n22 = h22 + hxu20 * kx02 + hxu21 * kx12;
                const double n23 = 0.5 * ((h23 + hxu20 * kx03 + hxu21 * kx13) + (h23 + hxu30 * kx02 + hxu31 * kx12));
                const double n33 = h33 + hxu30 * kx03 + hxu31 * kx13;
                e00 = hxu00 * kp00 + hxu01 * kp01; e01 = hxu00 * kp01 + hxu01 * kp11;
                e10 = hxu10 * kp00 + hxu11 * kp01; e11 = hxu10 * kp01 + hxu11 * kp11;
                e20 = hxu20 * kp00 + hxu21 * kp01; e21 = hxu20 * kp01 + hxu21 * kp11;
                e30 = hxu30 * kp00 + hxu31 * kp01; e31 = hxu30 * kp01 + hxu31 * kp11;
                px0 = hx0 + hxu00 * kf0 + hxu01 * kf1;
                px1 = hx1 + hxu10 * kf0 + hxu11 * kf1;
                px2 = hx2 + hxu20 * kf0 + hxu21 * kf1;
                px3 = hx3 + hxu30 * kf0 + hxu31 * kf1;
                p00 = n00; p01 = n01; p02 = n02; p03 = n03; p11 = n11; p12 = n12; p13 = n13; p22 = n22; p23 = n23; p33 = n33;
                pp00 = rdk - rdk * kp00; pp01 = -rdk * kp01; pp11 = rdk - rdk * kp11;
                ppv0 = lp0 - rdk * kf0;
                ppv1 = lp1 - rdk * kf1;
            }
            if (!ok) {
                if (!gn) {
                    gn = true;  // convex Gauss-Newton model for this iteration
                } else {
                    delta_w = (delta_w == 0.0) ? 1e-8 : 100.0 * delta_w;  // numerically singular even so
                }
                if (delta_w > 1e40) break;
            }
        }
        if (!ok) {
            status_out = 2;
            break;
        }

        // =========================== linear forward sweep: Newton step, step-length limits ==============
        const double tau = fmax2(0.99, 1.0 - mu);
        double a_pr = 1.0, a_du = 1.0;
        linear_sweep(P, w, cur, mu, tau, a_pr, a_du);

        // =========================== nonlinear rollout + Armijo on the barrier objective ================
        const double phi0 = Jcur + mu * barcur;
        const int tb = cur ^ 1;
        // Trial k uses alpha = a_pr * 4^-k, k = 0..5, and the first k that passes Armijo is accepted.  Trial 0 is
        // run (and stored) by all lanes; when it fails, the R = WS::kReplicas replica lanes of this instance evaluate
        // R further trials at once without storing, and the winner is re-run with stores - the same result as the
        // sequential search, in at most 3-4 rollouts instead of 6 for the instances that backtrack deep.
        double alpha = a_pr, Jn = 0.0, barn = 0.0;
        bool accepted = false;
        const double frac = 0.5 * (1.0 - tau);
        auto armijo = [&](double a, double Jt, double bt) {
            return Jt + mu * bt <= phi0 + 1e-4 * a * 2.0 * dV1 + 1e-12 * fabs(phi0);
        };
        if (rollout<CC, true>(P, w, cur, tb, false, alpha, frac, sf, ws_, wc_, wd_, wcoll, x0, Jn, barn) &&
            armijo(alpha, Jn, barn)) {
            accepted = true;
        } else {
            constexpr int R = WS::kReplicas;
            int k = 1;
            while (k < 6 && !accepted) {
                if (R > 1) {
                    const int myk = k + w.replica();
                    double a = a_pr;
                    for (int q = 0; q < myk; ++q) a *= 0.25;
                    double Jt = 0.0, bt = 0.0;
                    const bool pass = myk < 6 &&
                                      rollout<CC, false>(P, w, cur, tb, false, a, frac, sf, ws_, wc_, wd_, wcoll, x0, Jt, bt) &&
                                      armijo(a, Jt, bt);
                    const int r = w.first_passing(pass);  // smallest replica index of this instance that passed
                    if (r >= 0) {
                        alpha = a_pr;
                        for (int q = 0; q < k + r; ++q) alpha *= 0.25;
                        rollout<CC, true>(P, w, cur, tb, false, alpha, frac, sf, ws_, wc_, wd_, wcoll, x0, Jn, barn);
                        accepted = true;
                    }
                    k += R;
                } else {
                    alpha *= 0.25;
                    if (rollout<CC, true>(P, w, cur, tb, false, alpha, frac, sf, ws_, wc_, wd_, wcoll, x0, Jn, barn) &&
                        armijo(alpha, Jn, barn))
                        accepted = true;
                    ++k;
                }
            }
        }
        // =========================== dual step (own fraction-to-the-boundary length), accept ============
        dual_update(P, w, cur, accepted ? tb : cur, mu, a_du);
        if (accepted) {
            cur = tb;
            Jcur = Jn;
            barcur = barn;
            nfail = 0;
        } else if (++nfail >= 3) {
            // three consecutive iterations without an acceptable step: the primal point cannot move any more
            // (a kink of the collision cost at d = 1, or numerical stationarity) - stop instead of burning the budget
            status_out = 4;
            ++iter;
            break;
        }
    }
    iters_out = iter;
    cur_out = cur;
}

}  // namespace mpc
