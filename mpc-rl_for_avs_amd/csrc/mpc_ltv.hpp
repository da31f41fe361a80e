// mpc_ltv.hpp - the iterative-linear MPC of the reference (agents/pure_mpc_linear.py:153-271) on one wave64.
//
// One call of IterativeLinearMPC_Agent._solve: forward-simulate the stored control profile (predict_motion, :84-110),
// linearise the bicycle model about it stage by stage (linear_model_matrix, :62-82: steer_ref = 0 and NO affine term),
// and solve the convex QP of _linear_mpc_control (:205-257) that the reference hands to cvxpy -> ECOS:
//   min  sum_t [0.01 a_t^2 + 0.01 d_t^2] + sum_{t<T-1} [0.01 (a_{t+1}-a_t)^2 + (d_{t+1}-d_t)^2]
//        + sum_{t<T} [20 (v_t - vref_t)^2 + 0.5 (yaw_t - href_t)^2] + T [(x_T-xr)^2 + (y_T-yr)^2 + 0.5 (yaw_T-hr)^2]
//   s.t. x_{t+1} = A_t x_t + B_t u_t,  -5 <= a <= 2,  |d| <= 30 deg,  |d_{t+1} - d_t| <= 30 deg/s dt,  0 <= v_t <= 40/3.6.
// The QP is strictly convex in the controls, so its minimiser is unique; what matters for parity is the QP, not the
// path ECOS takes to it.
//
// Method: the states are functions of the controls (single shooting, exact because the model is linear), every
// inequality gets a slack, and the QP is solved by an infeasible-start primal-dual interior-point method with
// Mehrotra's predictor-corrector and one common step length.  Its Newton systems have the optimal-control structure
// and are solved by a Riccati sweep over the state augmented with the previous control (the rate limit and the
// input-difference cost couple u_t with u_{t-1}) in the 4x4 block form of mpc_wave.hpp's sweep4 (round 6: fourteen
// v_mfma_f64_4x4x4f64 per stage, the rank-two Woodbury terms of the two stiff rows riding in the same products,
// factor_and_solve below; rounds 1 - 5: an 8x8 block form with nine products and eight block moves); the corrector
// reuses the factorisation through a scalar gradient-only sweep.
// Everything separable runs stage-parallel, lane k = stage k: lane k owns the eight inequalities of stage k
// (a >= -5, a <= 2, d >= -30, d <= 30, rate-, rate+, v_{k+1} >= 0, v_{k+1} <= vmax) with their slacks and multipliers
// in registers.
//
// Internal state order is (x, y, yaw, v) - the order of mpc_wave.hpp, so that F = [A B; 0 I] has the same sparsity -
// the interface order (x, y, v, yaw) of the reference is restored by the caller.
#pragma once

#include "mpc_wave.hpp"

namespace mpc {
namespace ltv {

using wave::kLanes;
using wave::PerLane;

// reference constants, agents/pure_mpc_linear.py:27-37
constexpr double kR0 = 0.01, kR1 = 0.01;      // R
constexpr double kRd0 = 0.01, kRd1 = 1.0;     // Rd
constexpr double kQv = 20.0, kQyaw = 0.5;     // Q_v_yaw
constexpr double kQfXY = 1.0, kQfYaw = 0.5;   // Qf (x, y, v = 0, yaw), scaled by the horizon (:134)
constexpr double kMaxSteer = 30.0 * MPC_PI / 180.0;
constexpr double kMaxDSteer = 30.0 * MPC_PI / 180.0;
constexpr double kMaxAccel = 2.0, kMaxDecel = -5.0;
constexpr double kMaxSpeed = 40.0 / 3.6;

// interior-point constants (the CPU oracle uses the same)
constexpr double kSInitMin = 0.3, kZInit = 100.0;
constexpr double kTolP = 1e-9, kTolMu = 1e-10;   // |c - s|_inf, s.z / m
constexpr double kTolDRel = 1e-9;               // |grad L|_inf relative to max(1e3, its value at the start)

enum : int { ST_CONVERGED = 0, ST_MAX_ITER = 1, ST_FACTORIZATION = 2, ST_INFEASIBLE = 3 };

// per-stage LDS slots (doubles)
enum : int {
    L_X = 0,     // 4  node k: x, y, yaw, v
    L_U = 4,     // 2  stage k: a, d   (on entry: the nominal profile)
    L_LIN = 6,   // 8  a02 a03 a12 a13 a23 b01 b11 b21 of stage k (a23 = b01 = b11 = 0 for this model)
    L_DV = 14,   // 1  node k: z/s summed over its two speed bounds (enters the sweep as a rank-one term, not through L_H)
    L_QV = 15,   // 1  node k: net multiplier term of its speed bounds in the current right-hand side
    L_ZR = 16,   // 1  stage k: net multiplier of the rate limit
    L_H = 17,    // 6  stage Hessian diagonal WITHOUT the speed-bound and rate-limit barrier terms: (yaw,yaw) (v,v)
                 //    (p0,p0) (p1,p1) (u0,u0) (u1,u1); (p,u) = -(p,p)
    L_G = 23,    // 8  stage gradient: x 4, previous control 2, control 2
    L_KX = 31,   // 8  gains: d u = kf + Kx dx + Kp du_prev
    L_KP = 39,   // 4
    L_KF = 43,   // 2
    L_IH = 45,   // 3  inverse of the control block (for the gradient-only sweep)
    L_DU = 48,   // 2  step of the controls of stage k
    L_DX = 50,   // 4  step of node k
    L_Y = 54,    // 2  node k: adjoint of (yaw, v)
    L_DR = 56,   // 1  stage k: z/s summed over its two rate-limit rows
    L_M = 57,    // 12 kap [g1 g2]' (2 x 6): the stiff rows' steps in constraint space are w / lambda, w = t + M [dx; du_prev]
    L_Z = 69,    // 4  Z = Y kap, gives t = kap Fu' kf0 = -Z' hu for a new right-hand side
    L_T = 73,    // 2  t of the current right-hand side
    // the constants 1 and dt of the stage matrix F = [A B; 0 I], once per stage (round 6, as in mpc_wave.hpp): every operand of a
    // Riccati stage is then a stage-relative word (the constant 0: L_G + 0, the x-row of the stage gradient, always zero)
    L_ONE = 75,  // 1
    L_DTC = 76,  // 1
    L_SLOTS = 77 // (odd: lane k of a stage-parallel phase addresses word k * stride + slot)
};
enum : int { SC_ZERO = 0, SC_ONE = 1, SC_DT = 2, SC_TOLD = 3, SC_X0 = 4, SC_T = 8, SC_MINEQ = 9, SC_SIZE = 10 };   // SC_X0: the ego state x, y, yaw, v; SC_TOLD: the dual tolerance of this solve; SC_T, SC_MINEQ: horizon, number of inequalities

MPC_HD constexpr int lds_doubles(int N) { return L_SLOTS * (N + 1) + SC_SIZE; }

struct LtvParams {
    int N;
    int max_iter;
    int passes;   // linearisation passes per call (the loop at agents/pure_mpc_linear.py:189)
    double dt;
};

// CTX::kRelax (absent: 0) - what the build may keep in registers (mpc_engine.hip: the two builds of mpc_ltv_kernel):
//   bit 3: the primal residuals stay in registers from the residual phase to the predictor step instead of being recomputed
template <class CTX, class = void>
struct relax_bits { static constexpr int value = 0; };
template <class CTX>
struct relax_bits<CTX, decltype((void)CTX::kRelax)> { static constexpr int value = CTX::kRelax; };

template <class CTX>
struct Solver {
    static constexpr bool kKeepResidual = (relax_bits<CTX>::value & 8) != 0;
    const LtvParams &P;
    CTX &c;
    const int N, SCR;
    const double dt;

    // the ego state (x, y, yaw, v) is parked in LDS: it is read at the start of a pass only, and four wave-uniform doubles
    // that the vector unit converted from the float32 observation would otherwise sit in eight vector registers for the
    // whole kernel
    MPC_HD Solver(const LtvParams &P_, CTX &c_, const double *x0_)
        : P(P_), c(c_), N(P_.N), SCR(L_SLOTS * (P_.N + 1)), dt(P_.dt) {
        for (int i = 0; i < 4; ++i) c.st(SCR + SC_X0 + i, x0_[i]);
        c.phase([&](int) {});
    }
    MPC_HD double X0(int i) const { return c.ld(SCR + SC_X0 + i); }
    MPC_HD double S(int k, int slot) const { return c.ld(k * L_SLOTS + slot); }
    MPC_HD void S(int k, int slot, double v) { c.st(k * L_SLOTS + slot, v); }

    PerLane<double> red_a, red_b, red_c, red_d, red_e;
    PerLane<double> s_[8], z_[8], rp_[8], pr_[8];   // slack, multiplier, primal residual, predictor product per inequality
    // matrix-core roles (round 6: the 4x4 form of mpc_wave.hpp's sweep4): lane l = 16 hi + 4 blk + lo holds element [hi][lo] of a
    // 4x4 operand, the same in all four blocks; LDS word of this lane's element of each operand, stage-relative
    PerLane<int> q_a, q_b, q_lxx, q_m, q_lx, q_lu, q_st;

    // reference columns of node k: window row min(target + k, M - 1) (:178-187)
    MPC_HD double xr(int k) const { return c.ref(k, R_X); }
    MPC_HD double yr(int k) const { return c.ref(k, R_Y); }
    MPC_HD double hr(int k) const { return c.ref(k, R_H); }
    MPC_HD double vr(int k) const { return c.refv(k); }

    MPC_HD int f_word(int r, int cl) const {
        return wave::stage_transition_word(r, cl, L_LIN, L_G + 0, L_ONE, L_DTC);
    }
    // (pure functions of the lane id, recomputed at the start of every factorisation from a lane id the compiler cannot
    // see through - as in mpc_wave.hpp - so that their registers are live during the sweep only)
    MPC_HD void set_roles() {
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
            const int zero = L_G + 0;
            q_a.at(lane_) = f_word(hi, lo);                          // A[hi][lo]
            q_b.at(lane_) = f_word(hi, lo < 2 ? 6 + lo : 4);         // B~[hi][lo]: the columns a, delta of F, then zeros
            q_lxx.at(lane_) = (hi == lo && hi >= 2) ? L_H + (hi - 2) : zero;      // Lxx = diag(0, 0, H_yaw, H_v)
            q_m.at(lane_) = (hi == lo && hi < 2) ? L_H + 4 + hi : zero;          // Luu = diag(H_a, H_delta) without the stiff rows
            q_lx.at(lane_) = lo == 0 ? L_G + hi : zero;                          // lx in column 0
            q_lu.at(lane_) = (lo == 0 && hi < 2) ? L_G + 6 + hi : zero;          // lu in column 0
            // what this lane stores at the end of the stage: block 0 the gains Kx[hi][lo] (rows 0, 1), block 1 the x-part of
            // kap [g1 g2]' (held in rows 2, 3)
            int out = -1;
            out = (blk == 0 && hi < 2) ? L_KX + 4 * hi + lo : out;
            out = (blk == 1 && hi >= 2) ? L_M + 6 * (hi - 2) + lo : out;
            q_st.at(lane_) = out;
        });
    }

    // value c_i of inequality i of stage k at the current iterate, and d c_i / d(argument)
    MPC_HD static double csign(int i) { return (i == 0 || i == 2 || i == 5 || i == 6) ? 1.0 : -1.0; }
    MPC_HD double cval(int k, int i) const {
        switch (i) {
            case 0: return S(k, L_U + 0) - kMaxDecel;
            case 1: return kMaxAccel - S(k, L_U + 0);
            case 2: return S(k, L_U + 1) + kMaxSteer;
            case 3: return kMaxSteer - S(k, L_U + 1);
            case 4: return kMaxDSteer * dt - (S(k, L_U + 1) - S(k - 1, L_U + 1));
            case 5: return kMaxDSteer * dt + (S(k, L_U + 1) - S(k - 1, L_U + 1));
            case 6: return S(k + 1, L_X + 3);
            default: return kMaxSpeed - S(k + 1, L_X + 3);
        }
    }
    // The rows the sweep treats by the Woodbury identity get their step in constraint space, f' [dx; dp; du] = w / lambda
    // with w = t + M [dx_k; du_{k-1}]: differencing the controls (d_k - d_{k-1}) or summing accelerations into v would
    // leave an absolute rounding error eps |du|, which the weights z / s ~ 1e16 of active rows turn into O(1) errors of dz.
    // dr: step of d_k - d_{k-1} (rate rows), dv: step of v_{k+1} (speed rows), for the direction parked in L_DU / L_DX
    MPC_HD void stiff_steps(int k, double &dr, double &dv) const {
        double w0 = S(k, L_T + 0), w1 = S(k, L_T + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double dj = S(k, L_DX + j);
            w0 += S(k, L_M + j) * dj;
            w1 += S(k, L_M + 6 + j) * dj;
        }
        if (k >= 1) {
            const double p0 = S(k - 1, L_DU + 0), p1 = S(k - 1, L_DU + 1);
            w0 += S(k, L_M + 4) * p0 + S(k, L_M + 5) * p1;
            w1 += S(k, L_M + 10) * p0 + S(k, L_M + 11) * p1;
        }
        dv = w0 * frcp(S(k + 1, L_DV));
        dr = k >= 1 ? w1 * frcp(S(k, L_DR)) : 0.0;
    }
    // step of the argument of inequality i of stage k
    MPC_HD double darg(int k, int i, double dr, double dv) const {
        if (i < 2) return S(k, L_DU + 0);
        if (i < 4) return S(k, L_DU + 1);
        return i < 6 ? dr : dv;
    }
    MPC_HD static bool valid(int k, int i) { return k >= 1 || (i != 4 && i != 5); }
    // primal residual c_i - s_i of inequality i of stage k; a residual below the rounding resolution of its constraint
    // value is zero (kept, it would be multiplied by z / s).  Recomputed where it is needed - the iterate does not change
    // between the residual phase and the predictor step - instead of being held in registers across the factorisation.
    MPC_HD double presid(int k, int i, double s) const {
        const double rr = cval(k, i) - s;
        return fabs(rr) <= 8.0 * 2.220446049250313e-16 * kMaxSpeed ? 0.0 : rr;
    }

    // stage gradient of stage k for the per-inequality right-hand-side terms q (g = grad f - C' q), own part: the
    // control and previous-control rows and the speed-bound term of node k + 1
    MPC_HD void put_own_gradient(int k, const double *q) {
        const double qa = q[0] - q[1], qd = q[2] - q[3], qr = q[5] - q[4], qv = q[6] - q[7];
        const double a = S(k, L_U + 0), d = S(k, L_U + 1);
        double da = 0.0, dd = 0.0;
        if (k >= 1) {
            da = 2.0 * kRd0 * (a - S(k - 1, L_U + 0));
            dd = 2.0 * kRd1 * (d - S(k - 1, L_U + 1));
        }
        S(k, L_G + 4, -da);
        S(k, L_G + 5, -dd + qr);
        S(k, L_G + 6, 2.0 * kR0 * a + da - qa);
        S(k, L_G + 7, 2.0 * kR1 * d + dd - qd - qr);
        S(k + 1, L_QV, qv);
    }
    // state rows of the stage gradient of stage k (needs L_QV of node k, written by lane k - 1)
    MPC_HD void put_state_gradient(int k) {
        S(k, L_G + 0, 0.0);
        S(k, L_G + 1, 0.0);
        if (k >= 1) {
            S(k, L_G + 2, 2.0 * kQyaw * (S(k, L_X + 2) - hr(k)));
            S(k, L_G + 3, 2.0 * kQv * (S(k, L_X + 3) - vr(k)) - S(k, L_QV));
        } else {
            S(k, L_G + 2, 0.0);
            S(k, L_G + 3, 0.0);
        }
    }
    // gradient of the terminal cost at node N with the speed-bound term of node N
    MPC_HD void terminal_gradient(double *pT) const {
        const double T = (double)N;
        pT[0] = 2.0 * T * kQfXY * (S(N, L_X + 0) - xr(N));
        pT[1] = 2.0 * T * kQfXY * (S(N, L_X + 1) - yr(N));
        pT[2] = 2.0 * T * kQfYaw * (S(N, L_X + 2) - hr(N));
        pT[3] = -S(N, L_QV);
    }

    // ---- Riccati factorisation + first solve on the matrix core ----------------------------------------------------
    // Round 6: the 4x4 form of mpc_wave.hpp (sweep4) instead of the 8x8 block form with block moves of rounds 1 - 5.  With the
    // value function of node k + 1 split as V(x, p) = 1/2 x'Pxx x + x'Pxp p + 1/2 p'Ppp p + px'x + pp'p (x: state 4, p: previous
    // control 2) and F = [A B; 0 I] the BASE block of the stage (everything but the two stiff rows) is
    //     Qxx = Lxx + A'Pxx A      Qux = (Pxx B + Pxp)'A      Quu = Luu + B'(Pxx B + Pxp) + Pxp'B + Ppp      Qpu = -diag(h)
    //     qx = lx + A'px           qu = lu + B'px + pp        G = Quu^-1,  K0x = -G Qux,  K0p = G diag(h),  kb = -G qu
    // and the two stiff rows - speed bounds of node k + 1 (weight l1 on f1 = [e_v; dt e_a]) and the rate limit of the stage (weight
    // l2 on f2 = [-e_p1; e_delta]) - come back through the Woodbury identity exactly as before (never subtracting large numbers):
    //     kap = (diag(1/l) + Fu'G Fu)^-1,  Y = G Fu,  Z = Y kap,  g = Fx + K0'Fu (g1 = e_v + dt K0(0,.)', g2 = -e_p1 + K0(1,.)'),
    //     P = P0 + g kap g',  K = K0 - Z g',  kf = kb - Y (kap Fu'kb),  p = p0(kf) + Fx (kap Fu'kb).
    // Everything 4-dimensional is a matrix-core block, everything 2-dimensional wave-uniform scalar algebra.  The rank-two
    // update rides in the SAME products as the base recursion: with Ya4 = -adj(Quu) Qux in rows 0, 1 AND again in rows 2, 3,
    //     a operand = [Qux / det (rows 0, 1) ; kap [g1 g2]'_x (rows 2, 3)],   b operand = [Ya4 (rows 0, 1) ; [g1 g2]'_x (rows 2, 3)]
    // gives Pxx' = Qxx + Qux'K0x + gx kap gx' in ONE product, likewise Pxp'.  Fourteen products per stage, no block moves, every
    // operand a stage-relative word; the 8x8 form needed nine products, eight block moves and per-lane select chains over the
    // gain rows - 456 instructions per stage against ~200 (the kernel is bound by instruction issue, tools/ubench/issue_probe.hip).
    MPC_HD bool factor_and_solve() {
        set_roles();
        double pT[4];
        terminal_gradient(pT);
        const double T = (double)N;
        PerLane<double> PXX, PXP, PX;
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int hi = lane >> 4, lo = lane & 3;
            PXX.at(lane_) = (hi == lo && hi < 3) ? (hi < 2 ? 2.0 * T * kQfXY : 2.0 * T * kQfYaw) : 0.0;
            PXP.at(lane_) = 0.0;
            const double c0 = lo == 0 ? 1.0 : 0.0;
            PX.at(lane_) = c0 * ((hi == 0 ? 1.0 : 0.0) * pT[0] + (hi == 1 ? 1.0 : 0.0) * pT[1] + (hi == 2 ? 1.0 : 0.0) * pT[2] + (hi == 3 ? 1.0 : 0.0) * pT[3]);
        });
        double pp00 = 0.0, pp01 = 0.0, pp11 = 0.0, ppv0 = 0.0, ppv1 = 0.0;    // Ppp, pp of the node behind the stage
#pragma unroll 1
        for (int k = N - 1; k >= 0; --k) {
            PerLane<double> RA, RB, QXX, QUX, M, QX, QU, T0, T1;
            c.lanes([&](int lane) {
                const int base = k * L_SLOTS;
                RA.at(lane) = c.ld(base + q_a.at(lane));
                RB.at(lane) = c.ld(base + q_b.at(lane));
                QXX.at(lane) = c.ld(base + q_lxx.at(lane));
                M.at(lane) = c.ld(base + q_m.at(lane));
                QX.at(lane) = c.ld(base + q_lx.at(lane));
                QU.at(lane) = c.ld(base + q_lu.at(lane));
                QUX.at(lane) = 0.0;
                T0.at(lane) = 0.0;
                T1.at(lane) = PXP.at(lane);
            });
            const double gp0 = S(k, L_G + 4), gp1 = S(k, L_G + 5), h44 = S(k, L_H + 2), h55 = S(k, L_H + 3);
            const double l1 = S(k + 1, L_DV), l2 = S(k, L_DR);
            // ---- the value function of node k + 1 through the stage
            c.mfma(PXX, RA, T0);       // T0 = Pxx A
            c.mfma(PXX, RB, T1);       // T1 = Pxx B + Pxp
            c.mfma(PXP, RB, M);        // M  = Luu + Pxp'B ...
            c.mfma(RA, PX, QX);        // qx = lx + A'px
            c.mfma(RB, PX, QU);        // qu = lu + B'px        (+ pp below)
            c.mfma(RB, T1, M);         // ... + B'T1 = Quu - Ppp
            c.mfma(T1, RA, QUX);       // Qux = T1'A
            c.mfma(RA, T0, QXX);       // Qxx = Lxx + A'T0
            // ---- 2x2 control block (uniform)
            const double ha = c.lane_get(M, 0) + pp00, hb = c.lane_get(M, 1) + pp01, hc = c.lane_get(M, 17) + pp11;
            const double hu0 = c.lane_get(QU, 0) + ppv0, hu1 = c.lane_get(QU, 16) + ppv1;
            const double det = ha * hc - hb * hb;
            // (one combined condition, no short-circuit branches: a single branch on the sweep's critical path)
            if (!((ha > 0.0) & (hc > 0.0) & (det > 1e-14 * ha * hc))) return false;
            const double idet = frcp(det);
            const double i00 = hc * idet, i01 = -hb * idet, i11 = ha * idet;
            const double kb0 = -(i00 * hu0 + i01 * hu1), kb1 = -(i01 * hu0 + i11 * hu1);   // feed-forward of the base block
            // ---- Woodbury terms of the two stiff rows (scalars; see the head of this function)
            const double y00 = dt * i00, y01 = i01, y10 = dt * i01, y11 = i11;          // Y = G Fu, Fu = diag(dt, 1)
            double k00, k01, k11;                                                       // kap
            {
                const double a = sqrt(l1), b = sqrt(l2);
                const double m00 = 1.0 + l1 * (dt * y00), m01 = a * b * (dt * y01), m11 = 1.0 + l2 * y11;
                const double rdm = frcp(m00 * m11 - m01 * m01);
                k00 = l1 * m11 * rdm;
                k01 = -(a * b) * m01 * rdm;
                k11 = l2 * m00 * rdm;
            }
            const double t0 = k00 * (dt * kb0) + k01 * kb1, t1 = k01 * (dt * kb0) + k11 * kb1;   // kap Fu' kb
            const double kf0 = kb0 - (y00 * t0 + y01 * t1), kf1 = kb1 - (y10 * t0 + y11 * t1);
            const double kp00 = i00 * h44, kp01 = i01 * h55, kp10 = i01 * h44, kp11 = i11 * h55;     // K0p = G diag(h)
            const double g14 = dt * kp00, g15 = dt * kp01, g24 = kp10, g25 = kp11 - 1.0;           // p-part of g1, g2
            const double z00 = y00 * k00 + y01 * k01, z01 = y00 * k01 + y01 * k11;      // Z = Y kap
            const double z10 = y10 * k00 + y11 * k01, z11 = y10 * k01 + y11 * k11;
            const double m14 = k00 * g14 + k01 * g24, m24 = k01 * g14 + k11 * g24;      // p-part of kap [g1 g2]'
            const double m15 = k00 * g15 + k01 * g25, m25 = k01 * g15 + k11 * g25;
            // ---- matrix-core part of the stage
            PerLane<double> NWA4, Ya4, GB4, KAP4, GA, AOP, BXX, BXP, KFV, ZN, PXXn, PXPn, PXn, KX;
            // (the operands are written as sums of products with 0 / 1 weights of the lane's position, not as select chains over
            // the scalars: a chain of selects between values a lambda captured by reference becomes a select of ADDRESSES plus a
            // load, which puts every captured scalar - and with them the solver object - into scratch memory)
            c.lanes([&](int lane_) {
                const int lane = c.opaque(lane_);
                const int hi = lane >> 4, lo = lane & 3;
                auto w = [&](int r, int cc) { return (hi == r && lo == cc) ? 1.0 : 0.0; };
                const double e00 = w(0, 0), e01 = w(0, 1), e02 = w(0, 2), e03 = w(0, 3), e10 = w(1, 0), e11 = w(1, 1), e12 = w(1, 2), e13 = w(1, 3);
                const double e20 = w(2, 0), e21 = w(2, 1), e30 = w(3, 0), e31 = w(3, 1);
                // -adj(Quu) with its two columns repeated: Ya4 = -adj(Quu) Qux in rows 0, 1 and again in rows 2, 3
                NWA4.at(lane_) = (e01 + e03 + e10 + e12) * hb - (e00 + e02) * hc - (e11 + e13) * ha;
                Ya4.at(lane_) = 0.0;
                // kap, placed so that kap [g1 g2]'_x comes out in rows 2, 3
                KAP4.at(lane_) = e02 * k00 + (e03 + e12) * k01 + e13 * k11;
                GA.at(lane_) = 0.0;
                // b operand of Pxp': [adj(Quu) diag(h) (rows 0, 1) ; p-part of [g1 g2]' (rows 2, 3)], columns 0, 1
                BXP.at(lane_) = e00 * (hc * h44) - e01 * (hb * h55) - e10 * (hb * h44) + e11 * (ha * h55) + e20 * g14 + e21 * g15 + e30 * g24 + e31 * g25;
                KFV.at(lane_) = e00 * kf0 + e10 * kf1;
                // -Z' (a operand of the gains' correction K = K0 - Z [g1 g2]')
                ZN.at(lane_) = -(e00 * z00 + e01 * z10 + e10 * z01 + e11 * z11);
                PXXn.at(lane_) = QXX.at(lane_);
                PXPn.at(lane_) = 0.0;
                PXn.at(lane_) = QX.at(lane_) + e30 * t0;      // ... + Fx (kap Fu'kb), x-part
                // what the later passes of the iteration need of the stage's scalars (stored here, before the products, so that the
                // two dozen wave-uniform values of the Woodbury algebra are dead while the matrix core works)
                if (lane_ == 0) {
                    S(k, L_KF + 0, kf0);
                    S(k, L_KF + 1, kf1);
                    S(k, L_KP + 0, kp00 - (z00 * g14 + z01 * g24));
                    S(k, L_KP + 1, kp01 - (z00 * g15 + z01 * g25));
                    S(k, L_KP + 2, kp10 - (z10 * g14 + z11 * g24));
                    S(k, L_KP + 3, kp11 - (z10 * g15 + z11 * g25));
                    S(k, L_IH + 0, i00 - (z00 * y00 + z01 * y01));
                    S(k, L_IH + 1, i01 - (z00 * y10 + z01 * y11));
                    S(k, L_IH + 2, i11 - (z10 * y10 + z11 * y11));
                    S(k, L_Z + 0, z00);
                    S(k, L_Z + 1, z01);
                    S(k, L_Z + 2, z10);
                    S(k, L_Z + 3, z11);
                    S(k, L_T + 0, t0);
                    S(k, L_T + 1, t1);
                    S(k, L_M + 4, m14);
                    S(k, L_M + 5, m15);
                    S(k, L_M + 10, m24);
                    S(k, L_M + 11, m25);
                }
            });
            // ---- the value function's 2x2 part: Ppp' = diag(h) - diag(h) G diag(h) + gp kap gp',  pp' = gp - diag(h) kf - t1 e_p1
            pp00 = h44 - h44 * kp00 + (g14 * m14 + g24 * m24);
            pp01 = -h44 * kp01 + (g14 * m15 + g24 * m25);
            pp11 = h55 - h55 * kp11 + (g15 * m15 + g25 * m25);
            ppv0 = gp0 - h44 * kf0;
            ppv1 = gp1 - h55 * kf1 - t1;
            c.mfma(NWA4, QUX, Ya4);
            c.mfma(QUX, KFV, PXn);          // px' = qx + Qux'kf (+ t0 e_v)
            c.lanes([&](int lane_) {
                const int lane = c.opaque(lane_);
                const int hi = lane >> 4, lo = lane & 3;
                const bool ev = (hi & 1) == 0;
                // [g1 g2]'_x in rows 0, 1 and again in rows 2, 3: g1 = dt K0x(0, .) + e_v, g2 = K0x(1, .), K0x = Ya4 / det
                const double k0 = idet * Ya4.at(lane_);
                GB4.at(lane_) = (ev ? dt : 1.0) * k0 + ((ev && lo == 3) ? 1.0 : 0.0);
                KX.at(lane_) = k0;
            });
            c.mfma(KAP4, GB4, GA);          // rows 2, 3: kap [g1 g2]'_x
            c.mfma(ZN, GB4, KX);            // rows 0, 1: Kx = K0x - Z [g1 g2]'_x
            c.lanes([&](int lane_) {
                const int lane = c.opaque(lane_);
                const int hi = lane >> 4;
                AOP.at(lane_) = idet * QUX.at(lane_) + GA.at(lane_);        // Qux / det in rows 0, 1 (GA is zero there), GA in rows 2, 3
                BXX.at(lane_) = hi < 2 ? Ya4.at(lane_) : GB4.at(lane_);
            });
            c.mfma(AOP, BXX, PXXn);         // Pxx' = Qxx + Qux'K0x + gx kap gx'
            c.mfma(AOP, BXP, PXPn);         // Pxp' = Qux'G diag(h) + gx kap gp'
            c.lanes([&](int lane) {
                PXX.at(lane) = PXXn.at(lane);
                PXP.at(lane) = PXPn.at(lane);
                PX.at(lane) = PXn.at(lane);
                const int so = q_st.at(lane);
                const double v = so >= L_M ? GA.at(lane) : KX.at(lane);
                if (so >= 0) S(k, so, v);
            });
        }
        c.phase([&](int) {});
        return true;
    }

    // ---- second solve with the same factorisation: only the feed-forward terms change -----------------------------
    // Round 6: on the matrix core like the factorisation (until round 5 wave-uniform scalar code, 96 instructions and 35 LDS
    // words per stage in every lane).  h = g_x + A'p and hu = g_u + B'p are the factorisation's own two products, the new
    // value-function gradient p' = h + Kx'hu a third with the stored gains as A operand; the 2-dimensional rest (kf = -IH hu,
    // t = -Z'hu, pp' = g_p + Kp'hu) stays scalar.
    MPC_HD void resolve_gradient() {
        double pT[4];
        terminal_gradient(pT);
        PerLane<int> r_a, r_b, r_lx, r_lu, r_kx;
        PerLane<double> PX, E00, E10;
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int hi = lane >> 4, lo = lane & 3;
            const int zero = L_G + 0;
            r_a.at(lane_) = f_word(hi, lo);
            r_b.at(lane_) = f_word(hi, lo < 2 ? 6 + lo : 4);
            r_lx.at(lane_) = lo == 0 ? L_G + hi : zero;
            r_lu.at(lane_) = (lo == 0 && hi < 2) ? L_G + 6 + hi : zero;
            r_kx.at(lane_) = hi < 2 ? L_KX + 4 * hi + lo : zero;            // Kx[hi][lo]: the A operand of Kx'hu
            const double c0 = lo == 0 ? 1.0 : 0.0;
            E00.at(lane_) = (hi == 0 && lo == 0) ? 1.0 : 0.0;
            E10.at(lane_) = (hi == 1 && lo == 0) ? 1.0 : 0.0;
            PX.at(lane_) = c0 * ((hi == 0 ? 1.0 : 0.0) * pT[0] + (hi == 1 ? 1.0 : 0.0) * pT[1] + (hi == 2 ? 1.0 : 0.0) * pT[2] + (hi == 3 ? 1.0 : 0.0) * pT[3]);
        });
        double pp0 = 0.0, pp1 = 0.0;
#pragma unroll 1
        for (int k = N - 1; k >= 0; --k) {
            PerLane<double> RA, RB, QX, QU, KX, HU;
            c.lanes([&](int lane) {
                const int base = k * L_SLOTS;
                RA.at(lane) = c.ld(base + r_a.at(lane));
                RB.at(lane) = c.ld(base + r_b.at(lane));
                QX.at(lane) = c.ld(base + r_lx.at(lane));
                QU.at(lane) = c.ld(base + r_lu.at(lane));
                KX.at(lane) = c.ld(base + r_kx.at(lane));
            });
            const double i00 = S(k, L_IH + 0), i01 = S(k, L_IH + 1), i11 = S(k, L_IH + 2);
            const double z00 = S(k, L_Z + 0), z01 = S(k, L_Z + 1), z10 = S(k, L_Z + 2), z11 = S(k, L_Z + 3);
            const double kp00 = S(k, L_KP + 0), kp01 = S(k, L_KP + 1), kp10 = S(k, L_KP + 2), kp11 = S(k, L_KP + 3);
            const double gp0 = S(k, L_G + 4), gp1 = S(k, L_G + 5);
            c.mfma(RA, PX, QX);        // h  = g_x + A'p
            c.mfma(RB, PX, QU);        // hu = g_u + B'p        (+ pp below)
            const double hu0 = c.lane_get(QU, 0) + pp0, hu1 = c.lane_get(QU, 16) + pp1;
            c.lanes([&](int lane) {
                HU.at(lane) = E00.at(lane) * hu0 + E10.at(lane) * hu1;
                if (lane == 0) {
                    S(k, L_KF + 0, -(i00 * hu0 + i01 * hu1));
                    S(k, L_KF + 1, -(i01 * hu0 + i11 * hu1));
                    S(k, L_T + 0, -(z00 * hu0 + z10 * hu1));
                    S(k, L_T + 1, -(z01 * hu0 + z11 * hu1));
                }
            });
            c.mfma(KX, HU, QX);        // p' = h + Kx'hu
            c.lanes([&](int lane) { PX.at(lane) = QX.at(lane); });
            pp0 = gp0 + kp00 * hu0 + kp10 * hu1;
            pp1 = gp1 + kp01 * hu0 + kp11 * hu1;
        }
        c.phase([&](int) {});
    }

    // ---- Newton direction from the gains: L_DU of every stage, L_DX of every node -----------------------------------
    // Round 6: on the matrix core (until round 5 wave-uniform scalar code, 52 instructions per stage).  The node's step d
    // (column 0 of a block) and the previous stage's control step dp go through four products per stage -
    //     du = kf + Kx d + Kp dp        d' = A d + B du
    // with the gains and the model TRANSPOSED as A operands (role tables of LDS words, nothing is moved), and each lane of
    // blocks 0 / 1 stores its component of d' / du with one write.
    MPC_HD void forward_sweep() {
        PerLane<int> r_at, r_bt, r_kxt, r_kpt, r_kf, r_st;
        PerLane<double> D, DP, WB0, WB1;
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
            const int zero = L_G + 0;
            r_at.at(lane_) = f_word(lo, hi);                                       // A[lo][hi]
            r_bt.at(lane_) = hi < 2 ? f_word(lo, 6 + hi) : zero;                   // B~[lo][hi]
            r_kxt.at(lane_) = lo < 2 ? L_KX + 4 * lo + hi : zero;                  // Kx[lo][hi]
            r_kpt.at(lane_) = (lo < 2 && hi < 2) ? L_KP + 2 * lo + hi : zero;      // Kp[lo][hi]
            r_kf.at(lane_) = (lo == 0 && hi < 2) ? L_KF + hi : zero;
            // block 0, column 0: component hi of d' -> L_DX of node k + 1; block 1, column 0, rows 0, 1: du -> L_DU of stage k
            int out = -1;
            out = (blk == 0 && lo == 0) ? L_SLOTS + L_DX + hi : out;
            out = (blk == 1 && lo == 0 && hi < 2) ? L_DU + hi : out;
            r_st.at(lane_) = out;
            WB0.at(lane_) = blk == 1 ? 0.0 : 1.0;
            WB1.at(lane_) = blk == 1 ? 1.0 : 0.0;
            D.at(lane_) = 0.0;
            DP.at(lane_) = 0.0;
        });
        c.phase([&](int lane) {
            if (lane < 4) S(0, L_DX + lane, 0.0);
        });
#pragma unroll 1
        for (int k = 0; k < N; ++k) {
            PerLane<double> AT, BT, KXT, KPT, DU, DN;
            c.lanes([&](int lane) {
                const int base = k * L_SLOTS;
                AT.at(lane) = c.ld(base + r_at.at(lane));
                BT.at(lane) = c.ld(base + r_bt.at(lane));
                KXT.at(lane) = c.ld(base + r_kxt.at(lane));
                KPT.at(lane) = c.ld(base + r_kpt.at(lane));
                DU.at(lane) = c.ld(base + r_kf.at(lane));
                DN.at(lane) = 0.0;
            });
            c.mfma(KXT, D, DU);        // du = kf + Kx d
            c.mfma(AT, D, DN);         // d' = A d
            c.mfma(KPT, DP, DU);       //      ... + Kp dp
            c.mfma(BT, DU, DN);        //      ... + B du
            c.lanes([&](int lane) {
                const int so = r_st.at(lane);
                const double v = WB0.at(lane) * DN.at(lane) + WB1.at(lane) * DU.at(lane);      // (exact: products with 0 / 1)
                if (so >= 0) S(k, so, v);
                D.at(lane) = DN.at(lane);
                DP.at(lane) = DU.at(lane);
            });
        }
        c.phase([&](int) {});
    }

    // sincos_b of mpc_core.hpp with every 64-bit literal behind CTX::fresh: used twice per linearisation pass only, but as plain
    // literals the fifteen constants are hoisted out of the pass loop and sit in thirty vector registers through every
    // iteration (the throughput build then spills)
    MPC_HD void sincos_f(double x, double &sn, double &cs) const {
        auto F = [&](double v) { return c.fresh(v); };
        const double n = rint(x * F(6.36619772367581382433e-01));
        double r = fma(-n, F(1.57079632679489655800e+00), x);
        r = fma(-n, F(6.12323399573676603587e-17), r);
        const double z = r * r;
        const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, F(1.58969099521155010221e-10), F(-2.50507602534068634195e-08)),
                                                  F(2.75573137070700676789e-06)),
                                           F(-1.98412698298579493134e-04)),
                                    F(8.33333333332248946124e-03)),
                             F(-1.66666666666666324348e-01));
        const double sr = fma(z * r, ps, r);
        const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, F(-1.13596475577881948265e-11), F(2.08757232129817482790e-09)),
                                                  F(-2.75573143513906633035e-07)),
                                           F(2.48015872894767294178e-05)),
                                    F(-1.38888888888741095749e-03)),
                             F(4.16666666666666019037e-02));
        const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
        const int q = ((int)n) & 3;
        const double sa = (q & 1) ? cr : sr, ca = (q & 1) ? sr : cr;
        sn = (q & 2) ? -sa : sa;
        cs = ((q + 1) & 2) ? -ca : ca;
    }

    // ---- the solve.  On entry L_U holds the stored profile (oa, od); on exit (status 0) the new one, L_X the
    //      predicted states of the linear model ----------------------------------------------------------------
    MPC_HD void solve(int &status_out, int &iters_out) {
        status_out = ST_MAX_ITER;
        iters_out = 0;
        if (!(X0(3) >= 0.0) || !(X0(3) <= kMaxSpeed)) {   // x[2, 0] == v0 against 0 <= x[2, t] <= MAX_SPEED (:252-256)
            status_out = ST_INFEASIBLE;
            return;
        }
        c.st(SCR + SC_ZERO, 0.0);
        c.st(SCR + SC_ONE, 1.0);
        c.st(SCR + SC_DT, dt);
        c.phase([&](int lane) {
            for (int k = lane; k <= N; k += kLanes) {
                S(k, L_ONE, 1.0);
                S(k, L_DTC, dt);
            }
        });
        // ---- nominal trajectory (predict_motion, :84-110); only its speed and yaw enter the model
        c.phase([&](int lane) {
            for (int k = lane; k < N; k += kLanes) {
                double sd, cd;
                sincos_f(S(k, L_U + 1), sd, cd);
                S(k, L_G + 0, sd / cd);
            }
        });
        {
            double v = X0(3), yaw = X0(2);
#pragma unroll 1
            for (int k = 0; k < N; ++k) {
                S(k, L_G + 1, v);
                S(k, L_G + 2, yaw);
                v += S(k, L_U + 0) * dt;
                v = fmax2(0.0, fmin2(v, kMaxSpeed));
                yaw += (v * kInvWheelbase) * S(k, L_G + 0) * dt;
            }
        }
        c.phase([&](int) {});
        // ---- linearisation (linear_model_matrix, :62-82, steer_ref = 0), start u = 0
        c.phase([&](int lane) {
            for (int k = lane; k < N; k += kLanes) {
                const double vb = S(k, L_G + 1), yb = S(k, L_G + 2);
                double sy, cy;
                sincos_f(yb, sy, cy);
                S(k, L_LIN + 0, -dt * vb * sy);
                S(k, L_LIN + 1, dt * cy);
                S(k, L_LIN + 2, dt * vb * cy);
                S(k, L_LIN + 3, dt * sy);
                S(k, L_LIN + 4, 0.0);
                S(k, L_LIN + 5, 0.0);
                S(k, L_LIN + 6, 0.0);
                S(k, L_LIN + 7, dt * vb * kInvWheelbase);
                S(k, L_U + 0, 0.0);
                S(k, L_U + 1, 0.0);
            }
        });
        {
            double q0 = X0(0), q1 = X0(1), q2 = X0(2), q3 = X0(3);
#pragma unroll 1
            for (int k = 0; k < N; ++k) {
                S(k, L_X + 0, q0); S(k, L_X + 1, q1); S(k, L_X + 2, q2); S(k, L_X + 3, q3);
                const double n0 = q0 + S(k, L_LIN + 0) * q2 + S(k, L_LIN + 1) * q3;
                const double n1 = q1 + S(k, L_LIN + 2) * q2 + S(k, L_LIN + 3) * q3;
                const double n2 = q2 + S(k, L_LIN + 4) * q3;
                q0 = n0; q1 = n1; q2 = n2;
            }
            S(N, L_X + 0, q0); S(N, L_X + 1, q1); S(N, L_X + 2, q2); S(N, L_X + 3, q3);
        }
        c.phase([&](int) {});
        c.phase([&](int lane) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool on = lane < N && valid(lane, i);
                s_[i].at(lane) = on ? fmax2(cval(lane, i), kSInitMin) : 1.0;
                z_[i].at(lane) = on ? kZInit : 0.0;
            }
        });
        // (wave-uniform values of the whole solve that the vector unit would otherwise hold in registers across every phase: the
        // horizon-derived constants are re-derived where they are used, the dual tolerance sits in LDS)
        c.st(SCR + SC_T, (double)N);
        c.st(SCR + SC_MINEQ, (double)(8 * N - 2));
        c.phase([&](int) {});
        auto m_ineq_ = [&]() { return c.ld(SCR + SC_MINEQ); };
        auto T_ = [&]() { return c.ld(SCR + SC_T); };
        int iter = 0;
        for (iter = 0; iter <= P.max_iter; ++iter) {
            // ============ residuals: primal (per inequality), complementarity, node terms of the adjoint
            const double T = T_();
            const double y0 = 2.0 * T * kQfXY * (S(N, L_X + 0) - xr(N)), y1 = 2.0 * T * kQfXY * (S(N, L_X + 1) - yr(N));
            c.phase([&](int lane) {
                red_a.at(lane) = 0.0;
                red_b.at(lane) = 0.0;
                if (lane >= N) {
                    if (kKeepResidual)
                        for (int i = 0; i < 8; ++i) rp_[i].at(lane) = 0.0;
                    return;
                }
                const int k = lane;
                double rpm = 0.0, sz = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!valid(k, i)) {
                        if (kKeepResidual) rp_[i].at(lane) = 0.0;
                        continue;
                    }
                    const double r = presid(k, i, s_[i].at(lane));
                    if (kKeepResidual) rp_[i].at(lane) = r;
                    rpm = fmax2(rpm, fabs(r));
                    sz += s_[i].at(lane) * z_[i].at(lane);
                }
                red_a.at(lane) = rpm;
                red_b.at(lane) = sz;
                const double zv = z_[6].at(lane) - z_[7].at(lane);
                S(k, L_ZR, z_[5].at(lane) - z_[4].at(lane));
                const int j = k + 1;   // node whose adjoint terms this lane prepares
                if (j < N) {
                    S(j, L_Y + 0, 2.0 * kQyaw * (S(j, L_X + 2) - hr(j)) + S(j, L_LIN + 0) * y0 + S(j, L_LIN + 2) * y1);
                    S(j, L_Y + 1, 2.0 * kQv * (S(j, L_X + 3) - vr(j)) - zv + S(j, L_LIN + 1) * y0 + S(j, L_LIN + 3) * y1);
                } else {
                    S(j, L_Y + 0, 2.0 * T * kQfYaw * (S(j, L_X + 2) - hr(j)));
                    S(j, L_Y + 1, -zv);
                }
            });
            const double res_p = c.wave_max(red_a), mu = c.wave_sum(red_b) / m_ineq_();
            // ============ adjoint of (yaw, v): x and y carry y0, y1 unchanged, so the yaw adjoint is a suffix sum of the node
            //              terms and the speed adjoint one of (term + a23 yaw-adjoint of the next node): two wave scans,
            //              lane j = node j + 1
            {
                PerLane<double> sc;
                c.phase([&](int lane) { sc.at(lane) = lane < N ? S(lane + 1, L_Y + 0) : 0.0; });
                c.wave_suffix_sum(sc);
                c.phase([&](int lane) {
                    if (lane < N) S(lane + 1, L_Y + 0, sc.at(lane));
                });
                c.phase([&](int lane) {
                    const int k = lane + 1;
                    double h = 0.0;
                    if (lane < N) {
                        h = S(k, L_Y + 1);
                        if (k < N) h += S(k, L_LIN + 4) * S(k + 1, L_Y + 0);
                    }
                    sc.at(lane) = h;
                });
                c.wave_suffix_sum(sc);
                c.phase([&](int lane) {
                    if (lane < N) S(lane + 1, L_Y + 1, sc.at(lane));
                });
            }
            // ============ dual residual of the controls
            c.phase([&](int lane) {
                red_a.at(lane) = 0.0;
                if (lane >= N) return;
                const int k = lane;
                const double a = S(k, L_U + 0), d = S(k, L_U + 1);
                double r0 = 2.0 * kR0 * a - (z_[0].at(lane) - z_[1].at(lane)) + dt * S(k + 1, L_Y + 1);
                double r1 = 2.0 * kR1 * d - (z_[2].at(lane) - z_[3].at(lane)) - S(k, L_ZR) + S(k, L_LIN + 7) * S(k + 1, L_Y + 0) +
                            S(k, L_LIN + 5) * y0 + S(k, L_LIN + 6) * y1;
                if (k >= 1) {
                    r0 += 2.0 * kRd0 * (a - S(k - 1, L_U + 0));
                    r1 += 2.0 * kRd1 * (d - S(k - 1, L_U + 1));
                }
                if (k + 1 < N) {
                    r0 -= 2.0 * kRd0 * (S(k + 1, L_U + 0) - a);
                    r1 += -2.0 * kRd1 * (S(k + 1, L_U + 1) - d) + S(k + 1, L_ZR);
                }
                red_a.at(lane) = fmax2(fabs(r0), fabs(r1));
            });
            const double res_d = c.wave_max(red_a);
            // all three at the same iterate (the steps of the stiff rows are computed in constraint space, see darg: the
            // measured dual residual keeps falling to ~1e-12 instead of drowning in rounding noise z^2 eps / mu)
            if (iter == 0) c.st(SCR + SC_TOLD, kTolDRel * fmax2(1e3, res_d));
            const bool dual_ok = res_d <= c.ld(SCR + SC_TOLD);
            if (res_p <= kTolP && dual_ok && mu <= kTolMu) {
                status_out = ST_CONVERGED;
                break;
            }
            if (iter == P.max_iter) break;

            // ============ predictor: right-hand side q = -z r_p / s, barrier weights z / s
            c.phase([&](int lane) {
                if (lane >= N) return;
                const int k = lane;
                double q[8], D[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double rs = frcp(s_[i].at(lane));
                    D[i] = z_[i].at(lane) * rs;            // 0 for the two absent rows of stage 0
                    q[i] = kKeepResidual ? -D[i] * rp_[i].at(lane) : (valid(k, i) ? -D[i] * presid(k, i, s_[i].at(lane)) : 0.0);
                }
                put_own_gradient(k, q);
                S(k + 1, L_DV, D[6] + D[7]);
                const double rd0 = k >= 1 ? 2.0 * kRd0 : 0.0, rd1 = k >= 1 ? 2.0 * kRd1 : 0.0, Dr = D[4] + D[5];
                S(k, L_H + 2, rd0);
                S(k, L_H + 3, rd1);
                S(k, L_DR, Dr);
                S(k, L_H + 4, 2.0 * kR0 + rd0 + D[0] + D[1]);
                S(k, L_H + 5, 2.0 * kR1 + rd1 + D[2] + D[3]);
            });
            c.phase([&](int lane) {
                if (lane >= N) return;
                const int k = lane;
                put_state_gradient(k);
                S(k, L_H + 0, k >= 1 ? 2.0 * kQyaw : 0.0);
                S(k, L_H + 1, k >= 1 ? 2.0 * kQv : 0.0);
            });
            if (!factor_and_solve()) {
                status_out = ST_FACTORIZATION;
                break;
            }
            forward_sweep();
            // ============ predictor step: largest step to the boundary and the complementarity it would leave
            double sigma_mu;
            {
                double rn, rd, sA, sB, sC;
                c.phase([&](int lane) {
                    double bn = 0.0, bd = 1.0, a1 = 0.0, a2 = 0.0, a0 = 0.0;
                    // (rp_ and pr_ are written on every path: they are live from here to the update of the iterate only)
                    if (lane < N) {
                        const int k = lane;
                        double dr, dv;
                        stiff_steps(k, dr, dv);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (!valid(k, i)) {
                                rp_[i].at(lane) = 0.0;
                                pr_[i].at(lane) = 0.0;
                                continue;
                            }
                            const double s = s_[i].at(lane), z = z_[i].at(lane);
                            const double rp = kKeepResidual ? rp_[i].at(lane) : presid(k, i, s);
                            rp_[i].at(lane) = rp;
                            const double ds = csign(i) * darg(k, i, dr, dv) + rp;
                            const double dz = -z - z * ds * frcp(s);
                            pr_[i].at(lane) = ds * dz;
                            if (wave::ratio_greater(-ds, s, bn, bd)) { bn = -ds; bd = s; }
                            if (wave::ratio_greater(-dz, z, bn, bd)) { bn = -dz; bd = z; }
                            a0 += s * z;
                            a1 += s * dz + z * ds;
                            a2 += ds * dz;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            rp_[i].at(lane) = 0.0;
                            pr_[i].at(lane) = 0.0;
                        }
                    }
                    red_a.at(lane) = bn;
                    red_b.at(lane) = bd;
                    red_c.at(lane) = a0;
                    red_d.at(lane) = a1;
                    red_e.at(lane) = a2;
                });
                c.wave_max_ratio(red_a, red_b, rn, rd);
                sA = c.wave_sum(red_c);
                sB = c.wave_sum(red_d);
                sC = c.wave_sum(red_e);
                const double a_aff = (rn > rd) ? rd / rn : 1.0;
                const double mu_aff = (sA + a_aff * (sB + a_aff * sC)) / m_ineq_();
                const double ratio = mu_aff / mu;
                sigma_mu = fmax2(ratio * ratio * ratio * mu, 0.1 * kTolMu);   // never aim below the stopping threshold
            }
            // ============ corrector: q = (sigma mu - ds_aff dz_aff - z r_p) / s, same factorisation
            c.phase([&](int lane) {
                if (lane >= N) return;
                const int k = lane;
                double q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double rs = frcp(s_[i].at(lane));
                    q[i] = valid(k, i) ? (sigma_mu - pr_[i].at(lane) - z_[i].at(lane) * rp_[i].at(lane)) * rs : 0.0;
                }
                put_own_gradient(k, q);
            });
            c.phase([&](int lane) {
                if (lane < N) put_state_gradient(lane);
            });
            resolve_gradient();
            forward_sweep();
            // ============ step: one common length, 0.99 of the way to the boundary at most
            double alpha;
            {
                double rn, rd;
                c.phase([&](int lane) {
                    double bn = 0.0, bd = 1.0;
                    if (lane < N) {
                        const int k = lane;
                        double dr, dv;
                        stiff_steps(k, dr, dv);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (!valid(k, i)) {
                                rp_[i].at(lane) = 0.0;
                                pr_[i].at(lane) = 0.0;
                                continue;
                            }
                            const double s = s_[i].at(lane), z = z_[i].at(lane);
                            const double ds = csign(i) * darg(k, i, dr, dv) + rp_[i].at(lane);
                            const double dz = (sigma_mu - pr_[i].at(lane) - s * z - z * ds) * frcp(s);
                            rp_[i].at(lane) = ds;     // the residual is not needed any more this iteration
                            pr_[i].at(lane) = dz;
                            if (wave::ratio_greater(-ds, s, bn, bd)) { bn = -ds; bd = s; }
                            if (wave::ratio_greater(-dz, z, bn, bd)) { bn = -dz; bd = z; }
                        }
                    }
                    red_a.at(lane) = bn;
                    red_b.at(lane) = bd;
                });
                c.wave_max_ratio(red_a, red_b, rn, rd);
                alpha = (0.99 * rd < rn) ? 0.99 * rd / rn : 1.0;
            }
            c.phase([&](int lane) {
                if (lane >= N) return;
                const int k = lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s_[i].at(lane) += alpha * rp_[i].at(lane);
                    z_[i].at(lane) += alpha * pr_[i].at(lane);
                }
                S(k, L_U + 0, S(k, L_U + 0) + alpha * S(k, L_DU + 0));
                S(k, L_U + 1, S(k, L_U + 1) + alpha * S(k, L_DU + 1));
                for (int e = 0; e < 4; ++e) S(k + 1, L_X + e, S(k + 1, L_X + e) + alpha * S(k + 1, L_DX + e));
            });
        }
        iters_out = iter;
    }

};

}  // namespace ltv
}  // namespace mpc
