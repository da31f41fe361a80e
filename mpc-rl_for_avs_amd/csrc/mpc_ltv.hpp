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
// input-difference cost couple u_t with u_{t-1}): the 8x8 stage block [x 4 | u_{t-1} 2 | u_t 2] lives in the C/D
// layout of v_mfma_f64_4x4x4f64 exactly as in mpc_wave.hpp (same operand roles, same nine matrix-core
// instructions per stage); the corrector reuses the factorisation through a scalar gradient-only sweep.
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
    L_SLOTS = 75
};
enum : int { SC_ZERO = 0, SC_ONE = 1, SC_DT = 2, SC_X0 = 4, SC_SIZE = 8 };   // SC_X0: the ego state x, y, yaw, v

MPC_HD constexpr int lds_doubles(int N) { return L_SLOTS * (N + 1) + SC_SIZE; }

struct LtvParams {
    int N;
    int max_iter;
    int passes;   // linearisation passes per call (the loop at agents/pure_mpc_linear.py:189)
    double dt;
};

// CTX::kRelax (absent: 0) - what the build may keep in registers (mpc_engine.hip: the two builds of mpc_ltv_kernel):
//   bit 3: the primal residuals stay in registers from the residual phase to the predictor step instead of being recomputed
//   bit 4: the gain rows g1, g2 of a Riccati stage are formed once as wave-uniform values instead of per lane
template <class CTX, class = void>
struct relax_bits { static constexpr int value = 0; };
template <class CTX>
struct relax_bits<CTX, decltype((void)CTX::kRelax)> { static constexpr int value = CTX::kRelax; };

template <class CTX>
struct Solver {
    static constexpr bool kKeepResidual = (relax_bits<CTX>::value & 8) != 0;
    static constexpr bool kUniformGains = (relax_bits<CTX>::value & 16) != 0;
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
    // matrix-core roles, as in mpc_wave.hpp: lane l = 16 hi + 4 (2 I + J) + lo holds element (4 I + hi, 4 J + lo)
    PerLane<int> m_row, m_col, m_fa0, m_fa1, m_fb0, m_fb1;
    PerLane<int> m_lslot;   // stage-Hessian slot of this lane's element, + 256 where it enters with a minus sign; -1: none

    // reference columns of node k: window row min(target + k, M - 1) (:178-187)
    MPC_HD double xr(int k) const { return c.ref(k, R_X); }
    MPC_HD double yr(int k) const { return c.ref(k, R_Y); }
    MPC_HD double hr(int k) const { return c.ref(k, R_H); }
    MPC_HD double vr(int k) const { return c.refv(k); }

    MPC_HD int f_word(int r, int cl) const {
        return wave::stage_transition_word(r, cl, L_LIN, -(SCR + SC_ZERO + 1), -(SCR + SC_ONE + 1), -(SCR + SC_DT + 1));
    }
    // (pure functions of the lane id, recomputed at the start of every factorisation from a lane id the compiler cannot
    // see through - as in mpc_wave.hpp - so that their 11 registers are live during the sweep only)
    MPC_HD void set_roles() {
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int hi = lane >> 4, blk = (lane >> 2) & 3, I = blk >> 1, J = blk & 1, lo = lane & 3;
            const int row = 4 * I + hi, col = 4 * J + lo;
            m_row.at(lane) = row;
            m_col.at(lane) = col;
            m_fa0.at(lane) = f_word(0 + hi, 4 * I + lo);
            m_fa1.at(lane) = f_word(4 + hi, 4 * I + lo);
            m_fb0.at(lane) = f_word(0 + hi, 4 * J + lo);
            m_fb1.at(lane) = f_word(4 + hi, 4 * J + lo);
            const int a = row < col ? row : col, b = row < col ? col : row;
            int slot = -1;
            slot = (a == b && a >= 2) ? L_H + (a - 2) : slot;
            slot = (a == 4 && b == 6) ? 256 + L_H + 2 : slot;
            slot = (a == 5 && b == 7) ? 256 + L_H + 3 : slot;
            m_lslot.at(lane) = slot;
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

    // ---- Riccati factorisation + first solve on the matrix core (operands as in mpc_wave.hpp) -------------------
    MPC_HD bool factor_and_solve() {
        set_roles();
        double pT[4];
        terminal_gradient(pT);
        const double T = (double)N;
        PerLane<double> Pd, pvd;
        c.lanes([&](int lane) {
            const int r = m_row.at(lane), cl = m_col.at(lane);
            double pe = 0.0;
            if (r == cl && r < 3) pe = r < 2 ? 2.0 * T * kQfXY : 2.0 * T * kQfYaw;
            Pd.at(lane) = pe;
            pvd.at(lane) = (cl == 0 && r < 4) ? (r == 0 ? pT[0] : (r == 1 ? pT[1] : (r == 2 ? pT[2] : pT[3]))) : 0.0;
        });
#pragma unroll 1
        for (int k = N - 1; k >= 0; --k) {
            PerLane<double> FA0, FA1, FB0, FB1, Hm, hv;
            c.lanes([&](int lane) {
                const int base = k * L_SLOTS;
                const int w0 = m_fa0.at(lane), w1 = m_fa1.at(lane), w2 = m_fb0.at(lane), w3 = m_fb1.at(lane);
                FA0.at(lane) = c.ld(w0 >= 0 ? base + w0 : -w0 - 1);
                FA1.at(lane) = c.ld(w1 >= 0 ? base + w1 : -w1 - 1);
                FB0.at(lane) = c.ld(w2 >= 0 ? base + w2 : -w2 - 1);
                FB1.at(lane) = c.ld(w3 >= 0 ? base + w3 : -w3 - 1);
                const int ls = m_lslot.at(lane);
                const double lv = c.ld(base + (ls >= 0 ? (ls & 255) : 0));
                Hm.at(lane) = ls < 0 ? 0.0 : (ls >= 256 ? -lv : lv);
                const bool col0 = m_col.at(lane) == 0;
                const double gv = c.ld(base + (col0 ? L_G + m_row.at(lane) : 0));
                hv.at(lane) = col0 ? gv : 0.0;
            });
            // T = P F   (P symmetric: block (K, I) in the C/D layout is block (I, K) as A operand)
            PerLane<double> PA0, PA1, Tm;
            c.template take_blocks<wave::BM_K0_I>(PA0, Pd);
            c.template take_blocks<wave::BM_K1_I>(PA1, Pd);
            c.lanes([&](int lane) { Tm.at(lane) = 0.0; });
            c.mfma(PA0, FB0, Tm);
            c.mfma(PA1, FB1, Tm);
            // H = L + F' T,  h = l + F' p   (F in the C/D layout is F' as A operand)
            PerLane<double> TB0, TB1, pB0, pB1;
            c.template take_blocks<wave::BM_K0_J>(TB0, Tm);
            c.template take_blocks<wave::BM_K1_J>(TB1, Tm);
            c.template take_blocks<wave::BM_K0_J>(pB0, pvd);
            c.template take_blocks<wave::BM_K1_J>(pB1, pvd);
            c.mfma(FA0, TB0, Hm);
            c.mfma(FA1, TB1, Hm);
            c.mfma(FA0, pB0, hv);
            c.mfma(FA1, pB1, hv);
            PerLane<double> HB, HA;
            c.template take_blocks<wave::BM_K1_J>(HB, Hm);
            c.template take_blocks<wave::BM_K1_I>(HA, Hm);
            // control block: elements (6,6) (6,7) (7,6) (7,7) in lanes 46 47 62 63, gradient rows 6, 7 in lanes 40, 56
            const double ha = c.lane_get(Hm, 46), hb = 0.5 * (c.lane_get(Hm, 47) + c.lane_get(Hm, 62)),
                         hc = c.lane_get(Hm, 63);
            const double hu0 = c.lane_get(hv, 40), hu1 = c.lane_get(hv, 56);
            const double det = ha * hc - hb * hb;
            // (one combined condition, no short-circuit branches: a single branch on the sweep's critical path)
            if (!((ha > 0.0) & (hc > 0.0) & (det > 1e-14 * ha * hc))) return false;
            // W = adj(Huu) H(u, .) on the matrix core while the reciprocal of the determinant is computed
            PerLane<double> G, nHA, W, kfB;
            c.lanes([&](int lane) {
                const int hi = lane >> 4, lo = lane & 3;
                G.at(lane) = (hi == 2 && lo == 2) ? hc : ((hi == 3 && lo == 3) ? ha : ((hi >= 2 && lo >= 2) ? -hb : 0.0));
                W.at(lane) = 0.0;
            });
            c.mfma(G, HB, W);
            const double idet = frcp(det);
            const double i00 = hc * idet, i01 = -hb * idet, i11 = ha * idet;
            const double kb0 = -(i00 * hu0 + i01 * hu1), kb1 = -(i01 * hu0 + i11 * hu1);   // feed-forward of the base block
            // ---- the two barrier terms that make Riccati recursions cancel catastrophically when their weights grow
            //      like 1/mu - the speed bounds of node k + 1 (weight l1 on f1 = e_v of the next state = [e_v; dt e_u0]
            //      of this stage) and the rate limit of stage k (weight l2 on f2 = e_u1 - e_p1) - are kept out of the
            //      block H above and put back through the Woodbury identity, which never subtracts large numbers:
            //        Huu^-1 = G - Y kap Y',  K = K0 - Y kap g',  P = P0 + g kap g',  kap = (diag(1/l) + Fu' G Fu)^-1,
            //        Y = G Fu,  g = Fx - H(x,u) G Fu = Fx + K0' Fu            (G, K0, P0: the base block's)
            const double l1 = S(k + 1, L_DV), l2 = S(k, L_DR);
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
            const double t0 = k00 * (dt * kb0) + k01 * kb1, t1 = k01 * (dt * kb0) + k11 * kb1;   // kap Fu' kf0
            const double kf0 = kb0 - (y00 * t0 + y01 * t1), kf1 = kb1 - (y10 * t0 + y11 * t1);
            c.lanes([&](int lane) {
                const int hi = lane >> 4;
                nHA.at(lane) = -idet * HA.at(lane);
                kfB.at(lane) = (m_col.at(lane) == 0) ? (hi == 2 ? kf0 : (hi == 3 ? kf1 : 0.0)) : 0.0;
            });
            c.mfma(nHA, W, Hm);      // Hm <- H - H(., u) G H(u, .)   (base block)
            c.mfma(HA, kfB, hv);     // hv <- h + H(., u) kf
            const double h44 = S(k, L_H + 2), h55 = S(k, L_H + 3);
            // base gains K0 = -G H(u, .): state columns from the matrix core's W = adj(Huu) H(u, .) (row 6 + a, column j
            // sits in lane 16 (2 + a) + j), previous-control columns in closed form (H(u, p) = -diag(H44, H55)).
            // g1 = [dt K0(0, .) + e_v], g2 = [K0(1, .) - e_p1] (6 entries each) are needed per lane by row and by column:
            // every lane selects its entries from the eight raw values of W (scalar registers) and scales them itself,
            // instead of 20 wave-uniform doubles being kept in vector registers to the end of the stage.
            if (kUniformGains) {
                double g1[6], g2[6], kx0[4], kx1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kx0[j] = -idet * c.lane_get(W, 32 + j);
                    kx1[j] = -idet * c.lane_get(W, 48 + j);
                    g1[j] = dt * kx0[j];
                    g2[j] = kx1[j];
                }
                g1[3] += 1.0;
                const double kp00 = i00 * h44, kp01 = i01 * h55, kp10 = i01 * h44, kp11 = i11 * h55;
                g1[4] = dt * kp00;
                g1[5] = dt * kp01;
                g2[4] = kp10;
                g2[5] = kp11 - 1.0;
                const double z00 = y00 * k00 + y01 * k01, z01 = y00 * k01 + y01 * k11;      // Z = Y kap
                const double z10 = y10 * k00 + y11 * k01, z11 = y10 * k01 + y11 * k11;
                c.lanes([&](int lane) {
                    const int r = m_row.at(lane), cl = m_col.at(lane);
                    double g1r = 0.0, g2r = 0.0, g1c = 0.0, g2c = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        g1r = r == j ? g1[j] : g1r;
                        g2r = r == j ? g2[j] : g2r;
                        g1c = cl == j ? g1[j] : g1c;
                        g2c = cl == j ? g2[j] : g2c;
                    }
                    const double add = g1r * (k00 * g1c + k01 * g2c) + g2r * (k01 * g1c + k11 * g2c);
                    Pd.at(lane) = (r < 6 && cl < 6) ? Hm.at(lane) + add : 0.0;
                    const double padd = r == 3 ? t0 : (r == 5 ? -t1 : 0.0);                  // Fx (kap Fu' kf0)
                    pvd.at(lane) = (r < 6 && cl == 0) ? hv.at(lane) + padd : 0.0;
                    if (lane < 4) {
                        S(k, L_KX + lane, kx0[lane] - (z00 * g1[lane] + z01 * g2[lane]));
                        S(k, L_KX + 4 + lane, kx1[lane] - (z10 * g1[lane] + z11 * g2[lane]));
                    }
                    if (lane == 0) {
                        S(k, L_KF + 0, kf0);
                        S(k, L_KF + 1, kf1);
                        S(k, L_KP + 0, kp00 - (z00 * g1[4] + z01 * g2[4]));
                        S(k, L_KP + 1, kp01 - (z00 * g1[5] + z01 * g2[5]));
                        S(k, L_KP + 2, kp10 - (z10 * g1[4] + z11 * g2[4]));
                        S(k, L_KP + 3, kp11 - (z10 * g1[5] + z11 * g2[5]));
                        S(k, L_IH + 0, i00 - (z00 * y00 + z01 * y01));
                        S(k, L_IH + 1, i01 - (z00 * y10 + z01 * y11));
                        S(k, L_IH + 2, i11 - (z10 * y10 + z11 * y11));
                        S(k, L_Z + 0, z00);
                        S(k, L_Z + 1, z01);
                        S(k, L_Z + 2, z10);
                        S(k, L_Z + 3, z11);
                        S(k, L_T + 0, t0);
                        S(k, L_T + 1, t1);
                    }
                    if (lane < 6) {
                        S(k, L_M + lane, k00 * g1[lane] + k01 * g2[lane]);
                        S(k, L_M + 6 + lane, k01 * g1[lane] + k11 * g2[lane]);
                    }
                });
                continue;
            }
            double w0[4], w1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w0[j] = c.lane_get(W, 32 + j);
                w1[j] = c.lane_get(W, 48 + j);
            }
            const double kp00 = i00 * h44, kp01 = i01 * h55, kp10 = i01 * h44, kp11 = i11 * h55;
            const double z00 = y00 * k00 + y01 * k01, z01 = y00 * k01 + y01 * k11;      // Z = Y kap
            const double z10 = y10 * k00 + y11 * k01, z11 = y10 * k01 + y11 * k11;
            c.lanes([&](int lane) {
                if (lane == 0) {
                    const double g14 = dt * kp00, g15 = dt * kp01, g24 = kp10, g25 = kp11 - 1.0;
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
                }
            });
            c.lanes([&](int lane) {
                const int r = m_row.at(lane), cl = m_col.at(lane);
                // entry j of g1 / g2 for j = row and j = column of this lane
                double r0 = 0.0, r1 = 0.0, c0 = 0.0, c1 = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r0 = r == j ? w0[j] : r0;
                    r1 = r == j ? w1[j] : r1;
                    c0 = cl == j ? w0[j] : c0;
                    c1 = cl == j ? w1[j] : c1;
                }
                const double kr0 = -idet * r0, kr1 = -idet * r1, kc0 = -idet * c0, kc1 = -idet * c1;   // K0(a, row / col)
                const double g1r = r < 4 ? dt * kr0 + (r == 3 ? 1.0 : 0.0) : (r == 4 ? dt * kp00 : (r == 5 ? dt * kp01 : 0.0));
                const double g2r = r < 4 ? kr1 : (r == 4 ? kp10 : (r == 5 ? kp11 - 1.0 : 0.0));
                const double g1c = cl < 4 ? dt * kc0 + (cl == 3 ? 1.0 : 0.0) : (cl == 4 ? dt * kp00 : (cl == 5 ? dt * kp01 : 0.0));
                const double g2c = cl < 4 ? kc1 : (cl == 4 ? kp10 : (cl == 5 ? kp11 - 1.0 : 0.0));
                const double mc1 = k00 * g1c + k01 * g2c, mc2 = k01 * g1c + k11 * g2c;     // column cl of kap [g1 g2]'
                const double add = g1r * mc1 + g2r * mc2;
                Pd.at(lane) = (r < 6 && cl < 6) ? Hm.at(lane) + add : 0.0;
                const double padd = r == 3 ? t0 : (r == 5 ? -t1 : 0.0);                  // Fx (kap Fu' kf0)
                pvd.at(lane) = (r < 6 && cl == 0) ? hv.at(lane) + padd : 0.0;
                // lanes 0 .. 5 hold row 0, column = lane: their column entries are entry `lane` of g1, g2 and of K0
                if (lane < 4) {
                    S(k, L_KX + lane, kc0 - (z00 * g1c + z01 * g2c));
                    S(k, L_KX + 4 + lane, kc1 - (z10 * g1c + z11 * g2c));
                }
                if (lane < 6) {
                    S(k, L_M + lane, mc1);
                    S(k, L_M + 6 + lane, mc2);
                }
            });
                }
        c.phase([&](int) {});
        return true;
    }

    // ---- second solve with the same factorisation: only the feed-forward terms change (uniform, serial) --------
    MPC_HD void resolve_gradient() {
        double pT[4];
        terminal_gradient(pT);
        double p0 = pT[0], p1 = pT[1], p2 = pT[2], p3 = pT[3], pp0 = 0.0, pp1 = 0.0;
#pragma unroll 1
        for (int k = N - 1; k >= 0; --k) {
            const double a02 = S(k, L_LIN + 0), a03 = S(k, L_LIN + 1), a12 = S(k, L_LIN + 2), a13 = S(k, L_LIN + 3),
                         a23 = S(k, L_LIN + 4), b01 = S(k, L_LIN + 5), b11 = S(k, L_LIN + 6), b21 = S(k, L_LIN + 7);
            const double h0 = S(k, L_G + 0) + p0, h1 = S(k, L_G + 1) + p1;
            const double h2 = S(k, L_G + 2) + a02 * p0 + a12 * p1 + p2;
            const double h3 = S(k, L_G + 3) + a03 * p0 + a13 * p1 + a23 * p2 + p3;
            const double hu0 = S(k, L_G + 6) + dt * p3 + pp0;
            const double hu1 = S(k, L_G + 7) + b01 * p0 + b11 * p1 + b21 * p2 + pp1;
            const double i00 = S(k, L_IH + 0), i01 = S(k, L_IH + 1), i11 = S(k, L_IH + 2);
            S(k, L_KF + 0, -(i00 * hu0 + i01 * hu1));
            S(k, L_KF + 1, -(i01 * hu0 + i11 * hu1));
            S(k, L_T + 0, -(S(k, L_Z + 0) * hu0 + S(k, L_Z + 2) * hu1));
            S(k, L_T + 1, -(S(k, L_Z + 1) * hu0 + S(k, L_Z + 3) * hu1));
            p0 = h0 + S(k, L_KX + 0) * hu0 + S(k, L_KX + 4) * hu1;
            p1 = h1 + S(k, L_KX + 1) * hu0 + S(k, L_KX + 5) * hu1;
            p2 = h2 + S(k, L_KX + 2) * hu0 + S(k, L_KX + 6) * hu1;
            p3 = h3 + S(k, L_KX + 3) * hu0 + S(k, L_KX + 7) * hu1;
            pp0 = S(k, L_G + 4) + S(k, L_KP + 0) * hu0 + S(k, L_KP + 2) * hu1;
            pp1 = S(k, L_G + 5) + S(k, L_KP + 1) * hu0 + S(k, L_KP + 3) * hu1;
        }
    }

    // ---- Newton direction from the gains (uniform, serial): L_DU of every stage, L_DX of every node ------------
    MPC_HD void forward_sweep() {
        double d0 = 0, d1 = 0, d2 = 0, d3 = 0, dp0 = 0, dp1 = 0;
        S(0, L_DX + 0, 0.0); S(0, L_DX + 1, 0.0); S(0, L_DX + 2, 0.0); S(0, L_DX + 3, 0.0);
#pragma unroll 1
        for (int k = 0; k < N; ++k) {
            const double du0 = S(k, L_KF + 0) + S(k, L_KX + 0) * d0 + S(k, L_KX + 1) * d1 + S(k, L_KX + 2) * d2 +
                               S(k, L_KX + 3) * d3 + S(k, L_KP + 0) * dp0 + S(k, L_KP + 1) * dp1;
            const double du1 = S(k, L_KF + 1) + S(k, L_KX + 4) * d0 + S(k, L_KX + 5) * d1 + S(k, L_KX + 6) * d2 +
                               S(k, L_KX + 7) * d3 + S(k, L_KP + 2) * dp0 + S(k, L_KP + 3) * dp1;
            const double a02 = S(k, L_LIN + 0), a03 = S(k, L_LIN + 1), a12 = S(k, L_LIN + 2), a13 = S(k, L_LIN + 3),
                         a23 = S(k, L_LIN + 4), b01 = S(k, L_LIN + 5), b11 = S(k, L_LIN + 6), b21 = S(k, L_LIN + 7);
            const double n0 = d0 + a02 * d2 + a03 * d3 + b01 * du1;
            const double n1 = d1 + a12 * d2 + a13 * d3 + b11 * du1;
            const double n2 = d2 + a23 * d3 + b21 * du1;
            const double n3 = d3 + dt * du0;
            d0 = n0; d1 = n1; d2 = n2; d3 = n3;
            dp0 = du0; dp1 = du1;
            S(k, L_DU + 0, du0);
            S(k, L_DU + 1, du1);
            S(k + 1, L_DX + 0, d0);
            S(k + 1, L_DX + 1, d1);
            S(k + 1, L_DX + 2, d2);
            S(k + 1, L_DX + 3, d3);
        }
        c.phase([&](int) {});
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
        // ---- nominal trajectory (predict_motion, :84-110); only its speed and yaw enter the model
        c.phase([&](int lane) {
            for (int k = lane; k < N; k += kLanes) {
                double sd, cd;
                sincos_b(S(k, L_U + 1), sd, cd);
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
                sincos_b(yb, sy, cy);
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
        const double m_ineq = (double)(8 * N - 2);
        const double T = (double)N;
        int iter = 0;
        double tol_d = 0.0;
        for (iter = 0; iter <= P.max_iter; ++iter) {
            // ============ residuals: primal (per inequality), complementarity, node terms of the adjoint
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
            const double res_p = c.wave_max(red_a), mu = c.wave_sum(red_b) / m_ineq;
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
            if (iter == 0) tol_d = kTolDRel * fmax2(1e3, res_d);
            const bool dual_ok = res_d <= tol_d;
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
                const double mu_aff = (sA + a_aff * (sB + a_aff * sC)) / m_ineq;
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
