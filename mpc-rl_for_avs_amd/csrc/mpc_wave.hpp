// mpc_wave.hpp - wave-cooperative interior-point DDP solver: ONE wave64 per MPC instance.
//
// The NLP is the reference's (agents/pure_mpc.py:80-318), the algorithm the one DESIGN.md section 2 specifies,
// organised so that the 64 lanes of a wave work on a single instance:
//   * everything that is independent per stage (trig and linearisation of the dynamics, cost and collision-potential
//     values and derivatives, barrier terms, complementarity, stage Hessians, dual residual, step-length ratios, dual
//     update) runs stage-parallel, lane k = stage k, with register-level wave reductions;
//   * the Riccati / DDP factorisation stage (state 4 + previous control 2 + control 2) runs on the FP64 matrix cores as
//     twelve 4x4 products per stage, every operand used where the product before it left it, the 2x2 control block as
//     wave-uniform scalar algebra (sweep4 below; until round 4 an 8x8 block form with nine products and eight block moves
//     per stage); nothing is exchanged through LDS;
//   * the line search integrates all four trial step lengths and the linearised Newton step at once: trial t = the 16-lane
//     DPP row t, the lanes of a row = the components of the stage, matrix-vector products as v_mov_b64_dpp row_newbcast +
//     FMA (rollouts() below; until round 4 lane t = trial t, four active lanes);
//   * the adjoint recursion is three suffix sums over the stages (A' = I + strictly triangular): wave scans;
//   * only the true recursions (the rollout with the linearised step, the Riccati sweep) are serial over the stages.
// Why: measured on MI355X, a one-lane-per-instance kernel (the first design, since removed) is bound by the serial
// FP64 instruction stream of its slowest instance (~47 k instructions per iteration), with 16 of 64 lanes and 1 of 4
// SIMDs per CU usable because the per-instance state has to sit in LDS.
// Resources (round 5, profiles/r05_resource_usage.txt): two builds of this source - 229 VGPRs for a batch up to four waves
// per SIMD deep, 166 beyond (there no constant of the solve lives in a register across the iteration loop: LDS table,
// CTX::fresh) - neither with scratch; 13.5 KB of LDS at N = 20 with 8 vehicles (12 instances per CU), the same layout in both;
// the trial trajectories of the line search live in slots that are dead while it runs (trial_x / trial_u below).
//
// The code is written as alternating "uniform" sections (identical in every lane) and `ctx.phase(f)` sections
// (f(lane) per lane; other lanes may read afterwards what a lane wrote to LDS); tests/cpu_wave_harness.cpp runs the
// phases as loops over the 64 lanes and models the matrix core, the lane permutations and the reductions, so the same
// source is validated on the CPU against the oracle.  The device side of CTX is mpc_wave_dev.hpp.
#pragma once

#include "mpc_core.hpp"

namespace mpc {
namespace wave {

constexpr int kLanes = 64;
constexpr int kTrials = 4;   // step lengths tried by the line search: a_pr * 4^-t (six gained nothing: 4083 against 4084
                             // of 4096 config-3 instances converged in the oracle, and cost 2 KB of LDS per instance)
// A rollout that would take theta or v of the next node out of its bounds gets the one control that decides it (delta
// resp. a) pulled back so that the node keeps this fraction of its slack
constexpr double kProjKeep = 0.1;   // (0.2 until round 3; 0.1 with kKappaEps 30 measured 5 % fewer iterations)
// ... and kProjKeepEnd of it once the barrier parameter is at most kProjEndMu (round 5): in the end-game a slack has to
// shrink by orders of magnitude to meet its multiplier; a projection that keeps 10 % per iteration turns the full Newton step
// into something the Armijo test rejects (the barrier objective rises by 40 x the predicted decrease), and the solve closes
// with 15 - 20 steps of alpha = 1/4, the error falling by exactly 0.75 per iteration (BASELINE config 2's slowest instances,
// egos on the exit straight with the heading next to its bound: 46 -> 34 iterations)
constexpr double kProjKeepEnd = 1e-3, kProjEndMu = 1e-6;
// Round 5, the iteration tail (the CPU restatement under oracle/ carries the same statements):
// (vii) a streak of iterations that needed the whole-sweep Gauss-Newton fallback is watched - every kGnWatch iterations of it
// the scaled KKT error must have halved; if not, the next kGnSkip iterations of the streak go from the failed exact sweep
// straight to IPOPT's inertia correction of the EXACT Hessian (delta_w ladder with memory).  The Gauss-Newton model is convex:
// near a saddle point of the barrier problem its steps are attracted to the saddle, each accepted at full length with a
// predicted decrease of 1e-6, and the iterate drifts off along the unstable direction by 8 % per iteration (half of round 4's
// instances at the iteration cap); the regularised exact Hessian takes a long step along the direction of negative curvature.
constexpr int kGnWatch = 4, kGnSkip = 4;
// (xi) a negative cost weight (the v1 action domain: weights anywhere in [-1, 1]^3, agents/ppo_mpc.py:407-420) makes the stage
// cost itself non-convex, so the Gauss-Newton model is not convex either - it only leaves out the constraint curvature.  Far
// from the solution that is still the better model (fewer iterations); in the end game it costs the quadratic convergence and
// the iteration dithers between the two models at 1e-6.  Such an instance goes from the failed exact sweep straight to the
// inertia ladder once mu <= kGnEndMu (tools/v1_study.py: 98.9 -> 99.9 % of the synthetic v1 draw, 151 -> 160 of the 160 fixture
// states; instances without a negative weight are not touched).
constexpr double kGnEndMu = 1e-2;
// ... and its control blocks are indefinite in nearly every iteration: an iteration that follows one which needed the inertia
// ladder starts at the ladder's first rung directly (a third of the value that worked last), in the same model; every
// kDwProbe-th iteration tries delta_w = 0 first as before (sweeps per iteration on the c4v1 fixture states 1.99 -> 1.50).
constexpr int kDwProbe = 4;
// an instance still iterating after this many iterations is one of the stragglers its batch waits for: its wave asks for the
// highest issue priority on its SIMD (s_setprio).  Same box, alternating: 0.5 - 1 % on a batch of 4096 (seed 4: 4.36 -> 4.33 ms),
// inside the run-to-run spread of seed 0; a threshold of 0 or 16 iterations measures the same
constexpr int kBoostIter = 28;
// (viii) a linearised Newton step smaller than this in the states and the previous control of a stage is applied OPEN LOOP
// there (alpha times the linearised control step) instead of through the feedback law: the feedback acts on the difference of
// two rolled-out trajectories, which carries the rounding of positions ~50 m (7e-15); times gains of 10 - 100 where a control and
// the state bound it decides are both active that is 5e-13 of noise in a control whose slack is 1e-9, and the dual infeasibility
// dithers at 1e-6 - 1e-5 for the rest of the iteration budget (round 4's cap-runners with a standing ego).  The linearised step
// is a recursion on small numbers and keeps its relative precision; the two differ by O(step^2).
constexpr double kOpenLoopStep = 1e-9;
// (ix) IPOPT's acceptable-level termination with IPOPT's defaults (acceptable_tol 1e-6, acceptable_iter 15): live in the
// reference, which sets only max_iter, tol and print options (agents/pure_mpc.py:291-296).  Can only end a solve whose tol is
// tighter than acceptable_tol - never one at the reference's tol 1e-6.  Status 6 (7 with a vehicle held at d = 1).
constexpr double kAcceptableTol = 1e-6;
constexpr int kAcceptableIter = 15;
// (x) the Levenberg-Marquardt term kept across iterations stops here (1e6 until round 4): an instance that needs more than
// the curvature scale of the scaled problem to get a step accepted crawls (alpha = 1/4 at 16 for hundreds of iterations), the
// line-search-failure exit (status 4) ends what cannot move
constexpr double kRegMax = 1.0;
// the barrier parameter is lowered as soon as the error of the barrier problem is below kKappaEps * mu (IPOPT's
// kappa_epsilon, 10 there and here until round 3)
constexpr double kKappaEps = 30.0;
// No trial may bring a control or a bounded state nearer to its bound than this.  Late in a solve tau = 1 - mu lets a slack
// shrink by the factor mu ~ 1e-9 per iteration; two such steps take a control at -5 below one ulp of its bound: slack
// exactly 0, 1 / slack infinite, a NaN in the sweep that no regularisation repairs (status 2, seen on the GPU about once in
// 4000 instances).  IPOPT corrects slacks that become too small likewise (Waechter & Biegler 2006, section 3.5).
constexpr double kMinSlack = 1e-14;
// how far inside its heading bound the cold start puts a node that zero controls would leave on or outside it (rad)
constexpr double kInitPush = 1e-2;

// per-stage slots in LDS (doubles); trajectory buffer b lives at b*6
enum : int {
    W_X = 0,     // 4 (buffer 0; buffer 1 at +6)
    W_U = 4,     // 2
    // stage linearisation of the current trajectory as the 3 x 4 table F[c][i] at W_LIN + 4 c + i: row i = state component
    // (x, y, theta, v), column c = what it is multiplied with (d theta, d v, d delta; the d a column is the constant (0, 0, 0, dt)
    // and lives in a register).  In this form lane 8 + i of a rollout row reads row i with three loads at the SAME offsets from
    // its own base (round 5: row-cooperative rollout, see rollouts()); the structural zeros are written with the rest by the
    // preparation phase.
    W_LIN = 12,  // 12
    W_ZXL = 24,  // 2
    W_ZXU = 26,  // 2
    W_ZUL = 28,  // 2
    W_ZUU = 30,  // 2
    W_Y = 32,    // 4  node gradient g_k, then adjoint y_k; after the factorisation: parked Newton step (du0, du1, dtheta, dv)
    // gains, TRANSPOSED since round 5 so that the two control lanes of a rollout row read their rows at the same offsets:
    // Kx[i][j] at W_KX + 2 j + i, Kp[i][j] at W_KP + 2 j + i, kf[i] at W_KF + i - contiguous, 14 words
    W_KX = 36,   // 8
    W_KP = 44,   // 4
    W_KF = 48,   // 2
    W_RV = 50,   // 1
    // the constants 1 and dt of the stage matrix F = [A B; 0 I], once per stage (written by init_tables, never again): every
    // operand of a Riccati stage is then a stage-relative word and its address one addition - fetched from the table of
    // constants behind the stages they cost a select per operand and seven lane masks in scalar registers
    W_ONE = 51,  // 1
    W_DTC = 52,  // 1
    // The stage stride is kept ODD: lane k of a stage-parallel phase addresses word k * stride + slot, and with 64 LDS
    // banks of 4 bytes an even number of doubles per stage puts every 4th (56 slots) or 16th (46, 54 slots) stage on the
    // same banks - measured with 56 slots: SQ_LDS_BANK_CONFLICT 17 % of the LDS cycles, against 3 % in round 1.
    W_SLOTS = 53,
    W_LX = 53,   // 2  (collision-cost variant) potential gradient; after the factorisation: parked (dx, dy) of the node
    W_Q = 55,    // 3  exact 2x2 curvature of the potential; after the factorisation: parked wall slack, wall dual step
    W_QG = 58,   // 3  its Gauss-Newton part
    W_ZW = 61,   // 1  multiplier of the node's wall constraint |p - o_j|^2 - 1 >= 0
    W_WJ = 62,   // 1  its vehicle j (as a double), -1: none
    W_CROSS = 63,  // 1  after the line search: vehicle a rejected trial took across d = 1
    W_SLOTS_CC = 65   // (one spare word keeps the stride odd)
};
// positions in the W_LIN table of the eight values that are not structural zeros
constexpr int kZeroWord = W_LIN + 2;      // a structural zero of the table: the stage-relative address of the constant 0
enum : int { LIN_A02 = 0, LIN_A12 = 1, LIN_A03 = 4, LIN_A13 = 5, LIN_A23 = 6, LIN_B01 = 8, LIN_B11 = 9, LIN_B21 = 10 };
// parked values (valid between the factorisation and the next preparation phase)
enum : int { W_DXY = W_LX, W_GW = W_Q, W_DZW = W_Q + 1 };
// The four trial trajectories of the line search (x 4, u 2 per node, stage stride like everything else) live in slots
// that are dead while the line search runs, so that an instance does not pay 24 more words per stage for them:
//   trial 0  the spare trajectory buffer (the accepted trial ends up there)
//   trial 1  W_LIN + 0..5          the linearisation is recomputed by the next preparation phase
//   trial 2  W_LIN + 6..11         likewise
//   trial 3  W_KX + 0..5           the gains of stage k have been read by every lane (one wave, program order) when
//                                  the stage's results are stored; nothing after the rollout reads gains
// trial_x(t) is the slot of element 0, trial_u(t) the slot of element 4 minus 4: element e of node k sits at
// k * stride + (e < 4 ? trial_x : trial_u) + e.
static_assert(kTrials == 4, "the trial areas below are laid out for four trials");
MPC_HD constexpr int trial_x(bool, int t, int TB) { return t == 0 ? TB : (t == 1 ? W_LIN : (t == 2 ? W_LIN + 6 : W_KX)); }
MPC_HD constexpr int trial_u(bool cc, int t, int TB) { return trial_x(cc, t, TB); }
// scratch behind the stage arrays: the three constants the F operands are made of besides the linearisation values
enum : int {
    SC_SPARE = 0,  // 0.0, 1.0, dt
    SC_TRIG = 4,   // the sine / cosine kernel coefficients (mpc_core.hpp TrigCoef), loaded per phase
    SC_LOG = SC_TRIG + kTrigWords,   // coefficients of log_pos
    SC_BND = SC_LOG + kLogWords,     // the bounds of the NLP: xlo(0), xhi(0), xlo(1), xhi(1), ulo(0), uhi(0), ulo(1), uhi(1)
    SC_K = SC_BND + 8,               // constants of this solve: objective scale and what is derived from it (K_* below)
    SC_NB = SC_K + 6,                // -kNoBound, (unused), +kNoBound: the "projection box" of a lane that has none
    SC_SIZE = SC_NB + 3
};
enum : int { K_SF = 0, K_RD, K_RC, K_QTT, K_Q33, K_MUMIN };
constexpr int kMaxHorizon = kLanes;   // lane k = stage k in the stage-parallel phases
// compact stage cost Hessian / gradient, assembled for all stages before a sweep into slots that are free during it:
// the stage's gain slots (overwritten by the gains once the stage is done) and the trial trajectory buffer
enum : int {
    A_L00 = W_KX + 0, A_L01 = W_KX + 1, A_L11 = W_KX + 2, A_H22 = W_KX + 3, A_H23 = W_KX + 4, A_H33 = W_KX + 5,
    A_WTD = W_KX + 6, A_WVD = W_KX + 7, A_H66 = W_KP + 0, A_H77 = W_KP + 1, A_HV6 = W_KP + 2, A_HV7 = W_KF + 0,
    A_HV4 = W_KF + 1,
    A_HV0 = W_X,   // + trial buffer offset: hv0..3
    A_HV5 = W_U    // + trial buffer offset
};

// What a trial rollout applies at stage k that does not depend on the trial - the fraction-to-the-boundary box of the two
// controls, the projection box of theta / v of the next node, the feasibility margins and the previous control of the current
// iterate - is computed once per line search, stage-parallel, into kPreSlots words per stage behind the stage's own slots
// (round 3: 12 words, only in the builds with LDS to spare; round 5: every build, laid out for the row-cooperative rollout).
// PQ[m][j] at 4 m + j, m = 0, 1: lane 2 + j of a rollout row (j = 0: theta, 1: v, 2: a, 3: delta) - theta, v: feasibility
//     margin at the lower / upper bound; a, delta: lower / upper edge of the clamp box
// PB[m][j'] at 8 + 2 m + j', m = 0, 1: lower / upper edge of a projection box - j' = 0 (lane 2): theta of the next node; j' = 1
//     (lane 4): the box that keeps v of the next node inside ITS projection box.  v and delta have no such box: their lanes read
//     -kNoBound / +kNoBound from the table of constants (SC_NB)
constexpr int kPreSlots = 12;
enum : int { PQ = 0, PB = 8 };
constexpr double kNoBound = 1e300;

MPC_HD constexpr int stage_slots(bool cc) { return (cc ? W_SLOTS_CC : W_SLOTS) + kPreSlots; }
// doubles of LDS one instance needs: stage arrays + constants + other vehicles
MPC_HD constexpr int lds_doubles(bool cc, int N, int V) { return stage_slots(cc) * (N + 1) + SC_SIZE + (cc ? 4 * V : 0); }

// section ids for CTX::tick (cycle attribution in tools/ubench/wave_sections.hip; a no-op in the product kernel)
enum : int {
    T_PREP = 0, T_ADJOINT, T_DUALRES, T_RIC_INIT, T_RIC_SCALARS, T_RIC_L1, T_RIC_L2, T_RIC_2X2, T_RIC_L4, T_LINEAR,
    T_RATIOS, T_ROLL_DYN, T_ROLL_COST, T_DUALUPD,
    // finer attribution inside one rollout stage (only a profiling context with kFine = true ticks these)
    T_R_FEEDBACK, T_R_CLAMP, T_R_DYN, T_R_STORE, T_R_CHECK, T_R_PROJ, T_R_LDSW, T_COUNT
};
template <class CTX, class = void>
struct fine_ticks { static constexpr bool value = false; };
template <class CTX>
struct fine_ticks<CTX, decltype((void)CTX::kFine)> { static constexpr bool value = CTX::kFine; };

// A value that differs per lane and lives across phases: one register per lane on the device; the host emulation,
// which runs the lanes of a phase one after the other, keeps all 64.
template <class T>
struct PerLane {
#if defined(__HIPCC__)   // both passes of a hipcc compile (the host pass never runs this code)
    T v;
    MPC_HD T &at(int) { return v; }
#else
    T v[kLanes];
    MPC_HD T &at(int lane) { return v[lane]; }
#endif
};

// (n2/d2 > n1/d1) for positive denominators, division-free.  Equal ratios prefer the larger denominator, so that the
// two partners of a symmetric exchange always agree on the winner.
MPC_HD bool ratio_greater(double n2, double d2, double n1, double d1) {
    const double a = n2 * d1, b = n1 * d2;
    return a > b || (a == b && d2 > d1);
}

// Host-side model of the wave reductions (mpc_wave_dev.hpp): symmetric partner exchanges inside each 16-lane row
// (xor 1, xor 2, mirror within 8, mirror within 16), then the four row results as (r0 op r1) op (r2 op r3).
#if !defined(__HIPCC__)
inline int row_partner(int l, int step) {
    return step == 0 ? (l ^ 1) : (step == 1 ? (l ^ 2) : (step == 2 ? ((l & ~7) | (7 - (l & 7))) : ((l & ~15) | (15 - (l & 15)))));
}
template <class T, class OP>
inline void host_row_reduce(PerLane<T> &p, OP op) {
    for (int step = 0; step < 4; ++step) {
        T nv[kLanes];
        for (int l = 0; l < kLanes; ++l) nv[l] = op(p.v[l], p.v[row_partner(l, step)]);
        for (int l = 0; l < kLanes; ++l) p.v[l] = nv[l];
    }
}
template <class T, class OP>
inline T host_reduce(PerLane<T> &p, OP op) {
    host_row_reduce(p, op);
    return op(op(p.v[0], p.v[16]), op(p.v[32], p.v[48]));
}
#endif

// F = [A B; 0 I; 0 0] (8 x 8 with two zero rows; state 4, previous control 2, control 2) of the bicycle model - exact or
// linearised about a nominal trajectory - is made of 0, 1, dt and eight stored values a02 a03 a12 a13 a23 b01 b11 b21
// (slots lin .. lin + 7 of the stage): LDS word of element (r, c), stage-relative (>= 0) or, for the three constants,
// the absolute word encoded as -(word + 1).  Shared by the NLP solver and the LTV-QP solver (mpc_ltv.hpp).
// (written as one chain of selects on the key 8 r + c: the role tables are recomputed every iteration in every lane,
// and a per-lane `if` / `switch` cascade costs an exec-mask save / restore and a branch per case)
MPC_HD constexpr int stage_transition_word(int r, int c, int lin, int zero, int one, int dtw) {
    const int key = 8 * r + c;
    int w = zero;
    w = (r < 4 && r == c) ? one : w;
    w = key == 8 * 4 + 6 ? one : w;
    w = key == 8 * 5 + 7 ? one : w;
    w = key == 8 * 0 + 2 ? lin + 0 : w;
    w = key == 8 * 0 + 3 ? lin + 1 : w;
    w = key == 8 * 1 + 2 ? lin + 2 : w;
    w = key == 8 * 1 + 3 ? lin + 3 : w;
    w = key == 8 * 2 + 3 ? lin + 4 : w;
    w = key == 8 * 3 + 6 ? dtw : w;
    w = key == 8 * 0 + 7 ? lin + 5 : w;
    w = key == 8 * 1 + 7 ? lin + 6 : w;
    w = key == 8 * 2 + 7 ? lin + 7 : w;
    return w;
}

// ---------------------------------------------------------------------------------------------------
// CTX (one per wave / instance) provides
//   double ld(int i), void st(int i, double v)   LDS words of this instance
//   void phase(F f)                              f(lane) for the 64 lanes; LDS written in it is visible to all afterwards
//   double ref(int k, int c)                     reference path column c at stage k
//   static constexpr int kN                      compile-time horizon or 0
//   void tick(int section)                       attributes the time since the previous tick to `section`
//   double wave_sum / wave_max / wave_min(PerLane<double>&)   reduction over the 64 lanes, result in every lane
//   void wave_max_ratio(PerLane<double>& n, PerLane<double>& d, double &rn, double &rd)   pair with the largest n/d
//   void wave_sum2(PerLane<double>&, double &lo, double &hi)   sums over lanes 0..31 and 32..63
//   int wave_bcast(PerLane<int>&, int lane)                    value of one lane
//   void lanes(F f)                              f(lane) for the 64 lanes, no barrier (register-only work)
//   void mfma(PerLane<double>& a, PerLane<double>& b, PerLane<double>& cd)   cd += a x b, v_mfma_f64_4x4x4f64:
//        lane l = 16 hi + 4 blk + lo holds A_blk[row lo][k hi], B_blk[k hi][col lo], C/D_blk[row hi][col lo]
//   double lane_get(PerLane<double>&, int lane)  value of one lane, in every lane
//   void wave_suffix_sum(PerLane<double>&)       in place: lane i <- sum of lanes i..63
//   void sched_fence()                           the compiler schedules no instruction across this point
//   double keep(double v)                        v, computed by this point (an opaque use: the value cannot sink below)
//   double fresh(double v)                       a wave-uniform v, but the compiler cannot see where it comes from (keeps
//                                                loop-invariant expressions of solve constants out of vector registers)
//   int opaque(int v)                            v, but the compiler cannot see that (keeps recomputable per-lane tables
//                                                from being hoisted out of the iteration loop and held in registers)
// The caller has already stored W_RV (all stages) and the other vehicles (x, y, dx, dy per vehicle).
// ---------------------------------------------------------------------------------------------------
template <bool CC, class CTX>
struct Solver {
    const SolveParams &P;
    CTX &c;
    const int N, SL, SCR, OTH;
    const double dt;
    double x0[4];
    double ws_, wc_, wd_, wcoll;   // read through WS() ... WCOLL(), SF()
    double sf = 1.0;
    MPC_HD double SF() const { return sc(SC_K + K_SF); }
    MPC_HD double RD() const { return sc(SC_K + K_RD); }
    MPC_HD double RC() const { return sc(SC_K + K_RC); }
    MPC_HD double WS() const { return c.fresh(ws_); }
    MPC_HD double WC() const { return c.fresh(wc_); }
    MPC_HD double WD() const { return c.fresh(wd_); }
    MPC_HD double WCOLL() const { return c.fresh(wcoll); }

    MPC_HD Solver(const SolveParams &P_, CTX &c_, const double *x0_, double ws, double wc, double wd, double wcl)
        : P(P_), c(c_), N(CTX::kN > 0 ? CTX::kN : P_.N), SL(stage_slots(CC)), SCR(SL * (N + 1)), OTH(SCR + SC_SIZE),
          dt(P_.dt), ws_(c_.uni(ws)), wc_(c_.uni(wc)), wd_(c_.uni(wd)), wcoll(c_.uni(wcl)) {
        x0[0] = x0_[0]; x0[1] = x0_[1]; x0[2] = x0_[2]; x0[3] = x0_[3];
    }
    MPC_HD double S(int k, int slot) const { return c.ld(k * SL + slot); }
    MPC_HD void S(int k, int slot, double v) { c.st(k * SL + slot, v); }
    MPC_HD double sc(int i) const { return c.ld(SCR + i); }
    MPC_HD void sc(int i, double v) { c.st(SCR + i, v); }
    MPC_HD double oth(int j, int q) const { return c.ld(OTH + j * 4 + q); }
    // Bounds and function coefficients come from a table in LDS, not from literals: a 64-bit literal that is used in
    // more than one place gets hoisted out of the iteration loop into a register pair for the whole solve, and two dozen
    // of those are the difference between 3 and 4 resident waves per SIMD.  A table entry is loaded where it is used
    // (one broadcast ds_read), or once in front of a serial loop.
    MPC_HD double xlo(int i) const { return sc(SC_BND + 0 + 2 * i); }
    MPC_HD double xhi(int i) const { return sc(SC_BND + 1 + 2 * i); }
    MPC_HD double ulo(int i) const { return sc(SC_BND + 4 + 2 * i); }
    MPC_HD double uhi(int i) const { return sc(SC_BND + 5 + 2 * i); }
    MPC_HD double flog(double x) const {
        double K[kLogWords];
        for (int i = 0; i < kLogWords; ++i) K[i] = sc(SC_LOG + i);
        return log_pos(K, x);
    }
    MPC_HD TrigCoef trig() const {
        TrigCoef K;
        for (int i = 0; i < 6; ++i) {
            K.s[i] = sc(SC_TRIG + i);
            K.c[i] = sc(SC_TRIG + 6 + i);
        }
        return K;
    }

    PerLane<double> red_a, red_b, red_c, red_w;   // per-lane operands of the wave reductions
    PerLane<int> ls_feas;                  // line search: the lanes of row t = trial t stayed inside the fraction-to-the-boundary box
    // d = 1 discontinuity of the collision cost (archive/pure_mpc.py:189-196: 100/d^2 outside, 1000/d^2 inside).  A
    // vehicle that a rejected trial took across d = 1 inwards is kept outside from then on by the constraint
    // |p_k - o_jk|^2 - 1 >= 0 of that node (one per node, multiplier W_ZW): the cost jumps upwards there, so a minimiser
    // pressed against d = 1 is a constrained stationary point of the outer branch (status 5), which no smooth method
    // reaches otherwise (round 1: 3.7 % of the config-3 instances ended "stalled").  any_wall: some node has one.
    int any_wall = 0;
    MPC_HD int f_word(int r, int c) const {
        // stage_transition_word() names the eight stored values lin + 0..7 = a02 a03 a12 a13 a23 b01 b11 b21; here they sit in
        // the F[c][i] table of W_LIN
        // ... and the structural constants in words of the stage too: 0 is one of the zeros of that table, 1 and dt W_ONE, W_DTC
        const int w = stage_transition_word(r, c, 0, 1000, 1001, 1002);
        int t = w;
        t = w == 1000 ? kZeroWord : t;
        t = w == 1001 ? W_ONE : t;
        t = w == 1002 ? W_DTC : t;
        t = w == 0 ? W_LIN + LIN_A02 : t;
        t = w == 1 ? W_LIN + LIN_A03 : t;
        t = w == 2 ? W_LIN + LIN_A12 : t;
        t = w == 3 ? W_LIN + LIN_A13 : t;
        t = w == 4 ? W_LIN + LIN_A23 : t;
        t = w == 5 ? W_LIN + LIN_B01 : t;
        t = w == 6 ? W_LIN + LIN_B11 : t;
        t = w == 7 ? W_LIN + LIN_B21 : t;
        return t;
    }
    // ================================================================================================================
    // Riccati / DDP sweep on 4x4 blocks (round 4).  The stage matrix F = [A B; 0 I] has structure the 8x8 form above does
    // not use: the next "previous control" IS the control, so with the value function of node k + 1 split as
    //     V(x, p) = 1/2 x'Pxx x + x'Pxp p + 1/2 p'Ppp p + px'x + pp'p        (x: state 4, p: previous control 2)
    // the stage's quadratic model is
    //     Qxx = Lxx + A'Pxx A            Qux = Lux + (Pxx B + Pxp)'A          Quu = Luu + B'(Pxx B + Pxp) + Pxp'B + Ppp
    //     qx  = lx + A'px                qu  = lu + B'px + pp                 Qpp = rd I, Qpu = -rd I, Qxp = 0, qp = lp
    // and the new value function, with W = Quu^-1, kf = -W qu, Kx = -W Qux, Kp = rd W:
    //     Pxx' = Qxx + Qux'Kx            Pxp' = rd Qux'W           Ppp' = rd I - rd^2 W         px' = qx + Qux'kf
    //     pp'  = qp - rd kf.
    // Everything 4-dimensional is a 4x4 matrix-core block (B, Pxp, Qux padded with zeros), everything 2-dimensional is
    // wave-uniform scalar algebra.  ALL FOUR blocks of v_mfma_f64_4x4x4f64 compute the same product: a value held by lane
    // 16 hi + 4 blk + lo depends on (hi, lo) only.  The redundancy is free - the instruction processes four blocks whatever
    // they hold - and it is what removes every block move of the 8x8 form: with r[hi][lo] a register's content,
    //     mfma(a, b, cd):  cd += a'b    (A operand lane (hi, lo) = A[row lo][k hi]; B and C/D: [hi][lo])
    // so a product's output is directly the B operand of the next product, directly the A operand of a product with its
    // TRANSPOSE, and a symmetric matrix is its own A operand.  Twelve products per stage, operands as they fall:
    //     T0 = Pxx A             T1 = Pxx B + Pxp        Qxx = Lxx + A'T0        Qux = Lux + T1'A
    //     M  = Luu + Pxp'B + B'T1  (Quu = M + Ppp)       qx = lx + A'px          qu = lu + B'px (+ pp)
    //     Ya = -adj(Quu) Qux (= det Kx)      and with Qs = Qux / det:
    //     Pxx' = Qxx + Qs'Ya     Pxp' = Qs'(rd adj(Quu))      px' = qx + Qs'(-adj(Quu) qu)
    // The 8x8 form needed nine products and eight block moves (22 DPP moves + 20 copies + the masks: 52 of the stage's 165
    // instructions; 145 now).  Measured (profiles/r04_sweep4.txt): a wave alone on its SIMD 33.2 -> 32.3 us per iteration,
    // a straggler 39.1 -> 37.6, config 2 0.99 -> 0.95 ms, one launch of 65 536 2.70 -> 2.80 M solves/s, 231 -> 218 registers
    // in the latency build - which with it is also the faster one for a batch of 4096 (config 3 5.19 -> 5.03 ms).
    // ================================================================================================================
    // LDS word of this lane's element of each operand: stage-relative (>= 0) or absolute, encoded as -(word + 1) - the
    // constants 0, 1, dt of the table; an element that is structurally zero reads the 0 there, so no load needs a select
    PerLane<int> r_a, r_b, r_lxx, r_lux, r_m, r_lx, r_lu, r_kx;
    MPC_HD void set_roles4(int AB) {
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
            const int zero = kZeroWord;
            r_a.at(lane_) = f_word(hi, lo);                        // A[hi][lo]
            r_b.at(lane_) = f_word(hi, lo < 2 ? 6 + lo : 4);       // B~[hi][lo]: columns a, delta of F, then zeros
            const int a = hi < lo ? hi : lo, b = hi < lo ? lo : hi;
            const int key = a * 4 + b;
            int slot = zero;
            slot = key == 0 * 4 + 0 ? A_L00 : slot;
            slot = key == 0 * 4 + 1 ? A_L01 : slot;
            slot = key == 1 * 4 + 1 ? A_L11 : slot;
            slot = key == 2 * 4 + 2 ? A_H22 : slot;
            slot = key == 2 * 4 + 3 ? A_H23 : slot;
            slot = key == 3 * 4 + 3 ? A_H33 : slot;
            r_lxx.at(lane_) = slot;
            r_lux.at(lane_) = hi == 1 ? (lo == 2 ? A_WTD : (lo == 3 ? A_WVD : zero)) : zero;   // Lux[delta][theta], [delta][v]
            r_m.at(lane_) = (hi == 0 && lo == 0) ? A_H66 : ((hi == 1 && lo == 1) ? A_H77 : zero);
            r_lx.at(lane_) = lo == 0 ? AB + A_HV0 + hi : zero;                                 // in the trial buffer
            r_lu.at(lane_) = lo == 0 ? (hi == 0 ? A_HV6 : (hi == 1 ? A_HV7 : zero)) : zero;
            // where this lane stores its gain: the four blocks of the matrix-core layout hold the same values, so block 0 stores
            // Kx[hi][lo] (transposed layout), block 1 Kp[hi][lo], block 2 kf[hi] - one ds_write for all fourteen words
            int out = -1;
            out = (blk == 0 && hi < 2) ? W_KX + 2 * lo + hi : out;
            out = (blk == 1 && hi < 2 && lo < 2) ? W_KP + 2 * lo + hi : out;
            out = (blk == 2 && hi < 2 && lo == 0) ? W_KF + hi : out;
            r_kx.at(lane_) = out;
        });
    }

    MPC_HD void load_stage4(int k, int AB, PerLane<double> &RA, PerLane<double> &RB, PerLane<double> &QXX, PerLane<double> &QUX,
                            PerLane<double> &M, PerLane<double> &QX, PerLane<double> &QU, double &lp0, double &lp1) {
        c.lanes([&](int lane) {
            const int base = k * SL;
            auto word = [&](int w) { return c.ld(base + w); };      // every operand is a word of the stage (W_ONE, W_DTC, kZeroWord)
            RA.at(lane) = word(r_a.at(lane));
            RB.at(lane) = word(r_b.at(lane));
            QXX.at(lane) = word(r_lxx.at(lane));
            QUX.at(lane) = word(r_lux.at(lane));
            M.at(lane) = word(r_m.at(lane));
            QX.at(lane) = word(r_lx.at(lane));
            QU.at(lane) = word(r_lu.at(lane));
        });
        lp0 = S(k, A_HV4);
        lp1 = S(k, AB + A_HV5);
    }
    // one backward sweep over the stages with the stage Hessians / gradients the assembly phase left in LDS; false: a stage's
    // control block was not positive definite (also not with its Gauss-Newton terms alone)
    MPC_HD bool sweep4(int CB, int AB, double mu, double delta_w, bool gn, double &dV1) {
        // ---- terminal value function: barrier terms of (theta, v)_N
        double tsig[2], tgr[2];
        for (int i = 0; i < 2; ++i) {
            const double xi = S(N, CB + W_X + 2 + i);
            const double rl = frcp(xi - xlo(i)), ru = frcp(xhi(i) - xi);
            tsig[i] = S(N, W_ZXL + i) * rl + S(N, W_ZXU + i) * ru + delta_w;
            tgr[i] = mu * (ru - rl);
        }
        PerLane<double> PXX, PXP, PX;
        c.lanes([&](int lane_) {
            const int lane = c.opaque_shared(lane_);
            const int hi = lane >> 4, lo = lane & 3;
            PXX.at(lane_) = hi == lo ? (hi == 2 ? tsig[0] : (hi == 3 ? tsig[1] : delta_w)) : 0.0;
            PXP.at(lane_) = 0.0;
            PX.at(lane_) = lo == 0 ? (hi == 2 ? tgr[0] : (hi == 3 ? tgr[1] : 0.0)) : 0.0;
        });
        double pp00 = 0.0, pp01 = 0.0, pp11 = 0.0, ppv0 = 0.0, ppv1 = 0.0;    // Ppp, pp of the node behind the stage
        c.tick(T_RIC_INIT);
        const double rd_full = RD();
#pragma unroll 1      // (two stages per trip: no gain for a lone wave, 2 % slower batches - profiles/r04_sweep4.txt)
        for (int k = N - 1; k >= 0; --k) {
            const double rdk = (k >= 1) ? rd_full : 0.0;
            // ---- operands from LDS (they do not depend on the recursion; requesting them one stage ahead - behind this stage's
            //      products, pinned there by scheduling fences, verified in the ISA - is 1.5 % SLOWER: profiles/r04_sweep4.txt,
            //      like the four attempts on the 8x8 form before)
            PerLane<double> RA, RB, QXX, QUX, M, QX, QU, T0, T1;
            double lp0, lp1;
            load_stage4(k, AB, RA, RB, QXX, QUX, M, QX, QU, lp0, lp1);
            c.lanes([&](int lane) {
                T0.at(lane) = 0.0;
                T1.at(lane) = PXP.at(lane);
            });
            c.tick(T_RIC_SCALARS);
            // ---- the value function of node k + 1 through the stage
            c.mfma(PXX, RA, T0);       // T0 = Pxx A
            c.mfma(PXX, RB, T1);       // T1 = Pxx B + Pxp
            c.mfma(PXP, RB, M);        // M  = Luu + Pxp'B ...
            c.mfma(RA, PX, QX);        // qx = lx + A'px
            c.mfma(RB, PX, QU);        // qu = lu + B'px        (+ pp below)
            c.tick(T_RIC_L1);
            c.mfma(RB, T1, M);         // ... + B'T1 = Quu - Ppp
            c.mfma(T1, RA, QUX);       // Qux = Lux + T1'A
            c.mfma(RA, T0, QXX);       // Qxx = Lxx + A'T0
            c.tick(T_RIC_L2);
            // ---- 2x2 control block (uniform): elements (0,0) (0,1) (1,1) of M sit in lanes 0 1 17 (M is symmetric up to the
            //      rounding of the two products above), qu in lanes 0, 16; the value function's own 2x2 part joins here
            double ha = c.lane_get(M, 0) + pp00, hb = c.lane_get(M, 1) + pp01, hc = c.lane_get(M, 17) + pp11;
            const double hu0 = c.lane_get(QU, 0) + ppv0, hu1 = c.lane_get(QU, 16) + ppv1;
            double det = ha * hc - hb * hb;
            bool pd = (ha > 0.0) & (hc > 0.0) & (det > c.fresh(1e-12) * ha * hc);
            // The rest of the stage is computed BEFORE the test is acted on (profiles/r04_spec_sweep.txt); Qxx, Qux, qx stay
            // intact for the fallback.  The products are arranged so that only ONE multiplication sits between the reciprocal
            // of the determinant and the value function of this node: Ya = -adj(Quu) Qux, rd adj(Quu) and -adj(Quu) qu need no division
            // and run on the matrix cores while the reciprocal is refined; it then scales the other operand (Qs = Qux / det).
            PerLane<double> NWA, RWA, Ya, QS, KFB, PXXn, PXPn, PXn;
            double idet, i00, i01, i11, kf0, kf1;
            auto stage_tail = [&]() {
                const double kfa0 = hb * hu1 - hc * hu0, kfa1 = hb * hu0 - ha * hu1;      // -adj(Quu) qu = det kf
                c.lanes([&](int lane_) {
                    const int lane = c.opaque_shared(lane_);
                    const int hi = lane >> 4, lo = lane & 3;
                    const double w00 = (hi == 0 && lo == 0) ? 1.0 : 0.0, w11 = (hi == 1 && lo == 1) ? 1.0 : 0.0;
                    const double wof = ((hi == 0 && lo == 1) || (hi == 1 && lo == 0)) ? 1.0 : 0.0;
                    const double c1 = (hi == 1 && lo == 0) ? 1.0 : 0.0;
                    const double nwa = wof * hb - w00 * hc - w11 * ha;      // -adj(Quu), padded
                    NWA.at(lane_) = nwa;
                    RWA.at(lane_) = -rdk * nwa;                             // rd adj(Quu)
                    KFB.at(lane_) = w00 * kfa0 + c1 * kfa1;                 // det kf in column 0
                    Ya.at(lane_) = 0.0;
                    PXPn.at(lane_) = 0.0;
                    PXXn.at(lane_) = QXX.at(lane_);
                    PXn.at(lane_) = QX.at(lane_);
                });
                c.mfma(NWA, QUX, Ya);      // Ya = -adj(Quu) Qux   (rows 0, 1;  Kx = Ya / det)
                idet = frcp(det);
                c.lanes([&](int lane) { QS.at(lane) = idet * QUX.at(lane); });
                c.mfma(QS, Ya, PXXn);      // Pxx' = Qxx + Qux'Kx
                c.mfma(QS, RWA, PXPn);     // Pxp' = Qux'(rd W)
                c.mfma(QS, KFB, PXn);      // px' = qx + Qux'kf
                i00 = hc * idet;
                i01 = -hb * idet;
                i11 = ha * idet;
                kf0 = kfa0 * idet;
                kf1 = kfa1 * idet;
            };
            stage_tail();
            if (!pd && !gn) {
                // not positive definite with the exact Hessian: this stage alone falls back to its Gauss-Newton terms
                // (constraint curvature off, radial part of the collision potential) - they are taken out of Qxx, Qux, Quu,
                // the products with the value function stay
                double wdd, wtt = 0.0, wtv = 0.0, wtd = 0.0, wvd = 0.0, q00 = 0.0, q01 = 0.0, q11 = 0.0;
                {
                    const double v = S(k, CB + W_X + 3);
                    double Sn, Cn, sb, cb_, bp, bpp;
                    dyn_eval(trig(), S(k, CB + W_X + 2), S(k, CB + W_U + 1), Sn, Cn, sb, cb_);
                    beta_derivs(sb, cb_, bp, bpp);
                    const double yy0 = S(k + 1, W_Y + 0), yy1 = S(k + 1, W_Y + 1), yy2 = S(k + 1, W_Y + 2);
                    const double g = -(yy0 * Cn + yy1 * Sn), h = -(yy0 * Sn - yy1 * Cn);
                    wdd = dt * v * (g * bp * bp + h * bpp) + dt * yy2 * v * kInvWheelbase * (-sb * bp * bp + cb_ * bpp);
                    if (k >= 1) {
                        wtt = dt * v * g;
                        wtv = dt * h;
                        wtd = dt * v * g * bp;
                        wvd = dt * h * bp + dt * yy2 * cb_ * bp * kInvWheelbase;
                        if (CC) {
                            q00 = S(k, W_QG + 0) - S(k, W_Q + 0);
                            q01 = S(k, W_QG + 1) - S(k, W_Q + 1);
                            q11 = S(k, W_QG + 2) - S(k, W_Q + 2);
                            if (any_wall && S(k, W_WJ) >= 0.0) {
                                q00 += 2.0 * S(k, W_ZW);
                                q11 += 2.0 * S(k, W_ZW);
                            }
                        }
                    }
                }
                c.lanes([&](int lane_) {
                    const int lane = c.opaque_shared(lane_);
                    const int hi = lane >> 4, lo = lane & 3;
                    const int a = hi < lo ? hi : lo, b = hi < lo ? lo : hi;
                    double cx = 0.0;
                    switch (a * 4 + b) {
                        case 0 * 4 + 0: cx = q00; break;
                        case 0 * 4 + 1: cx = q01; break;
                        case 1 * 4 + 1: cx = q11; break;
                        case 2 * 4 + 2: cx = -wtt; break;
                        case 2 * 4 + 3: cx = -wtv; break;
                        default: break;
                    }
                    QXX.at(lane_) += cx;
                    QUX.at(lane_) += hi == 1 ? (lo == 2 ? -wtd : (lo == 3 ? -wvd : 0.0)) : 0.0;
                });
                hc -= wdd;
                det = ha * hc - hb * hb;
                pd = (ha > 0.0) & (hc > 0.0) & (det > c.fresh(1e-12) * ha * hc);
                if (pd) stage_tail();
            }
            if (!pd) return false;
            c.tick(T_RIC_2X2);
            dV1 += 0.5 * (kf0 * hu0 + kf1 * hu1);
            const double kp00 = rdk * i00, kp01 = rdk * i01, kp11 = rdk * i11;
            c.lanes([&](int lane) {
                PXX.at(lane) = PXXn.at(lane);
                PXP.at(lane) = PXPn.at(lane);
                PX.at(lane) = PXn.at(lane);
                // gains: Kx = -W Qux = Ya / det, Kp = rd W = rd adj(Quu) / det, kf = (det kf) / det - each lane its own word, the
                // same products in the same order as the scalars kf0, kf1, kp00 .. of the recursion (Kp of stage 0, which no
                // rollout multiplies with anything but zero, is stored as rd W as well)
                const int ks = r_kx.at(lane);
                const bool is_kp = (ks >= W_KP) & (ks < W_KF), is_kf = ks >= W_KF;
                const double t = is_kf ? KFB.at(lane) : (is_kp ? -NWA.at(lane) : Ya.at(lane));
                const double v = (idet * t) * (is_kp ? rd_full : 1.0);
                if (ks >= 0) S(k, ks, v);
            });
            pp00 = rdk - rdk * kp00;      // Ppp' = rd I - rd^2 W
            pp01 = -rdk * kp01;
            pp11 = rdk - rdk * kp11;
            ppv0 = lp0 - rdk * kf0;       // pp' = lp - rd kf
            ppv1 = lp1 - rdk * kf1;
            c.tick(T_RIC_L4);
        }
        return true;
    }

    // ---- cost pieces (agents/pure_mpc.py:119-214) --------------------------------------------------
    MPC_HD double track(int k, double x_0, double x_1, double x_2, double x_3, double *g) const {
        const double s = c.ref(k, R_SIN), cc = c.ref(k, R_COS);
        const double dx = x_0 - c.ref(k, R_X), dy = x_1 - c.ref(k, R_Y);
        const double perp = dx * s - dy * cc, para = dx * cc + dy * s;
        const double dv = x_3 - S(k, W_RV), dth = x_2 - c.ref(k, R_H);
        if (g) {
            g[0] = 10.0 * (8.0 * perp * s + 4.0 * para * cc);
            g[1] = 10.0 * (-8.0 * perp * cc + 4.0 * para * s);
            g[2] = 10.0 * dth;
            g[3] = 20.0 * WS() * dv;
        }
        return 10.0 * (4.0 * perp * perp + 2.0 * para * para + WS() * dv * dv + 0.5 * dth * dth);
    }
    // (j0, dj: the vehicles j0, j0 + dj, ... - the preparation phase splits them over lane groups, collision_parts below)
    MPC_HD double dist(int k, double x_0, double x_1, double *d8, int j0 = 0, int dj = 1) const {
        double J = 0.0, g0 = 0, g1 = 0, h00 = 0, h01 = 0, h11 = 0, c00 = 0, c01 = 0, c11 = 0;
        for (int j = j0; j < P.V; j += dj) {
            const double px = x_0 - (oth(j, 0) + k * oth(j, 2));
            const double py = x_1 - (oth(j, 1) + k * oth(j, 3));
            const double d2 = fma(px, px, py * py);
            const double rd = frsqrt(d2), d = d2 * rd;
            const double cst = (d2 < 1.0 ? c.fresh(1000.0) : c.fresh(100.0)) * P.w_distance;   // same expression as wall_slack() + 1
            const double rde = frcp(d + c.fresh(1e-6));
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
    // slack |p - o_jk|^2 - 1 of the wall constraint of node k and vehicle j at position (x_0, x_1), gradient (nx, ny)
    MPC_HD double wall_slack(int k, double x_0, double x_1, int j, double &nx, double &ny) const {
        const double px = x_0 - (oth(j, 0) + k * oth(j, 2));
        const double py = x_1 - (oth(j, 1) + k * oth(j, 3));
        nx = 2.0 * px;
        ny = 2.0 * py;
        return fma(px, px, py * py) - 1.0;
    }
    // scaled stage-cost gradient at node k of trajectory buffer cb
    MPC_HD void cost_grad(int cb, int k, double *lx) const {
        const int B = cb * 6;
        const double x_3 = S(k, B + W_X + 3);
        double g[4];
        track(k, S(k, B + W_X + 0), S(k, B + W_X + 1), S(k, B + W_X + 2), x_3, g);
        lx[0] = SF() * g[0];
        lx[1] = SF() * g[1];
        lx[2] = SF() * g[2];
        lx[3] = SF() * g[3];
        if (CC) {
            lx[0] += S(k, W_LX + 0);
            lx[1] += S(k, W_LX + 1);
            lx[3] += SF() * 2.0 * WCOLL() * x_3;
        }
    }

    // ---- objective / barrier terms of one stage of a trajectory stored at L[k * SL + (e < 4 ? bx : bu) + e],
    //      e = 0..3 state of node k, 4..5 control of stage k (trial_x / trial_u): control cost of stage k, tracking /
    //      collision cost and barrier of node k + 1
    MPC_HD void stage_terms(int bx, int bu, int k, double &Jk, double &bark) const {
        const int stride = SL, base = bu;
        const double u0 = c.ld(base + k * stride + 4), u1 = c.ld(base + k * stride + 5);
        const double hundredth = c.fresh(0.01);
        double J = hundredth * SF() * WC() * (u0 * u0 + u1 * u1);
        if (k >= 1) {
            const double d0 = u0 - c.ld(base + (k - 1) * stride + 4), d1 = u1 - c.ld(base + (k - 1) * stride + 5);
            J += hundredth * SF() * WD() * (d0 * d0 + d1 * d1);
        }
        const int nb = bx + (k + 1) * stride;
        const double y0 = c.ld(nb + 0), y1 = c.ld(nb + 1), y2 = c.ld(nb + 2), y3 = c.ld(nb + 3);
        double slack = (((u0 - ulo(0)) * (uhi(0) - u0)) * ((u1 - ulo(1)) * (uhi(1) - u1))) *
                       (((y2 - xlo(0)) * (xhi(0) - y2)) * ((y3 - xlo(1)) * (xhi(1) - y3)));
        if (k + 1 < N) {
            J += SF() * track(k + 1, y0, y1, y2, y3, (double *)nullptr);
            if (CC) {
                J += SF() * (dist(k + 1, y0, y1, (double *)nullptr) + WCOLL() * y3 * y3);
                if (any_wall) {
                    const double wjv = S(k + 1, W_WJ);
                    if (wjv >= 0.0) {
                        double nx, ny;
                        slack *= wall_slack(k + 1, y0, y1, (int)wjv, nx, ny);
                    }
                }
            }
        }
        Jk = J;
        bark = -flog(slack);
    }

    // ---- initial rollout of the controls already in buffer 0: serial dynamics (uniform) + stage-parallel cost.
    // project (cold start only): a node whose heading would sit on or outside its bound - an ego that drives along the
    // exit straight has theta_0 = -pi to float32 rounding, 5.6e-8 OUTSIDE [-pi, pi] (1 + 1e-8), and zero controls keep
    // every node there - gets the steering angle that puts it kInitPush inside (less when the vehicle is slow).  IPOPT
    // does the same to its starting point (bound_push) and accepts the 5.6e-8 at node 0 as a constraint violation below
    // its tolerance; without this such an ego was answered with status 3 and the action (0, 0) step after step.
    MPC_HD bool rollout_init(double &Jout, double &barout, bool project) {
        double x_0 = x0[0], x_1 = x0[1], x_2 = x0[2], x_3 = x0[3];
        const TrigCoef K = trig();
        const double tlo_ = xlo(0), thi_ = xhi(0), vlo_ = xlo(1), vhi_ = xhi(1);
#pragma unroll 1
        for (int k = 0; k < N; ++k) {
            const double u0 = S(k, W_U + 0);
            double u1 = S(k, W_U + 1);
            S(k, W_X + 0, x_0);
            S(k, W_X + 1, x_1);
            S(k, W_X + 2, x_2);
            S(k, W_X + 3, x_3);
            double Sn, Cn, sb, cb_;
            dyn_eval(K, x_2, u1, Sn, Cn, sb, cb_);
            double n2 = x_2 + dt * (x_3 * kInvWheelbase * sb);
            if (project && (!(n2 > tlo_) || !(n2 < thi_)) && x_3 > 1e-6) {
                const double reach = dt * x_3 * kInvWheelbase;             // |theta_{k+1} - theta_k| <= reach sin(beta)
                const double push = fmin2(kInitPush, 0.25 * reach);
                const double target = !(n2 > tlo_) ? tlo_ + push : thi_ - push;
                const double sreq = (target - x_2) / reach;
                if (fabs(sreq) < 0.9) {
                    u1 = warm_clamp(atan_b(K, 2.0 * sreq * frsqrt(1.0 - sreq * sreq)), 1);
                    S(k, W_U + 1, u1);
                    dyn_eval(K, x_2, u1, Sn, Cn, sb, cb_);
                    n2 = x_2 + dt * (x_3 * kInvWheelbase * sb);
                }
            }
            const double n0 = x_0 + dt * (x_3 * Cn);
            const double n1 = x_1 + dt * (x_3 * Sn);
            const double n3 = x_3 + dt * u0;
            if (!(n2 > tlo_) || !(n2 < thi_) || !(n3 > vlo_) || !(n3 < vhi_)) return false;
            x_0 = n0;
            x_1 = n1;
            x_2 = n2;
            x_3 = n3;
        }
        S(N, W_X + 0, x_0);
        S(N, W_X + 1, x_1);
        S(N, W_X + 2, x_2);
        S(N, W_X + 3, x_3);
        c.phase([&](int lane) {
            red_a.at(lane) = 0.0;
            red_b.at(lane) = 0.0;
            if (lane >= N) return;
            stage_terms(0, 0, lane, red_a.at(lane), red_b.at(lane));
        });
        Jout = c.wave_sum(red_a);
        barout = c.wave_sum(red_b);
        return true;
    }

    // the table of constants in LDS that every part of the solver reads bounds and function coefficients from
    MPC_HD void init_tables() {
        c.phase([&](int lane) {
            for (int k = lane; k <= N; k += kLanes) {
                S(k, W_ONE, 1.0);
                S(k, W_DTC, dt);
            }
        });
        sc(SC_SPARE + 0, 0.0);
        sc(SC_SPARE + 1, 1.0);
        sc(SC_SPARE + 2, dt);
        for (int i = 0; i < kTrigWords; ++i) sc(SC_TRIG + i, trig_coef(i));
        for (int i = 0; i < kLogWords; ++i) sc(SC_LOG + i, log_coef(i));
        sc(SC_K + K_SF, 1.0);            // until the objective scale is known (solve())
        sc(SC_NB + 0, -kNoBound);
        sc(SC_NB + 1, 0.0);
        sc(SC_NB + 2, kNoBound);
        for (int i = 0; i < 2; ++i) {
            sc(SC_BND + 0 + 2 * i, xlo_r(i));
            sc(SC_BND + 1 + 2 * i, xhi_r(i));
            sc(SC_BND + 4 + 2 * i, ulo_r(i));
            sc(SC_BND + 5 + 2 * i, uhi_r(i));
        }
    }

    // ---- diagnostics (mpc_eval_nlp): the NLP's functions at a GIVEN point z = (X, U), through the code the solver itself
    // runs - so that the reference's own f(z), g(z) (tests/golden/reference_sequences.npz: agents/pure_mpc.py:128-283 executed
    // numerically) pin the device's objective and model directly, not through a solution.  The caller has stored the point in
    // trajectory buffer 0 (X[k] at W_X, U[k] at W_U), the reference speeds and the other vehicles as for solve().  Returns the
    // UNSCALED objective: stage_terms() - what every line-search trial is judged by - for the control costs and the nodes
    // 1 .. N-1, plus the node-0 term, a constant of the solve that the solver never needs (same track() / dist()).
    // xn[i] of lane k = component i of the model's successor of (X[k], U[k]) with the formulas of rollout_init(); the
    // reference's dynamics constraint (:220-257) is X[k+1] - xn[k] = 0.
    MPC_HD double evaluate(PerLane<double> (&xn)[4]) {
        init_tables();
        any_wall = 0;
        c.phase([&](int lane) {
            red_a.at(lane) = 0.0;
            for (int i = 0; i < 4; ++i) xn[i].at(lane) = 0.0;
            if (lane >= N) return;
            double J, bar;
            stage_terms(0, 0, lane, J, bar);
            const double x_0 = S(lane, W_X + 0), x_1 = S(lane, W_X + 1), x_2 = S(lane, W_X + 2), x_3 = S(lane, W_X + 3);
            if (lane == 0) {
                J += SF() * track(0, x_0, x_1, x_2, x_3, (double *)nullptr);
                if (CC) J += SF() * (dist(0, x_0, x_1, (double *)nullptr) + WCOLL() * x_3 * x_3);
            }
            red_a.at(lane) = J;
            double Sn, Cn, sb, cb_;
            dyn_eval(trig(), x_2, S(lane, W_U + 1), Sn, Cn, sb, cb_);
            xn[0].at(lane) = x_0 + dt * (x_3 * Cn);
            xn[1].at(lane) = x_1 + dt * (x_3 * Sn);
            xn[2].at(lane) = x_2 + dt * (x_3 * kInvWheelbase * sb);
            xn[3].at(lane) = x_3 + dt * S(lane, W_U + 0);
        });
        return c.wave_sum(red_a);
    }

    // the serial part of the line search: all trials and the linearised step through the N stages (see line_search)
    MPC_HD void rollouts(const int CB, const int TB, const int W_PRE, const double idt, const double frac_wall, unsigned long long &bad) {
        PerLane<double> ZU;                                    // the rows' state, see above
        PerLane<double> ALPHA, MNL, WDEL, WTH, ISS, ISC, FW0, FW1, FW2, W3DT, F2, LOABS, HIABS, P0, P1, P2, P3, P4, P5;
        PerLane<int> o_zc, o_ck, o_g, o_q, a_pb, d_pb, o_f, o_st, m_un, m_an, is_lin, is_th, big;
        c.lanes([&](int lane_) {
            const int lane = c.opaque(lane_);
            const int q = lane & 15, t = lane >> 4;
            const bool st4 = q < 4, ct = q == 4 || q == 5, ls4 = q >= 8 && q < 12, lc = q == 12 || q == 13;
            double alpha = 1.0;
            for (int j = 0; j < t; ++j) alpha *= 0.25;
            ALPHA.at(lane_) = lc ? 1.0 : alpha;                 // the linearised step is the Newton step itself
            MNL.at(lane_) = q < 8 ? 1.0 : 0.0;                  // lanes of the nonlinear trial
            WDEL.at(lane_) = q < 2 ? 0.5 : 0.0;                 // lanes 0, 1: sin, cos of delta / 2
            WTH.at(lane_) = (q == 2 || q == 3) ? 0.25 : 0.0;    // lanes 2, 3: sin, cos of theta / 4
            ISS.at(lane_) = (q & 1) ? 0.0 : 1.0;
            ISC.at(lane_) = (q & 1) ? 1.0 : 0.0;
            FW0.at(lane_) = q == 0 ? 1.0 : 0.0;                 // x' = x + dt v cos(theta + beta)
            FW1.at(lane_) = q == 1 ? 1.0 : 0.0;                 // y' = y + dt v sin(theta + beta)
            FW2.at(lane_) = q == 2 ? kInvWheelbase : 0.0;       // theta' = theta + dt v sin(beta) / L
            W3DT.at(lane_) = q == 3 ? dt : 0.0;                 // v' = v + dt a
            F2.at(lane_) = q == 11 ? dt : 0.0;                  // its linearisation: the d a column of F, (0, 0, 0, dt)
            // (the bounds the feasibility test of theta, v is made against; every other lane: a test that never fails)
            LOABS.at(lane_) = q == 2 ? xlo(0) : (q == 3 ? xlo(1) : -kNoBound);
            HIABS.at(lane_) = q == 2 ? xhi(0) : (q == 3 ? xhi(1) : kNoBound);
            const int tw = SCR + SC_TRIG + ((q & 1) ? 6 : 0);   // sine lanes take the sine kernel's coefficients, cosine lanes the cosine's
            P0.at(lane_) = c.ld(tw + 0); P1.at(lane_) = c.ld(tw + 1); P2.at(lane_) = c.ld(tw + 2);
            P3.at(lane_) = c.ld(tw + 3); P4.at(lane_) = c.ld(tw + 4); P5.at(lane_) = c.ld(tw + 5);
            // stage-relative LDS words of this lane (a lane whose value is never used reads a word that is always initialised)
            // what ZU is compared with: the node's state; the control lanes hold the PREVIOUS stage's control when a stage begins
            // (stage 0: nothing, the constant 0 - see load_gains)
            o_zc.at(lane_) = st4 ? CB + W_X + q : (ct ? CB + W_U + (q - 4) - SL : CB + W_X);
            o_ck.at(lane_) = ct ? CB + W_U + (q - 4) : CB + W_U;                                // current control of the stage
            o_g.at(lane_) = W_KX + ((q == 5 || q == 13) ? 1 : 0);                               // this lane's row of the gains
            o_q.at(lane_) = W_PRE + PQ + ((q >= 2 && q < 6) ? q - 2 : 0);                       // its column of the PQ table
            // its projection box: theta (lane 2) and a (lane 4) have one per stage, delta (lane 5) the constants -+kNoBound; the
            // other lanes' values are never used.  a_pb: LDS word at stage 0, d_pb: what a stage adds
            a_pb.at(lane_) = q == 5 ? SCR + SC_NB : W_PRE + PB + (q == 4 ? 1 : 0);
            d_pb.at(lane_) = q == 5 ? 0 : SL;
            o_f.at(lane_) = W_LIN + (ls4 ? q - 8 : 3);      // its row of F[c][i] (the others: the row of v, which is zeros)
            // where the lane's component is stored: the trial's area, or (row 0) where the linearised step is parked
            const int tx = t == 0 ? TB : (t == 1 ? W_LIN : (t == 2 ? W_LIN + 6 : W_KX));
            int so = -1;
            so = (st4 || ct) ? tx + q : so;
            if (t == 0) {
                so = (q == 8 || q == 9) ? ((CC && any_wall) ? W_DXY + (q - 8) : -1) : so;
                so = (q == 10 || q == 11) ? W_Y + 2 + (q - 10) : so;
                so = lc ? W_Y + (q - 12) : so;
            }
            o_st.at(lane_) = so;
            m_un.at(lane_) = c.hide(ct ? -1 : 0);      // lanes that commit the trial's clamped control / the linearised control step
            m_an.at(lane_) = c.hide(lc ? -1 : 0);
            is_lin.at(lane_) = q >= 8 ? 1 : 0;
            is_th.at(lane_) = q == 2 ? 1 : 0;
            // (as a weighted sum: a select chain over x0[] becomes an indexed load, and an indexed load of a member puts the
            // whole solver object into scratch memory)
            ZU.at(lane_) = (q == 0 ? 1.0 : 0.0) * x0[0] + (q == 1 ? 1.0 : 0.0) * x0[1] + (q == 2 ? 1.0 : 0.0) * x0[2] +
                           (q == 3 ? 1.0 : 0.0) * x0[3];
        });
        const double vmin_ = c.fresh(1e-6);
        // which nodes 1 .. N - 1 carry a wall constraint, as a bit mask read once in front of the loop (round 6: until then the
        // loop asked LDS in EVERY stage of an instance with a wall anywhere - a wave-uniform load with nothing to hide its ~80
        // cycles behind, 1.6 k cycles per line search on exactly the instances a batch waits for)
        unsigned long long wall_nodes = 0;
        if (CC && any_wall) {
            PerLane<int> has;
            c.lanes([&](int lane) { has.at(lane) = (lane >= 1 && lane < N && S(lane, W_WJ) >= 0.0) ? 1 : 0; });
            wall_nodes = c.ballot(has);
        }
        // operands of a stage, requested one stage ahead (below): a lone wave cannot hide the LDS latency at the top of a stage
        PerLane<double> ZC, CK, G0, G1, G2, G3, G4, G5, G6, Q0, Q1, Q2, Q3, F0, F1, F3;
        auto load_gains = [&](int base) __attribute__((always_inline)) {
            c.lanes([&](int lane) {
                const int g = base + o_g.at(lane);
                // (stage 0: the control lanes' word would lie before the instance's memory - as an unsigned number beyond it - and
                // the minimum with the address of the constant 0 behind the stages picks that; every other word lies below it)
                const unsigned zw = (unsigned)(SCR + SC_SPARE + 0), az = (unsigned)(base + o_zc.at(lane));
                ZC.at(lane) = c.ld((int)(az < zw ? az : zw));
                CK.at(lane) = c.ld(base + o_ck.at(lane));
                G0.at(lane) = c.ld(g + 0); G1.at(lane) = c.ld(g + 2); G2.at(lane) = c.ld(g + 4); G3.at(lane) = c.ld(g + 6);
                G4.at(lane) = c.ld(g + 8); G5.at(lane) = c.ld(g + 10); G6.at(lane) = c.ld(g + 12);
            });
        };
        auto load_bounds = [&](int base) __attribute__((always_inline)) {
            c.lanes([&](int lane) {
                const int qq = base + o_q.at(lane), f = base + o_f.at(lane);
                Q0.at(lane) = c.ld(qq + 0); Q1.at(lane) = c.ld(qq + 4);
                Q2.at(lane) = c.ld(a_pb.at(lane) + 0); Q3.at(lane) = c.ld(a_pb.at(lane) + 2);
                a_pb.at(lane) += d_pb.at(lane);
                F0.at(lane) = c.ld(f + 0); F1.at(lane) = c.ld(f + 4); F3.at(lane) = c.ld(f + 8);
            });
        };
#pragma unroll 1
        for (int k = 0; k < N; ++k) {
            const int base = k * SL;
            load_gains(base);
            load_bounds(base);
            PerLane<double> E, ACC, AN, AL, T, V, TH, A, B, DTV, LOV, HIV, UN1, NX, BASE;
            // ---- e = ZU - current iterate (the linearised lanes: ZU itself)
            c.lanes([&](int lane) {
                E.at(lane) = ZU.at(lane) - MNL.at(lane) * ZC.at(lane);
                ACC.at(lane) = ALPHA.at(lane) * G6.at(lane);
                big.at(lane) = !(fabs(E.at(lane)) < c.fresh(kOpenLoopStep)) ? 1 : 0;
            });
            // (viii) a linearised step that reaches this stage below kOpenLoopStep is applied open loop (wave-uniform decision)
            // (lanes 8..13 of every row: the linearised step)
            const bool open_loop = (c.ballot(big) & 0x3f003f003f003f00ull) == 0;
            // ---- what depends on theta_k, v_k only - known since the previous stage, so all of this is off the stage's critical
            //      path (position -> feedback -> delta -> sin / cos(theta + beta) -> position): sin, cos of theta (lanes 2, 3: the
            //      fdlibm kernels at theta / 4, two angle doublings) folded into the coefficients (A, B) of (cos beta, sin beta) in
            //      this lane's model row - x: (cos theta, -sin theta), y: (sin theta, cos theta), theta: (0, 1 / L) - and dt v
            c.template row_bcast<2>(TH, ZU);
            c.template row_bcast<3>(V, ZU);
            {
                PerLane<double> R, SY, CY;
                c.lanes([&](int lane) {
                    const double r = WTH.at(lane) * TH.at(lane), z = r * r, z2 = z * z;
                    const double pa = fma(z, P0.at(lane), P1.at(lane)), pb = fma(z, P2.at(lane), P3.at(lane)), pc = fma(z, P4.at(lane), P5.at(lane));
                    const double p = fma(z2, fma(z2, pa, pb), pc);
                    const double m = ISS.at(lane) * r + ISC.at(lane) * z;
                    const double a0 = ISS.at(lane) * r + ISC.at(lane) * fma(-0.5, z, 1.0);
                    R.at(lane) = fma(z * m, p, a0);
                });
                c.template row_bcast<2>(SY, R);
                c.template row_bcast<3>(CY, R);
                c.lanes([&](int lane) {
                    const double sy = SY.at(lane), cy = CY.at(lane);
                    const double s2 = 2.0 * sy * cy, c2 = fma(-2.0 * sy, sy, 1.0);     // theta / 2
                    const double st = 2.0 * s2 * c2, ctt = fma(-2.0 * s2, s2, 1.0);
                    A.at(lane) = FW0.at(lane) * ctt + FW1.at(lane) * st;
                    B.at(lane) = FW1.at(lane) * ctt - FW0.at(lane) * st + FW2.at(lane);
                    DTV.at(lane) = dt * V.at(lane);
                    LOV.at(lane) = (Q2.at(lane) - V.at(lane)) * idt;       // lane 4: the accelerations that keep v of node k + 1 in its box
                    HIV.at(lane) = (Q3.at(lane) - V.at(lane)) * idt;
                });
            }
            // ---- feedback law of the trial (lanes 4, 5) and linearised control step (lanes 12, 13): the same gains
            //      - ONE accumulator: lanes 0..7 of a row hold the trial's sum, lanes 8..15 the linearised step's (row_bcast2)
            c.template row_bcast2<4>(T, E); c.lanes([&](int lane) { AN.at(lane) = ACC.at(lane) + G4.at(lane) * T.at(lane); });
            c.template row_bcast2<5>(T, E); c.lanes([&](int lane) { AN.at(lane) += G5.at(lane) * T.at(lane); });
            c.template row_bcast2<3>(T, E); c.lanes([&](int lane) { AN.at(lane) += G3.at(lane) * T.at(lane); });
            c.template row_bcast2<2>(T, E); c.lanes([&](int lane) { AN.at(lane) += G2.at(lane) * T.at(lane); });
            c.template row_bcast2<0>(T, E); c.lanes([&](int lane) { AN.at(lane) += G0.at(lane) * T.at(lane); });
            c.template row_bcast2<1>(T, E); c.lanes([&](int lane) { AN.at(lane) += G1.at(lane) * T.at(lane); });
            if (open_loop) {
                PerLane<double> T2;
                c.template row_bcast<12>(T, AN);
                c.template row_bcast<13>(T2, AN);
                c.lanes([&](int lane) {
                    const double ol = ALPHA.at(lane) * ((lane & 1) ? T2.at(lane) : T.at(lane));
                    AN.at(lane) = is_lin.at(lane) ? AN.at(lane) : ol;
                });
            }
            if (fine_ticks<CTX>::value) c.tick(T_R_FEEDBACK);
            // ---- the trial's controls, clamped to the fraction-to-the-boundary box.  delta is final here (the model step waits
            //      for it); a goes on: v of node k + 1 is decided by a_k alone and is kept inside the node's box (LOV, HIV of
            //      lane 4; none for delta).  The linearised step takes no clamp.
            c.lanes([&](int lane) { UN1.at(lane) = fmin2(fmax2(CK.at(lane) + AN.at(lane), Q0.at(lane)), Q1.at(lane)); });
            c.lanes([&](int lane) {
                const double a = fmin2(fmax2(UN1.at(lane), LOV.at(lane)), HIV.at(lane));
                const double un = fmin2(fmax2(a, Q0.at(lane)), Q1.at(lane));
                ZU.at(lane) = c.bit_select(m_un.at(lane), un, c.bit_select(m_an.at(lane), AN.at(lane), ZU.at(lane)));
            });
            c.template row_bcast<4>(T, ZU);
            c.lanes([&](int lane) { BASE.at(lane) = ZU.at(lane) + W3DT.at(lane) * T.at(lane); });      // v' = v + dt a; the others: ZU
            if (fine_ticks<CTX>::value) c.tick(T_R_CLAMP);
            // ---- model step: sin, cos of delta (lanes 0, 1: the kernels at delta / 2, one doubling), beta = atan(tan(delta) / 2) as
            //      (cos beta, sin beta) = q (2 cos delta, sin delta), q = (3 cos^2 delta + 1)^-1/2, and every state lane its row:
            //      x' = x + dt v q (A 2 cos delta + B sin delta).  A second pass follows a projection (below).
            PerLane<int> pj, redo;
            PerLane<double> KEEP, DSRC;
            c.lanes([&](int lane) {
                redo.at(lane) = 0;
                DSRC.at(lane) = UN1.at(lane);
            });
#pragma unroll 1
            for (int pass = 0;; ++pass) {
                PerLane<double> D, R, SX, CX;
                c.template row_bcast<5>(D, DSRC);
                c.lanes([&](int lane) {
                    const double r = WDEL.at(lane) * D.at(lane), z = r * r, z2 = z * z;
                    const double pa = fma(z, P0.at(lane), P1.at(lane)), pb = fma(z, P2.at(lane), P3.at(lane)), pc = fma(z, P4.at(lane), P5.at(lane));
                    const double p = fma(z2, fma(z2, pa, pb), pc);
                    const double m = ISS.at(lane) * r + ISC.at(lane) * z;
                    const double a0 = ISS.at(lane) * r + ISC.at(lane) * fma(-0.5, z, 1.0);
                    R.at(lane) = fma(z * m, p, a0);
                });
                c.template row_bcast<0>(SX, R);
                c.template row_bcast<1>(CX, R);
                c.lanes([&](int lane) {
                    const double sx = SX.at(lane), cx = CX.at(lane);
                    const double cd = fma(-2.0 * sx, sx, 1.0), sxcx = sx * cx;       // cos delta, sin delta / 2
                    const double qn = frsqrt(fma(3.0 * cd, cd, 1.0));
                    const double g = 2.0 * (A.at(lane) * cd + B.at(lane) * sxcx);
                    const double n = fma(DTV.at(lane), qn * g, BASE.at(lane));
                    // second pass (after a projection): the rows it did not concern keep what the first pass gave them
                    NX.at(lane) = (pass == 1 && !redo.at(lane)) ? KEEP.at(lane) : n;
                    pj.at(lane) = (is_th.at(lane) && ((n < Q2.at(lane)) | (n > Q3.at(lane))) && V.at(lane) > vmin_) ? 1 : 0;
                });
                if (pass == 1 || c.ballot(pj) == 0) break;
                // theta of node k + 1 is decided by delta_k alone (theta + dt v / L sin beta(delta)): a trial whose node leaves the
                // box gets the delta that puts it on the edge of the box (rare; the rows it does not concern keep their delta)
                PerLane<double> N2, TLO, THI, DLO, DHI, FL, PJF;
                c.lanes([&](int lane) { PJF.at(lane) = pj.at(lane) ? 1.0 : 0.0; });
                c.template row_bcast<2>(FL, PJF);
                c.template row_bcast<2>(N2, NX);
                c.template row_bcast<2>(TLO, Q2);
                c.template row_bcast<2>(THI, Q3);
                c.template row_bcast<5>(DLO, Q0);
                c.template row_bcast<5>(DHI, Q1);
                const TrigCoef K = trig();
                c.lanes([&](int lane) {
                    KEEP.at(lane) = NX.at(lane);
                    DSRC.at(lane) = ZU.at(lane);
                    if (FL.at(lane) == 0.0) return;
                    const double tgt = N2.at(lane) < TLO.at(lane) ? TLO.at(lane) : THI.at(lane);
                    // (keep(): the reciprocal stays in this rare path - a loop invariant of the pass loop otherwise, computed in every stage)
                    const double sreq = (tgt - TH.at(lane)) * (1.0 / kInvWheelbase) * frcp(dt * c.keep(V.at(lane)));
                    if (fabs(sreq) < 0.9) {
                        const double u1 = fmin2(fmax2(atan_b(K, 2.0 * sreq * frsqrt(1.0 - sreq * sreq)), DLO.at(lane)), DHI.at(lane));
                        if ((lane & 15) == 5) {
                            ZU.at(lane) = u1;
                            DSRC.at(lane) = u1;
                            BASE.at(lane) = u1;      // a control lane's "next state" is the control itself (it is committed below)
                        }
                        redo.at(lane) = 1;
                    }
                });
                if (fine_ticks<CTX>::value) c.tick(T_R_PROJ);
                if (c.ballot(redo) == 0) break;
            }
            if (fine_ticks<CTX>::value) c.tick(T_R_DYN);
            // ---- every lane stores its component of node / stage k (the trial's area; row 0: the parked linearised step)
            c.lanes([&](int lane) {
                const int so = o_st.at(lane);
                if (so >= 0) c.st(base + so, ZU.at(lane));
            });
            if (fine_ticks<CTX>::value) c.tick(T_R_LDSW);
            // ---- linearised state step d' = d + F (d theta, d v, d a, d delta) in lanes 8..11 (the other lanes' rows are zeros)
            c.template row_bcast<10>(T, ZU); c.lanes([&](int lane) { AL.at(lane) = F0.at(lane) * T.at(lane); });
            c.template row_bcast<11>(T, ZU); c.lanes([&](int lane) { AL.at(lane) += F1.at(lane) * T.at(lane); });
            c.template row_bcast<12>(T, ZU); c.lanes([&](int lane) { AL.at(lane) += F2.at(lane) * T.at(lane); });
            c.template row_bcast<13>(T, ZU); c.lanes([&](int lane) { AL.at(lane) += F3.at(lane) * T.at(lane); });
            // ---- commit (control lanes: NX = ZU), feasibility of theta, v of node k + 1 against the fraction-to-the-boundary
            //      margins.  A trial that leaves its box is infeasible; it is integrated to the end all the same (its lanes would
            //      idle otherwise) and whatever it computes from here on is never looked at (bounded polynomials, rsq / rcp of
            //      garbage give NaN at worst).
            PerLane<int> viol;
            c.lanes([&](int lane) {
                const double n = NX.at(lane) + AL.at(lane);
                ZU.at(lane) = n;
                viol.at(lane) = ((n - LOABS.at(lane) < Q0.at(lane)) | (HIABS.at(lane) - n < Q1.at(lane))) ? 1 : 0;
            });
            bad |= c.ballot(viol);
            if (CC && k + 1 < N && ((wall_nodes >> (k + 1)) & 1ull)) {      // (k + 1 < N <= 64: the shift stays below 64)
                const double wjv = S(k + 1, W_WJ);
                if (wjv >= 0.0) {
                    PerLane<double> X0, X1;
                    c.template row_bcast<0>(X0, ZU);
                    c.template row_bcast<1>(X1, ZU);
                    const double lim = frac_wall * S(k + 1, W_GW);
                    c.lanes([&](int lane) {
                        double nx, ny;
                        viol.at(lane) = wall_slack(k + 1, X0.at(lane), X1.at(lane), (int)wjv, nx, ny) < lim ? 1 : 0;
                    });
                    bad |= c.ballot(viol);
                }
            }
            if (fine_ticks<CTX>::value) c.tick(T_R_CHECK);
        }
        // the last node: x_N of the trials, d x_N of the linearised step
        c.lanes([&](int lane) {
            const int so = o_st.at(lane);
            if (so >= 0 && (lane & 15) != 4 && (lane & 15) != 5 && (lane & 15) < 12) c.st(N * SL + so, ZU.at(lane));
        });
    }

    // ---- line search on the barrier objective (Armijo, kTrials step lengths alpha_t = 4^-t, first passing wins).
    // Trial controls u_k = ucur_k + alpha kf_k + Kp_k (u_{k-1} - ucur_{k-1}) + Kx_k (x_k - xcur_k), clamped to the
    // fraction-to-the-boundary box and projected into the state bounds of the next node.
    //
    // ROW-COOPERATIVE ROLLOUT (round 5).  Until round 4 lane t integrated trial t on its own - four active lanes of 64, 335
    // instructions per stage, 53 % of an iteration, and an FP64 instruction occupies the vector unit for four cycles however
    // many lanes are active.  Now trial t owns the 16-lane DPP row t and the lanes of a row are the COMPONENTS of the stage:
    //     q = 0..3   x, y, theta, v of the trial's node k            q = 4, 5    a, delta of the trial (stage k - 1, then stage k)
    //     q = 8..11  linearised step d x, d y, d theta, d v          q = 12, 13  linearised control step (the same in every row)
    // all in ONE register ZU.  What couples components is a matrix-vector product, and gfx950 has the instruction for it:
    // v_mov_b64_dpp row_newbcast:j hands lane j of a row to all its lanes (CTX::row_bcast), so with row i of a matrix in lane i
    //     y_i += M[i][j] * bcast_j(x)        is one broadcast + one FMA for every row of the matrix and all four trials at once.
    // The feedback law (2 x 7), the model step (4 state rows) and the linearised recursion d' = A d + B du (4 x 4 nontrivial
    // entries, the F[c][i] table of W_LIN) are such products; the four Horner chains of the trigonometry (sin, cos of delta / 2 and
    // of theta / 4) run in lanes 0..3 with per-lane coefficients; bounds, margins and boxes are per-lane values loaded from the
    // PQ table instead of per-instruction constants; every lane stores its own component with one ds_write.  The linearised
    // Newton step (the dual step needs it) is the "fifth trial" in lanes 8..13: same gains, same instruction stream, its own
    // broadcasts.  ~140 instructions per stage.  Same arithmetic as before except for the association of a few sums (the
    // feedback law adds its terms in the order kf, Kp, Kx[v], Kx[theta], Kx[x], Kx[y] - what arrives last is added last).
    MPC_HD bool line_search(int cur, double frac, double keep, double phi0, double dV1, double mu_, double &Jn, double &barn,
                            int &acc_out) {
        const double a_pr = 1.0;
        const int CB = cur * 6, TB = (cur ^ 1) * 6;
        const double fracu = 2.0 * frac, idt = frcp(c.fresh(dt));
        constexpr int W_PRE = CC ? W_SLOTS_CC : W_SLOTS;
        // ---- what the trials' stages need of the current iterate and frac alone: stage-parallel, once per line search
        c.phase([&](int lane) {
            if (lane >= N) return;
            const int k = lane;
            const double tlo_ = xlo(0), thi_ = xhi(0), vlo_ = xlo(1), vhi_ = xhi(1);
            const double alo_ = ulo(0), ahi_ = uhi(0), dlo_ = ulo(1), dhi_ = uhi(1);
            const double ms_ = c.fresh(kMinSlack);
            const double c0 = S(k, CB + W_U + 0), c1 = S(k, CB + W_U + 1);
            const double o2 = S(k + 1, CB + W_X + 2), o3 = S(k + 1, CB + W_X + 3);
            S(k, W_PRE + PQ + 0, fmax2(frac * (o2 - tlo_), ms_));
            S(k, W_PRE + PQ + 1, fmax2(frac * (o3 - vlo_), ms_));
            S(k, W_PRE + PQ + 2, alo_ + fmax2(fracu * (c0 - alo_), ms_));
            S(k, W_PRE + PQ + 3, dlo_ + fmax2(fracu * (c1 - dlo_), ms_));
            S(k, W_PRE + PQ + 4, fmax2(frac * (thi_ - o2), ms_));
            S(k, W_PRE + PQ + 5, fmax2(frac * (vhi_ - o3), ms_));
            S(k, W_PRE + PQ + 6, ahi_ - fmax2(fracu * (ahi_ - c0), ms_));
            S(k, W_PRE + PQ + 7, dhi_ - fmax2(fracu * (dhi_ - c1), ms_));
            S(k, W_PRE + PB + 0, tlo_ + keep * (o2 - tlo_));
            S(k, W_PRE + PB + 1, vlo_ + keep * (o3 - vlo_));
            S(k, W_PRE + PB + 2, thi_ - keep * (thi_ - o2));
            S(k, W_PRE + PB + 3, vhi_ - keep * (vhi_ - o3));
        });
        unsigned long long bad = 0;     // bit l: lane l (theta or v of trial l / 16, or any lane of its row for a wall) left its box
        rollouts(CB, TB, W_PRE, idt, frac, bad);
        c.lanes([&](int lane) { ls_feas.at(lane) = ((bad >> (lane & ~15)) & 0xffffull) == 0 ? 1 : 0; });
        c.tick(T_ROLL_DYN);
        int acc = -1;
        if (N <= kLanes / 2) {
            // two trials per pass: lanes 0..31 / 32..63 = stages of trial 2p / 2p + 1
            double alpha = a_pr;
#pragma unroll 1
            for (int p = 0; p < kTrials / 2 && acc < 0; ++p, alpha *= 0.0625) {
                const int f0 = c.wave_bcast(ls_feas, 16 * (2 * p)), f1 = c.wave_bcast(ls_feas, 16 * (2 * p + 1));
                if (!f0 && !f1) continue;
                c.phase([&](int lane) {
                    red_a.at(lane) = 0.0;
                    red_b.at(lane) = 0.0;
                    const int h = lane >> 5, k = lane & 31, t = 2 * p + h;
                    if (k >= N || !(h ? f1 : f0)) return;
                    stage_terms(trial_x(CC, t, TB), trial_u(CC, t, TB), k, red_a.at(lane), red_b.at(lane));
                });
                double J0, J1, b0, b1;
                c.wave_sum2(red_a, J0, J1);
                c.wave_sum2(red_b, b0, b1);
                const double a1 = alpha * 0.25;
                if (f0 && J0 + mu_ * b0 <= phi0 + c.fresh(1e-4) * alpha * 2.0 * dV1 + c.fresh(1e-12) * fabs(phi0)) {
                    acc = 2 * p;
                    Jn = J0;
                    barn = b0;
                } else if (f1 && J1 + mu_ * b1 <= phi0 + c.fresh(1e-4) * a1 * 2.0 * dV1 + c.fresh(1e-12) * fabs(phi0)) {
                    acc = 2 * p + 1;
                    Jn = J1;
                    barn = b1;
                }
            }
        } else {
            // long horizons (33..64 stages): one trial per pass, lane k = stage k
            double alpha = a_pr;
#pragma unroll 1
            for (int t = 0; t < kTrials && acc < 0; ++t, alpha *= 0.25) {
                if (!c.wave_bcast(ls_feas, 16 * t)) continue;
                c.phase([&](int lane) {
                    red_a.at(lane) = 0.0;
                    red_b.at(lane) = 0.0;
                    if (lane >= N) return;
                    stage_terms(trial_x(CC, t, TB), trial_u(CC, t, TB), lane, red_a.at(lane), red_b.at(lane));
                });
                const double J0 = c.wave_sum(red_a), b0 = c.wave_sum(red_b);
                if (J0 + mu_ * b0 <= phi0 + c.fresh(1e-4) * alpha * 2.0 * dV1 + c.fresh(1e-12) * fabs(phi0)) {
                    acc = t;
                    Jn = J0;
                    barn = b0;
                }
            }
        }
        if (CC && P.V > 0 && acc != 0) {
            // vehicles that a rejected (fully integrated) trial took across d = 1 inwards, per node the one with the
            // smallest slack at the current iterate; the dual update turns them into wall constraints
            const int nrej = acc < 0 ? kTrials : acc;
            int fe[kTrials];
            for (int t = 0; t < kTrials; ++t) fe[t] = c.wave_bcast(ls_feas, 16 * t);
            c.phase([&](int lane) {
                const int k = lane;
                if (k < 1 || k >= N) return;
                const double x_0 = S(k, CB + W_X + 0), x_1 = S(k, CB + W_X + 1);
                int cross = -1;
                double gbest = INFINITY;
                for (int j = 0; j < P.V; ++j) {
                    double nx, ny;
                    const double g0 = wall_slack(k, x_0, x_1, j, nx, ny);
                    if (!(g0 > 0.0) || !(g0 < gbest)) continue;
                    bool crossed = false;
                    for (int t = 0; t < nrej; ++t) {
                        if (!fe[t]) continue;
                        const int o = trial_x(CC, t, TB) + k * SL;
                        if (wall_slack(k, c.ld(o + 0), c.ld(o + 1), j, nx, ny) < 0.0) crossed = true;
                    }
                    if (crossed) {
                        cross = j;
                        gbest = g0;
                    }
                }
                S(k, W_CROSS, (double)cross);
            });
        }
        if (acc >= 1) {
            const int bx = trial_x(CC, acc, TB), bu = trial_u(CC, acc, TB);
            c.phase([&](int lane) {
                for (int node = lane; node <= N; node += kLanes)
                    for (int e = 0; e < (node < N ? 6 : 4); ++e) S(node, TB + e, c.ld(node * SL + (e < 4 ? bx : bu) + e));
            });
        }
        c.tick(T_ROLL_COST);
        acc_out = acc;
        return acc >= 0;
    }

    // ---- the solve --------------------------------------------------------------------------------------
    // warm: buffer 0 already holds initial controls (clamped inside their bounds by the caller) instead of the
    // reference's cold start; if their rollout leaves the state bounds the cold start is used after all
    MPC_HD void solve(int &status_out, int &iters_out, int &cur_out, double &kkt_out, bool warm = false) {
        int cur = 0;
        status_out = 1;
        iters_out = 0;
        cur_out = 0;
        kkt_out = INFINITY;
        init_tables();
        // cold start of the reference (agents/pure_mpc.py:240-246: controls 0), multipliers 1
        c.phase([&](int lane) {
            if (lane >= N) return;
            for (int i = 0; i < 2; ++i) {
                S(lane, W_ZUL + i, 1.0);
                S(lane, W_ZUU + i, 1.0);
                S(lane + 1, W_ZXL + i, 1.0);
                S(lane + 1, W_ZXU + i, 1.0);
            }
            if (CC) {
                S(lane, W_ZW, 0.0);
                S(lane, W_WJ, -1.0);
            }
        });
        any_wall = 0;
        double Jcur = 0.0, barcur = 0.0;
        if (!(warm && rollout_init(Jcur, barcur, false))) {
            c.phase([&](int lane) {
                if (lane >= N) return;
                S(lane, W_U + 0, 0.0);
                S(lane, W_U + 1, 0.0);
            });
            if (x0[3] < 0.01) S(0, W_U + 0, (0.01 - x0[3]) / dt);
            if (!rollout_init(Jcur, barcur, true)) {
                status_out = 3;
                return;
            }
        }
        // objective scaling: sf = 100 / clamp(|grad f|_inf at the start, 100, 1e4)
        {
            c.phase([&](int lane) {
                double g = 0.0;
                if (lane >= 1 && lane < N) {
                    if (CC) {
                        double d8[8];
                        dist(lane, S(lane, W_X + 0), S(lane, W_X + 1), d8);
                        S(lane, W_LX + 0, d8[0]);
                        S(lane, W_LX + 1, d8[1]);
                    }
                    double lx[4];
                    cost_grad(0, lane, lx);
                    g = fmax2(fmax2(fabs(lx[0]), fabs(lx[1])), fmax2(fabs(lx[2]), fabs(lx[3])));
                }
                red_a.at(lane) = g;
            });
            const double gmax = fmax2(0.02 * (WC() + WD()) * fabs(S(0, W_U + 0)), c.wave_max(red_a));
            sf = c.uni(100.0 / fmin2(fmax2(100.0, gmax), 1e4));
            Jcur = c.uni(Jcur * sf);
            barcur = c.uni(barcur);
        }
        // wave-uniform doubles that live across the whole solve: in scalar registers (CTX::uni), not in vector registers
        // wave-uniform constants of the solve: in the LDS table (read where they are used), not in registers
        sc(SC_K + K_SF, sf);
        sc(SC_K + K_RD, 0.02 * sf * wd_);
        sc(SC_K + K_RC, 0.02 * sf * wc_);
        sc(SC_K + K_QTT, 10.0 * sf);
        sc(SC_K + K_Q33, sf * (20.0 * ws_ + (CC ? 2.0 * wcoll : 0.0)));
        sc(SC_K + K_MUMIN, P.tol / 10.0);
        double mu = P.mu_init;
        int iter = 0, nfail = 0;
        double reg = 0.0;
        // dw_last = the multiple of the identity the last inertia correction ended with (0: none yet in this solve); IPOPT's
        // algorithm IC: the first correction starts at 1e-4 and grows by 100, later ones start at a third of the last value and
        // grow by 8.  With negative cost weights (the v1 input domain) the control block needs ~1e-2 in EVERY iteration:
        // climbing 1e-8, 1e-6, ... from scratch (until round 3) ended at 1 - five wasted sweeps per iteration and steps a
        // hundred times shorter than the curvature warrants (121 iterations on average where this rule takes 40, scenario
        // c4v1).  Kept in the LDS table and read only when a correction is needed: no register across the iterations.
        sc(SC_SPARE + 3, 0.0);
        // progress guard (mpc_config.stall_window, off by default): the iteration at which the KKT error last fell below
        // half of its value at the previous such mark
        int i_mark = 0;
        double e_mark = INFINITY;
        int gn_streak = 0, gn_skip = 0;    // (vii)
        const bool nonconvex_cost = (ws_ < 0.0) | (wc_ < 0.0) | (wd_ < 0.0);      // (xi)
        bool prev_needed = false, prev_gn = false;
        double e_streak = INFINITY;
        int n_acceptable = 0;              // (ix)

        for (iter = 0; iter <= P.max_iter; ++iter) {
            const int CB = cur * 6;
            if (iter == kBoostIter) c.template set_priority<3>();
            c.tick(T_DUALUPD);
            // ============ collision potential: gradient, exact and Gauss-Newton curvature at the nodes 1 .. N - 1 - a loop over
            //              the vehicles, 130 instructions each, that occupies the vector unit for the same four cycles per
            //              instruction whether 19 lanes are active or 57.  The vehicles are dealt to up to three lane groups
            //              (lane = g (N - 1) + k - 1: group g takes the vehicles g, g + G, ...); a group leaves its partial
            //              sums in words of the stage that are dead at this point (group 0: where the result goes; 1, 2: the
            //              gains of the last sweep, the spare trajectory buffer) and the preparation phase adds them up.
            const int coll_nodes = N - 1;
            const int coll_groups = (!CC || coll_nodes < 1) ? 1 : (kLanes / coll_nodes >= 3 ? 3 : (kLanes / coll_nodes >= 2 ? 2 : 1));
            auto coll_word = [&](int g, int i) __attribute__((always_inline)) {
                return g == 0 ? W_LX + i : (g == 1 ? W_KX + i : (i < 6 ? W_KP + i : (cur ^ 1) * 6 + (i - 6)));
            };
            if (CC && coll_nodes >= 1) {
                c.phase([&](int lane) {
                    const int g = lane / coll_nodes, k = 1 + lane - g * coll_nodes;
                    if (g >= coll_groups) return;
                    double d8[8];
                    dist(k, S(k, CB + W_X + 0), S(k, CB + W_X + 1), d8, g, coll_groups);
                    for (int i = 0; i < 8; ++i) S(k, coll_word(g, i), d8[i]);
                });
            }
            // ============ stage-parallel preparation: dynamics trig, collision-potential derivatives,
            //              complementarity products
            c.phase([&](int lane) {
                if (lane >= N) {
                    red_a.at(lane) = 0.0;
                    red_b.at(lane) = INFINITY;
                    red_c.at(lane) = 0.0;
                    if (CC) red_w.at(lane) = -1.0;
                    return;
                }
                const int k = lane;
                const double u0 = S(k, CB + W_U + 0), u1 = S(k, CB + W_U + 1);
                const double xk0 = S(k, CB + W_X + 0), xk1 = S(k, CB + W_X + 1), xk2 = S(k, CB + W_X + 2),
                             xk3 = S(k, CB + W_X + 3);
                {
                    double Sn, Cn, sb, cb_, bp, bpp;
                    dyn_eval(trig(), xk2, u1, Sn, Cn, sb, cb_);
                    beta_derivs(sb, cb_, bp, bpp);
                    S(k, W_LIN + 2, 0.0); S(k, W_LIN + 3, 0.0); S(k, W_LIN + 7, 0.0); S(k, W_LIN + 11, 0.0);   // the structural zeros of F[c][i]
                    S(k, W_LIN + LIN_A02, -dt * xk3 * Sn);
                    S(k, W_LIN + LIN_A03, dt * Cn);
                    S(k, W_LIN + LIN_A12, dt * xk3 * Cn);
                    S(k, W_LIN + LIN_A13, dt * Sn);
                    S(k, W_LIN + LIN_A23, dt * sb * kInvWheelbase);
                    S(k, W_LIN + LIN_B01, -dt * xk3 * Sn * bp);
                    S(k, W_LIN + LIN_B11, dt * xk3 * Cn * bp);
                    S(k, W_LIN + LIN_B21, dt * xk3 * kInvWheelbase * cb_ * bp);
                }
                // gradient of the Lagrangian's separable part at node k (cost + bound multipliers), the start of the
                // adjoint recursion below; lane 0 supplies the terminal node
                double wall_z = -1.0, wall_c = 0.0;   // multiplier / complementarity product of the node's wall constraint
                if (k >= 1) {
                    double g[4];
                    track(k, xk0, xk1, xk2, xk3, g);
                    double lx0 = SF() * g[0], lx1 = SF() * g[1], lx2 = SF() * g[2], lx3 = SF() * g[3];
                    if (CC) {
                        double d8[8];
                        for (int i = 0; i < 8; ++i) {
                            d8[i] = S(k, coll_word(0, i));
                            if (coll_groups >= 2) d8[i] += S(k, coll_word(1, i));
                            if (coll_groups >= 3) d8[i] += S(k, coll_word(2, i));
                        }
                        S(k, W_LX + 0, SF() * d8[0]);
                        S(k, W_LX + 1, SF() * d8[1]);
                        S(k, W_Q + 0, SF() * d8[2]);
                        S(k, W_Q + 1, SF() * d8[3]);
                        S(k, W_Q + 2, SF() * d8[4]);
                        S(k, W_QG + 0, SF() * d8[5]);
                        S(k, W_QG + 1, SF() * d8[6]);
                        S(k, W_QG + 2, SF() * d8[7]);
                        lx0 += SF() * d8[0];
                        lx1 += SF() * d8[1];
                        lx3 += SF() * 2.0 * WCOLL() * xk3;
                        if (any_wall) {
                            const double wjv = S(k, W_WJ);
                            if (wjv >= 0.0) {
                                double nx, ny;
                                const double g = wall_slack(k, xk0, xk1, (int)wjv, nx, ny);
                                wall_z = S(k, W_ZW);
                                wall_c = g * wall_z;
                                lx0 -= wall_z * nx;
                                lx1 -= wall_z * ny;
                            }
                        }
                    }
                    S(k, W_Y + 0, lx0);
                    S(k, W_Y + 1, lx1);
                    S(k, W_Y + 2, lx2 - S(k, W_ZXL + 0) + S(k, W_ZXU + 0));
                    S(k, W_Y + 3, lx3 - S(k, W_ZXL + 1) + S(k, W_ZXU + 1));
                } else {
                    S(N, W_Y + 0, 0.0);
                    S(N, W_Y + 1, 0.0);
                    S(N, W_Y + 2, -S(N, W_ZXL + 0) + S(N, W_ZXU + 0));
                    S(N, W_Y + 3, -S(N, W_ZXL + 1) + S(N, W_ZXU + 1));
                }
                const double zul0 = S(k, W_ZUL + 0), zul1 = S(k, W_ZUL + 1), zuu0 = S(k, W_ZUU + 0), zuu1 = S(k, W_ZUU + 1);
                double cmx, cmn, sz = zul0 + zul1 + zuu0 + zuu1;
                {
                    const double c0 = (u0 - ulo(0)) * zul0, c1 = (uhi(0) - u0) * zuu0;
                    const double c2 = (u1 - ulo(1)) * zul1, c3 = (uhi(1) - u1) * zuu1;
                    cmx = fmax2(fmax2(c0, c1), fmax2(c2, c3));
                    cmn = fmin2(fmin2(c0, c1), fmin2(c2, c3));
                }
                for (int i = 0; i < 2; ++i) {
                    const double xi = S(k + 1, CB + W_X + 2 + i);
                    const double zl = S(k + 1, W_ZXL + i), zu = S(k + 1, W_ZXU + i);
                    const double c0 = (xi - xlo(i)) * zl, c1 = (xhi(i) - xi) * zu;
                    cmx = fmax2(cmx, fmax2(c0, c1));
                    cmn = fmin2(cmn, fmin2(c0, c1));
                    sz += zl + zu;
                }
                if (CC && wall_z >= 0.0) {
                    cmx = fmax2(cmx, wall_c);
                    cmn = fmin2(cmn, wall_c);
                    sz += wall_z;
                }
                red_a.at(lane) = cmx;
                red_b.at(lane) = cmn;
                red_c.at(lane) = sz;
                if (CC) red_w.at(lane) = wall_z;
            });
            const double cmax = c.wave_max(red_a), cmin = c.wave_min(red_b), sum_z = c.wave_sum(red_c);
            const double zw_max = (CC && any_wall) ? c.wave_max(red_w) : 0.0;
            c.tick(T_PREP);
            // ============ adjoint recursion y_k = g_k + A_k' y_{k+1}, in place over the node gradients.  A_k' = I + (strictly
            //              triangular): y0 and y1 are plain suffix sums of the node gradients, y2 a suffix sum of
            //              g2 + a02 y0+ + a12 y1+, y3 one of g3 + a03 y0+ + a13 y1+ + a23 y2+ (y+ = y_{k+1}) - three
            //              wave scans, lane j = node j + 1, instead of N dependent steps
            double sum_lam = 0.0;
            {
                PerLane<double> s0, s1;
                c.phase([&](int lane) {
                    const bool on = lane < N;
                    s0.at(lane) = on ? S(lane + 1, W_Y + 0) : 0.0;
                    s1.at(lane) = on ? S(lane + 1, W_Y + 1) : 0.0;
                });
                c.wave_suffix_sum(s0);
                c.wave_suffix_sum(s1);
                c.phase([&](int lane) {
                    if (lane >= N) return;
                    S(lane + 1, W_Y + 0, s0.at(lane));
                    S(lane + 1, W_Y + 1, s1.at(lane));
                });
                PerLane<double> s2;
                c.phase([&](int lane) {
                    const int k = lane + 1;
                    double h = 0.0;
                    if (lane < N) {
                        h = S(k, W_Y + 2);
                        if (k < N) h += S(k, W_LIN + LIN_A02) * S(k + 1, W_Y + 0) + S(k, W_LIN + LIN_A12) * S(k + 1, W_Y + 1);
                    }
                    s2.at(lane) = h;
                });
                c.wave_suffix_sum(s2);
                c.phase([&](int lane) {
                    if (lane < N) S(lane + 1, W_Y + 2, s2.at(lane));
                });
                PerLane<double> s3;
                c.phase([&](int lane) {
                    const int k = lane + 1;
                    double h = 0.0;
                    if (lane < N) {
                        h = S(k, W_Y + 3);
                        if (k < N)
                            h += S(k, W_LIN + LIN_A03) * S(k + 1, W_Y + 0) + S(k, W_LIN + LIN_A13) * S(k + 1, W_Y + 1) +
                                 S(k, W_LIN + LIN_A23) * S(k + 1, W_Y + 2);
                    }
                    s3.at(lane) = h;
                });
                c.wave_suffix_sum(s3);
                c.phase([&](int lane) {
                    if (lane < N) S(lane + 1, W_Y + 3, s3.at(lane));
                    red_a.at(lane) = lane < N ? fabs(s0.at(lane)) + fabs(s1.at(lane)) + fabs(s2.at(lane)) + fabs(s3.at(lane)) : 0.0;
                });
                sum_lam = c.wave_sum(red_a);
            }
            c.tick(T_ADJOINT);
            // ============ dual residual of the controls (stage-parallel)
            c.phase([&](int lane) {
                if (lane >= N) {
                    red_a.at(lane) = 0.0;
                    return;
                }
                const int k = lane;
                const double rd_full = RD(), rc = RC();
                const double rdk = (k >= 1) ? rd_full : 0.0;
                const double u0 = S(k, CB + W_U + 0), u1 = S(k, CB + W_U + 1);
                double um0 = 0.0, um1 = 0.0;
                if (k >= 1) {
                    um0 = S(k - 1, CB + W_U + 0);
                    um1 = S(k - 1, CB + W_U + 1);
                }
                double r0 = rc * u0 + rdk * (u0 - um0) - S(k, W_ZUL + 0) + S(k, W_ZUU + 0);
                double r1 = rc * u1 + rdk * (u1 - um1) - S(k, W_ZUL + 1) + S(k, W_ZUU + 1);
                if (k + 1 < N) {
                    r0 -= rd_full * (S(k + 1, CB + W_U + 0) - u0);
                    r1 -= rd_full * (S(k + 1, CB + W_U + 1) - u1);
                }
                const double b01 = S(k, W_LIN + LIN_B01), b11 = S(k, W_LIN + LIN_B11), b21 = S(k, W_LIN + LIN_B21);
                r0 += dt * S(k + 1, W_Y + 3);
                r1 += b01 * S(k + 1, W_Y + 0) + b11 * S(k + 1, W_Y + 1) + b21 * S(k + 1, W_Y + 2);
                red_a.at(lane) = fmax2(fabs(r0), fabs(r1));
            });
            const double err_d = c.wave_max(red_a);
            const double s_d = fmax2(100.0, (sum_lam + sum_z) / (10.0 * N)) / 100.0;
            const double s_c = fmax2(100.0, sum_z / (6.0 * N)) / 100.0;
            for (;;) {
                const double ec = fmax2(cmax - mu, mu - cmin);
                const double E_mu = fmax2(err_d / s_d, ec / s_c);
                if (E_mu <= c.fresh(kKappaEps) * mu && mu > sc(SC_K + K_MUMIN)) {
                    mu = c.uni(fmax2(sc(SC_K + K_MUMIN), fmin2(c.fresh(0.2) * mu, mu * sqrt(mu))));
                    continue;
                }
                break;
            }
            const double E0 = fmax2(err_d / s_d, cmax / s_c);
            kkt_out = E0;
            if (E0 <= P.tol) {
                status_out = (CC && zw_max > c.fresh(1e-6) * SF()) ? 5 : 0;
                break;
            }
            if (P.tol < kAcceptableTol) {
                n_acceptable = (E0 <= c.fresh(kAcceptableTol)) ? n_acceptable + 1 : 0;
                if (n_acceptable >= kAcceptableIter) {
                    status_out = (CC && zw_max > c.fresh(1e-6) * SF()) ? 7 : 6;
                    break;
                }
            }
            if (iter == P.max_iter) break;
            if (P.stall_window > 0) {
                if (E0 < 0.5 * e_mark) {
                    e_mark = c.uni(E0);
                    i_mark = iter;
                } else if (iter - i_mark >= P.stall_window) {
                    status_out = 4;      // the KKT error has not halved within the window: stop burning the budget
                    break;
                }
            }

            c.tick(T_DUALRES);
            // ============ Riccati / DDP factorisation: lane (i, j) = entry of the 8x8 stage block ============
            // Everything of a stage that does not depend on the value function (stage cost Hessian / gradient with
            // the barrier and constraint-curvature terms) is assembled for all stages at once, lane k = stage k, into
            // slots that are free during the sweep (the stage's gain slots and the trial trajectory buffer); the
            // sweep then needs three LDS exchanges per stage: T = P F, H = L + F'T, and (inverse, gains, P) together.
            const int AB = (cur ^ 1) * 6;   // trial buffer: hv0..3 at AB + W_X, hv5 at AB + W_U
            double dV1 = 0.0, delta_w = reg;
            bool ok = false, gn = false;
            int nmod = 0;                      // sweeps of this iteration that failed
            if (nonconvex_cost && prev_needed && iter % kDwProbe != 0) {
                // (xi) start at the ladder's first rung, in the model the last iteration ended in
                delta_w = fmax2(reg, c.fresh(1.0 / 3.0) * c.uni(sc(SC_SPARE + 3)));
                gn = prev_gn && !(mu <= c.fresh(kGnEndMu));
            }
            bool skipped_gn = false;           // (vii) the whole-sweep Gauss-Newton fallback was skipped in favour of the ladder
            set_roles4(AB);
            for (int attempt = 0; attempt < 16 && !ok; ++attempt) {
                ok = true;
                dV1 = 0.0;
                c.phase([&](int lane) {
                    // terminal value function: barrier terms of (theta, v)_N
                    if (lane >= N) return;
                    // ---- stage cost Hessian / gradient of stage k = lane (compact: the 10 distinct entries + 8 gradients)
                    const int k = lane;
                    const double rc = RC(), qtt = sc(SC_K + K_QTT), q33 = sc(SC_K + K_Q33);
                    const double rdk = (k >= 1) ? RD() : 0.0;
                    const double v = S(k, CB + W_X + 3);
                    const double u0 = S(k, CB + W_U + 0), u1 = S(k, CB + W_U + 1);
                    double sig[4], sgr[4];
                    for (int q = 0; q < 4; ++q) {   // 0: theta_k, 1: v_k, 2: a_k, 3: delta_k
                        const bool isx = q < 2;
                        const int b = isx ? q : q - 2;
                        const double val = isx ? S(k, CB + W_X + 2 + b) : (b == 0 ? u0 : u1);
                        const double lo = isx ? xlo(b) : ulo(b), hi = isx ? xhi(b) : uhi(b);
                        const double zl = isx ? S(k, W_ZXL + b) : S(k, W_ZUL + b);
                        const double zu = isx ? S(k, W_ZXU + b) : S(k, W_ZUU + b);
                        const double rl = frcp(val - lo), ru = frcp(hi - val);
                        sig[q] = zl * rl + zu * ru;
                        sgr[q] = mu * (ru - rl);
                    }
                    double wdd = 0.0, wtt = 0.0, wtv = 0.0, wtd = 0.0, wvd = 0.0;
                    if (!gn) {
                        double Sn, Cn, sb, cb_, bp, bpp;
                        dyn_eval(trig(), S(k, CB + W_X + 2), u1, Sn, Cn, sb, cb_);
                        beta_derivs(sb, cb_, bp, bpp);
                        const double yy0 = S(k + 1, W_Y + 0), yy1 = S(k + 1, W_Y + 1), yy2 = S(k + 1, W_Y + 2);
                        const double g = -(yy0 * Cn + yy1 * Sn), h = -(yy0 * Sn - yy1 * Cn);
                        wdd = dt * v * (g * bp * bp + h * bpp) + dt * yy2 * v * kInvWheelbase * (-sb * bp * bp + cb_ * bpp);
                        wtt = dt * v * g;
                        wtv = dt * h;
                        wtd = dt * v * g * bp;
                        wvd = dt * h * bp + dt * yy2 * cb_ * bp * kInvWheelbase;
                    }
                    double l00 = 0.0, l01 = 0.0, l11 = 0.0, h22 = 0.0, h23 = 0.0, h33 = 0.0, hv0 = 0.0, hv1 = 0.0, hv2 = 0.0,
                           hv3 = 0.0;
                    if (k >= 1) {
                        double lx[4];
                        cost_grad(cur, k, lx);
                        const double s = c.ref(k, R_SIN), cc = c.ref(k, R_COS);
                        l00 = SF() * 10.0 * (8.0 * s * s + 4.0 * cc * cc) + delta_w;
                        l01 = SF() * 10.0 * (-8.0 * s * cc + 4.0 * cc * s);
                        l11 = SF() * 10.0 * (8.0 * cc * cc + 4.0 * s * s) + delta_w;
                        if (CC) {
                            const int QS = gn ? W_QG : W_Q;
                            l00 += S(k, QS + 0);
                            l01 += S(k, QS + 1);
                            l11 += S(k, QS + 2);
                        }
                        h22 = wtt + qtt + sig[0] + delta_w;
                        h23 = wtv;
                        h33 = q33 + sig[1] + delta_w;
                        hv0 = lx[0];
                        hv1 = lx[1];
                        if (CC && any_wall) {
                            const double wjv = S(k, W_WJ);
                            if (wjv >= 0.0) {
                                // wall constraint g >= 0 of the node: (z/g) grad g grad g' - z d2g (d2g = 2 I, dropped by
                                // the Gauss-Newton model), barrier gradient -mu/g grad g
                                double nx, ny;
                                const double rg = frcp(wall_slack(k, S(k, CB + W_X + 0), S(k, CB + W_X + 1), (int)wjv, nx, ny));
                                const double zw = S(k, W_ZW), sg = zw * rg, cv = gn ? 0.0 : 2.0 * zw;
                                l00 += sg * nx * nx - cv;
                                l01 += sg * nx * ny;
                                l11 += sg * ny * ny - cv;
                                hv0 -= mu * rg * nx;
                                hv1 -= mu * rg * ny;
                            }
                        }
                        hv2 = lx[2] + sgr[0];
                        hv3 = lx[3] + sgr[1];
                    } else {
                        wtd = 0.0;
                        wvd = 0.0;
                    }
                    double um0 = 0.0, um1 = 0.0;
                    if (k >= 1) {
                        um0 = S(k - 1, CB + W_U + 0);
                        um1 = S(k - 1, CB + W_U + 1);
                    }
                    S(k, A_L00, l00);
                    S(k, A_L01, l01);
                    S(k, A_L11, l11);
                    S(k, A_H22, h22);
                    S(k, A_H23, h23);
                    S(k, A_H33, h33);
                    S(k, A_WTD, wtd);
                    S(k, A_WVD, wvd);
                    S(k, A_H66, rc + rdk + sig[2] + delta_w);
                    S(k, A_H77, rc + rdk + sig[3] + delta_w + wdd);
                    S(k, AB + A_HV0, hv0);
                    S(k, AB + A_HV0 + 1, hv1);
                    S(k, AB + A_HV0 + 2, hv2);
                    S(k, AB + A_HV0 + 3, hv3);
                    S(k, A_HV4, -rdk * (u0 - um0));
                    S(k, AB + A_HV5, -rdk * (u1 - um1));
                    S(k, A_HV6, rc * u0 + rdk * (u0 - um0) + sgr[2]);
                    S(k, A_HV7, rc * u1 + rdk * (u1 - um1) + sgr[3]);
                });
                ok = sweep4(CB, AB, mu, delta_w, gn, dV1);
                if (!ok) {
                    ++nmod;
                    if (gn_skip > 0) skipped_gn = true;
                    if (!gn && gn_skip == 0 && !(nonconvex_cost && mu <= c.fresh(kGnEndMu))) {
                        gn = true;
                    } else {
                        const double dw_last = c.uni(sc(SC_SPARE + 3));
                        if (dw_last == 0.0) {
                            delta_w = (delta_w < 1e-4) ? c.fresh(1e-4) : c.fresh(100.0) * delta_w;
                        } else {
                            const double third = c.fresh(1.0 / 3.0) * dw_last;
                            delta_w = (delta_w < third) ? third : c.fresh(8.0) * delta_w;
                        }
                    }
                    if (delta_w > 1e40) break;
                }
            }
            if (!ok) {
                status_out = 2;
                break;
            }
            if (delta_w > reg) sc(SC_SPARE + 3, delta_w);   // the ladder was needed: remember where it ended
            prev_needed = delta_w > reg;
            prev_gn = gn;
            if (nmod > 0) {
                if (gn_streak == 0) e_streak = c.uni(E0);
                ++gn_streak;
                if (gn_skip > 0) {
                    --gn_skip;
                } else if (gn_streak % kGnWatch == 0) {
                    if (E0 > 0.5 * e_streak) gn_skip = kGnSkip;
                    e_streak = c.uni(E0);
                }
            } else {
                gn_streak = 0;
                gn_skip = 0;
            }

            c.tick(T_RIC_INIT);
            const double tau = c.uni(fmax2(0.99, 1.0 - mu));
            c.tick(T_LINEAR);
            // ============ wall constraints: slack at the current iterate, which the trials' feasibility test needs
            if (CC && any_wall) {
                c.phase([&](int lane) {
                    const int kn = lane + 1;
                    if (kn >= N) return;
                    const double wjv = S(kn, W_WJ);
                    if (wjv >= 0.0) {
                        double nx, ny;
                        S(kn, W_GW, wall_slack(kn, S(kn, CB + W_X + 0), S(kn, CB + W_X + 1), (int)wjv, nx, ny));
                    }
                });
            }
            // ============ line search on the barrier objective (Armijo, 4 trials, factor 1/4), with the linearised Newton
            //              step computed alongside the rollouts
            const double phi0 = c.uni(Jcur + mu * barcur);
            const int tb = cur ^ 1;
            double Jn = 0.0, barn = 0.0;
            int acc_trial = -1;
            const double keep = c.uni(mu <= c.fresh(kProjEndMu) ? c.fresh(kProjKeepEnd) : c.fresh(kProjKeep));
            const bool accepted = line_search(cur, 0.5 * (1.0 - tau), keep, phi0, dV1, mu, Jn, barn, acc_trial);
            const bool have_cross = CC && P.V > 0 && acc_trial != 0;
            // ============ length of the dual step (stage-parallel, from the linearised step parked by the line search)
            c.phase([&](int lane) {
                if (lane >= N) {
                    red_b.at(lane) = 0.0;
                    red_c.at(lane) = 1.0;
                    return;
                }
                const int k = lane;
                double rdn = 0.0, rdd = 1.0;
                for (int i = 0; i < 4; ++i) {
                    const bool isu = i < 2;
                    const int j = isu ? i : i - 2;
                    const double lo = isu ? ulo(j) : xlo(j), hi = isu ? uhi(j) : xhi(j);
                    const int kk = isu ? k : k + 1;
                    const double val = isu ? S(kk, CB + W_U + j) : S(kk, CB + W_X + 2 + j);
                    const double d = S(kk, W_Y + i);
                    const double zl = isu ? S(kk, W_ZUL + j) : S(kk, W_ZXL + j);
                    const double zu = isu ? S(kk, W_ZUU + j) : S(kk, W_ZXU + j);
                    const double rsl = frcp(val - lo), rsu = frcp(hi - val);
                    const double dzl = (mu - zl * d) * rsl - zl, dzu = (mu + zu * d) * rsu - zu;
                    if (-dzl * rdd > rdn * zl) { rdn = -dzl; rdd = zl; }
                    if (-dzu * rdd > rdn * zu) { rdn = -dzu; rdd = zu; }
                }
                if (CC && any_wall && k + 1 < N) {
                    const double wjv = S(k + 1, W_WJ);
                    if (wjv >= 0.0) {
                        double nx, ny;
                        wall_slack(k + 1, S(k + 1, CB + W_X + 0), S(k + 1, CB + W_X + 1), (int)wjv, nx, ny);
                        const double g = S(k + 1, W_GW);
                        const double d = nx * S(k + 1, W_DXY + 0) + ny * S(k + 1, W_DXY + 1), rg = frcp(g);
                        const double zw = S(k + 1, W_ZW), dzw = (mu - zw * d) * rg - zw;
                        if (-dzw * rdd > rdn * zw) { rdn = -dzw; rdd = zw; }
                        S(k + 1, W_DZW, dzw);
                    }
                }
                red_b.at(lane) = rdn;
                red_c.at(lane) = rdd;
            });
            double a_du;
            {
                double rdn, rdd;
                c.wave_max_ratio(red_b, red_c, rdn, rdd);
                a_du = c.uni((rdn > tau * rdd) ? tau * rdd / rdn : 1.0);
            }
            c.tick(T_RATIOS);
            bool newwall = false;
            double bar_shift = 0.0;
            // Levenberg-Marquardt term kept across iterations: two or more backtracks (or no acceptable step) multiply it
            // by 4 (from 1e-3), a full first trial divides it by 4 (to 0 below 1e-3)
            if (accepted && acc_trial == 0) {
                reg = 0.25 * reg;
                if (reg < 1e-3) reg = 0.0;
            } else if (!accepted || acc_trial >= 2) {
                reg = (reg == 0.0) ? 1e-3 : fmin2(4.0 * reg, kRegMax);
            }
            reg = c.uni(reg);
            // ============ dual update (stage-parallel)
            {
                const int NB = (accepted ? tb : cur) * 6;
                c.phase([&](int lane) {
                    if (lane >= N) return;
                    const int k = lane;
                    for (int i = 0; i < 4; ++i) {
                        const bool isu = i < 2;
                        const int j = isu ? i : i - 2;
                        const double lo = isu ? ulo(j) : xlo(j), hi = isu ? uhi(j) : xhi(j);
                        const int kk = isu ? k : k + 1;
                        const int sv = isu ? (W_U + j) : (W_X + 2 + j);
                        const int szl = isu ? (W_ZUL + j) : (W_ZXL + j), szu = isu ? (W_ZUU + j) : (W_ZXU + j);
                        const double val = S(kk, CB + sv), d = S(kk, W_Y + i);
                        const double zl = S(kk, szl), zu = S(kk, szu);
                        const double dzl = (mu - zl * d) * frcp(val - lo) - zl, dzu = (mu + zu * d) * frcp(hi - val) - zu;
                        const double vn = S(kk, NB + sv);
                        const double ml = mu * frcp(vn - lo), mh = mu * frcp(hi - vn);
                        S(kk, szl, fmax2(fmin2(zl + (dzl > 0.0 ? 1.0 : a_du) * dzl, c.fresh(1e10) * ml), c.fresh(1e-10) * ml));
                        S(kk, szu, fmax2(fmin2(zu + (dzu > 0.0 ? 1.0 : a_du) * dzu, c.fresh(1e10) * mh), c.fresh(1e-10) * mh));
                    }
                });
                // wall constraints: dual step of the active ones; a vehicle that a rejected trial took across d = 1
                // becomes the node's wall (multiplier on the central path) if it is nearer than the present one
                if (CC && (any_wall || have_cross)) {
                    c.phase([&](int lane) {
                        red_a.at(lane) = 0.0;
                        red_b.at(lane) = 0.0;
                        const int kn = lane + 1;
                        if (kn >= N) return;
                        const double wjv = S(kn, W_WJ), cr = have_cross ? S(kn, W_CROSS) : -1.0;
                        const double xn0 = S(kn, NB + W_X + 0), xn1 = S(kn, NB + W_X + 1);
                        double nx, ny, gcur = INFINITY, flag = 0.0;
                        if (wjv >= 0.0) {
                            gcur = wall_slack(kn, xn0, xn1, (int)wjv, nx, ny);
                            const double zw = S(kn, W_ZW), dzw = S(kn, W_DZW), ml = mu * frcp(gcur);
                            S(kn, W_ZW, fmax2(fmin2(zw + (dzw > 0.0 ? 1.0 : a_du) * dzw, c.fresh(1e10) * ml), c.fresh(1e-10) * ml));
                            flag = 1.0;
                        }
                        if (cr >= 0.0 && cr != wjv) {
                            const double gc = wall_slack(kn, xn0, xn1, (int)cr, nx, ny);
                            if (gc > 0.0 && gc < gcur) {
                                S(kn, W_WJ, cr);
                                S(kn, W_ZW, mu * frcp(gc));
                                // the barrier value carried for the current point changes with the constraint set
                                red_b.at(lane) = (wjv >= 0.0 ? flog(gcur) : 0.0) - flog(gc);
                                flag = 3.0;
                            }
                        }
                        red_a.at(lane) = flag;
                    });
                    const double wf = c.wave_max(red_a);
                    any_wall = wf > 0.0;
                    newwall = wf > 2.0;
                    if (newwall) bar_shift = c.wave_sum(red_b);
                }
            }
            if (accepted) {
                cur = tb;
                Jcur = c.uni(Jn);
                barcur = c.uni(barn);
                nfail = 0;
            } else if (newwall) {
                nfail = 0;
                if (skipped_gn) gn_skip = 0;
            } else if (skipped_gn) {
                gn_skip = 0;      // (vii) no acceptable step with the regularised exact Hessian: Gauss-Newton first again, no strike
            } else if (++nfail >= 3) {
                status_out = 4;
                ++iter;
                break;
            }
            if (newwall) barcur = c.uni(barcur + bar_shift);
        }
        iters_out = iter;
        cur_out = cur;
    }
};

}  // namespace wave
}  // namespace mpc
