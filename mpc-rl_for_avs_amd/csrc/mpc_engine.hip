// mpc_engine.hip - gfx950 kernels and the C ABI (include/mpc_mi355x.h) of the batched MPC solve engine.
//
// Kernels (all: workgroup = one wave64, no inter-workgroup communication, HBM touched only for inputs / outputs):
//   mpc_solve_wave_kernel   one wave per MPC instance (mpc_wave.hpp + mpc_wave_dev.hpp): stage-parallel phases,
//                           Riccati stage on the FP64 matrix cores, all line-search step lengths at once;
//                           per-instance state (46-54 doubles per stage + trial areas, 14 KB at N = 20) in LDS
//   mpc_preamble_kernel     observation -> problem data (mpc_preamble_wave.hpp), one wave per environment
//   mpc_env_reset_kernel    episode boundaries of the per-environment detector state
// Host side: argument checks, staging of host-pointer calls through one device buffer on the caller's stream,
// per-handle device state (reference table, environment records, preamble outputs).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mpc_mi355x.h"
#include "mpc_core.hpp"
#include "mpc_ltv.hpp"
#include "mpc_wave.hpp"
#include "mpc_wave_dev.hpp"
#include "mpc_preamble.hpp"
#include "mpc_preamble_wave.hpp"
#include "mpc_synth_env.hpp"
#include "mpc_rollout_glue.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

constexpr int kBlock = 64;  // one wave64 per workgroup
constexpr int kMaxDevices = 64;
// Waves per SIMD the wave-cooperative kernel is compiled for.  Since round 5 an instance needs 13.5 KB of LDS at N = 20 with
// the collision cost and 8 vehicles (the linearisation table and the PQ / PB table of the row-cooperative rollout,
// mpc_wave.hpp), i.e. 12 instances per CU: the throughput build is compiled for 3 waves per SIMD (166 registers, no scratch),
// which is what LDS admits anyway.  (Rounds 2 - 4: 9.9 KB, 128 registers, 4 per SIMD.)
constexpr int kWaveOcc = 3, kWaveOccGeneric = 3;
// ... except the runtime-horizon build WITH the collision cost: at 3 waves per SIMD (168 registers) it spilled two vector
// registers to scratch memory (12 B per lane, round 5); compiled for 2 waves per SIMD it has none (round 6)
constexpr int kWaveOccGenericCC = 2;
// The build for batches that do not keep the SIMDs deep in work (WaveOpsT<RELAX>, mpc_wave_dev.hpp): occupancy 2, up to 256
// registers, fresh() / opaque() the identity (everything hoisted): less time per iteration for a wave that has its SIMD to
// itself.  Used up to FOUR waves per SIMD of batch depth (B <= 4096 on 256 CUs: every BASELINE configuration): two are
// resident, the rest of a 4096 batch is dispatched as slots free, which also balances the SIMDs better than static residents,
// and the stragglers a batch ends with run at the lone-wave rate.
constexpr int kWaveOccLat = 2, kLatDepth = 4;
constexpr int kRelaxLat = 7;

// ---------------------------------------------------------------------------------------------------
// wave-cooperative kernel: ONE wave64 per instance (mpc_wave.hpp); workgroup = 1 wave, grid = B
// ---------------------------------------------------------------------------------------------------
template <int NC, int RELAX = 0>
struct WaveCtx : mpc::wave::WaveOpsT<RELAX> {
    static constexpr int kN = NC;
    const double *table;  // [M][REF_COLS] in global memory (wave-uniform index in the serial parts -> scalar loads)
    int e0, M;
    __device__ __forceinline__ WaveCtx(mpc::wave::lds_double_t *l, const double *t, int e, int m)
        : mpc::wave::WaveOpsT<RELAX>{l}, table(t), e0(e), M(m) {}
    __device__ __forceinline__ void tick(int) const {}  // section timing hook, used by tools/ubench only
    __device__ __forceinline__ double ref(int k, int c) const {
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        return table[idx * mpc::REF_COLS + c];
    }
    __device__ __forceinline__ double refv(int k) const {   // speed column, stored behind the [M][REF_COLS] block
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        return table[M * mpc::REF_COLS + idx];
    }
};

// inputs of instance b -> LDS: reference speeds (lane k = stage k) and other vehicles (lane j = vehicle j, advanced per stage by
// speed x dt along the heading: agents/archive/pure_mpc.py:190-191)
template <bool CC, class CTX>
__device__ __forceinline__ void stage_problem(CTX &ctx, const mpc::SolveParams &P, int b, int lane, int N, int SL,
                                              const double *__restrict__ ref5, int M, const double *__restrict__ vref,
                                              const double *__restrict__ others, int Vin) {
    const int OTH = SL * (N + 1) + mpc::wave::SC_SIZE;
    for (int node = lane; node <= N; node += kBlock) {
        double rv;
        if (vref) {
            rv = vref[(size_t)b * (N + 1) + node];
        } else {
            int idx = ctx.e0 + node;
            idx = idx > M - 1 ? M - 1 : idx;
            idx = idx < 0 ? 0 : idx;
            rv = ref5[M * mpc::REF_COLS + idx];
        }
        ctx.st(node * SL + mpc::wave::W_RV, rv);
    }
    if (CC && lane < P.V) {
        const double *ov = others + ((size_t)b * Vin + lane) * 4;
        const double sp = ov[2] * P.dt, hh = ov[3];
        ctx.st(OTH + lane * 4 + 0, ov[0]);
        ctx.st(OTH + lane * 4 + 1, ov[1]);
        ctx.st(OTH + lane * 4 + 2, sp * cos(hh));
        ctx.st(OTH + lane * 4 + 3, sp * sin(hh));
    }
}

template <bool CC, int NC, int OCC, int RELAX>
__global__ __launch_bounds__(kBlock, OCC) void mpc_solve_wave_kernel(
    mpc::SolveParams P, int B, const double *__restrict__ ref5, int M, const double *__restrict__ state,
    const int32_t *__restrict__ ego_index, const double *__restrict__ vref, const double *__restrict__ weights,
    const uint8_t *__restrict__ is_collide, const double *__restrict__ others, int Vin,
    const int32_t *__restrict__ nveh, double w_collision, const double *u_init, int u_shift, uint8_t *u_valid,
    double *__restrict__ u0_out, double *U_out, double *__restrict__ X_out, int32_t *__restrict__ status_out,
    int32_t *__restrict__ iters_out, const int32_t *__restrict__ order) {
    extern __shared__ double smem[];
    const int N = NC > 0 ? NC : P.N;
    if ((int)blockIdx.x >= B) return;            // grid = B workgroups of one wave
    // which instance this workgroup solves: workgroups start in the order of their index, and a batch that does not fit
    // the GPU at once ends with whatever started last (mpc_order_kernel below)
    const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
    if (nveh) P.V = min(P.V, max(0, nveh[b]));   // vehicles actually present in this instance
    const int lane = threadIdx.x;
    constexpr int SL = mpc::wave::stage_slots(CC);
    WaveCtx<NC, RELAX> ctx((mpc::wave::lds_double_t *)smem, ref5, ego_index[b], M);
    stage_problem<CC>(ctx, P, b, lane, N, SL, ref5, M, vref, others, Vin);
    __syncthreads();
    double x0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x0[i] = state[(size_t)b * 4 + i];
    const bool collide = is_collide[b] != 0;
    const double ws_ = collide ? 100.0 : weights[(size_t)b * 3 + 0];  // agents/pure_mpc.py:143-147
    const double wcoll = (CC && collide) ? 3000.0 * w_collision : 0.0;
    mpc::wave::Solver<CC, WaveCtx<NC, RELAX>> solver(P, ctx, x0, ws_, weights[(size_t)b * 3 + 1], weights[(size_t)b * 3 + 2],
                                             wcoll);
    int status, iters, cur;
    double kkt;
    // opt-in warm start: initial controls u_init[b][min(k + u_shift, N-1)] (u_shift = 1: the previous solution of this
    // environment advanced by one stage), where u_valid (if given) says that there is one.  u_init may alias U_out: it
    // is read here, before the solve, and written after it by the same wave.
    const bool warm = u_init != nullptr && (u_valid == nullptr || u_valid[b] != 0);
    if (warm && lane < N) {
        const int src = min(lane + u_shift, N - 1);
        ctx.st(lane * SL + mpc::wave::W_U + 0, mpc::warm_clamp(u_init[((size_t)b * N + src) * 2 + 0], 0));
        ctx.st(lane * SL + mpc::wave::W_U + 1, mpc::warm_clamp(u_init[((size_t)b * N + src) * 2 + 1], 1));
    }
    __syncthreads();
    solver.solve(status, iters, cur, kkt, warm);
    __syncthreads();
    const int CB = cur * 6;
    // only a converged solve may seed the next warm start
    if (P.strict_kink && (status == MPC_STATUS_CONVERGED_ON_KINK || status == MPC_STATUS_ACCEPTABLE_ON_KINK))
        status = MPC_STATUS_KINK_UNSOLVED;      // MPC_FLAG_STRICT_DISCONTINUITY: same iterate, reported as the reference's IPOPT would
    if (lane == 0 && u_valid && U_out) u_valid[b] = MPC_STATUS_IS_SOLVED(status) ? 1 : 0;
    if (lane < 2) u0_out[(size_t)b * 2 + lane] = ctx.ld(CB + mpc::wave::W_U + lane);
    if (U_out && lane < N) {
        U_out[((size_t)b * N + lane) * 2 + 0] = ctx.ld(lane * SL + CB + mpc::wave::W_U + 0);
        U_out[((size_t)b * N + lane) * 2 + 1] = ctx.ld(lane * SL + CB + mpc::wave::W_U + 1);
    }
    if (X_out)
        for (int node = lane; node <= N; node += kBlock)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                X_out[((size_t)b * (N + 1) + node) * 4 + i] = ctx.ld(node * SL + CB + mpc::wave::W_X + i);
    if (lane == 0) {
        if (status_out) status_out[b] = status;
        if (iters_out) iters_out[b] = iters;
    }
}

// one wave that ends `ticks` of the constant 100 MHz clock after it started (mpc_streams_overlap): the clock advances whatever
// the wave does, so the loop ends
__global__ __launch_bounds__(64) void mpc_timer_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Launch order of a batch.  The hardware starts workgroups in index order, a SIMD's issue arbiter favours its oldest wave,
// and a batch lasts as long as its slowest instance - so the instances likely to be slow should be the first to start, and
// there should be no more of them than the GPU holds at once (2048 waves of the latency build).  The cheap predictors this NLP
// offers (oracle iteration counts of eight draws of 4096, profiles/r04_launch_order.txt): instances WITH a predicted collision
// are easy (w_s = 100, agents/pure_mpc.py:143-147, makes the speed term dominate: 99th percentile 26 - 29 iterations against
// 45 - 50 without; of the instances with >= 50 iterations 90 - 100 % have is_collide = 0) unless the ego stands still (speed on
// its bound 0: every exception found); and with the collision cost on, an instance without a vehicle within 15 m of the ego is
// the easy live objective in all but name (no instance with >= 70 iterations had its nearest vehicle farther away, unless it
// stood still).  Tiers, each in its original order (a stable partition): 0 = standing ego, or no predicted collision
// and (collision cost off or a vehicle within kNear); 1 = the other instances without a predicted collision; 2 = a predicted
// collision while the route turns; 3 = the rest.  (Tier 3, round 6: with a predicted collision the instances that still need
// >= 60 iterations - 3 to 6 of 65 536, one or two of them at the cap - sit where the reference heading changes between five rows
// behind the ego and the end of the look-ahead, 15 of 16 over four draws; started last, one of them alone is 13 % of a bulk
// launch.  The straight-route half of tier 2, 28 % of a batch with a maximum of 41 - 58 iterations, now ends the launch:
// 65 536 instances 21.2 -> 19.6 ms in the occupancy model that reproduced the three-tier measurements to 0.2 ms.)
// (Round 5 also built "what the same environment needed one step ago" as the first key for the closed loop; measured neutral to
// negative - profiles/r05_launch_order_prev.txt, docs/NOT_ADOPTED.md - and removed in round 6 together with its per-environment
// record and the environment variable that switched it on.)
constexpr double kOrderStanding = 0.1, kOrderNear = 15.0, kOrderTurn = 0.05;
constexpr int kOrderBehind = 5;
__device__ __forceinline__ int order_tier(const uint8_t *is_collide, const double *state, const double *others, int V,
                                          const int32_t *nveh, const int32_t *ego, const double *ref5, int M, int N, int i) {
    const double *x = state + (size_t)i * 4;
    if (x[3] < kOrderStanding) return 0;
    if (is_collide[i] != 0) {
        const int e0 = ego[i];
        const int a = min(M - 1, max(0, e0 - kOrderBehind)), b = min(M - 1, max(0, e0 + N));
        const double d = ref5[(size_t)b * mpc::REF_COLS + mpc::R_H] - ref5[(size_t)a * mpc::REF_COLS + mpc::R_H];
        return fabs(remainder(d, 6.283185307179586)) > kOrderTurn ? 2 : 3;      // (NaN compares false: tier 3)
    }
    if (!others || V <= 0) return 0;
    const int nv = nveh ? min(V, max(0, nveh[i])) : V;
    bool near = false;
    for (int j = 0; j < nv; ++j) {
        const double *o = others + ((size_t)i * V + j) * 4;
        const double dx = o[0] - x[0], dy = o[1] - x[1];
        near = near || dx * dx + dy * dy < kOrderNear * kOrderNear;
    }
    return near ? 0 : 1;
}
__global__ __launch_bounds__(1024) void mpc_order_kernel(int B, const uint8_t *__restrict__ is_collide,
                                                         const double *__restrict__ state, const double *__restrict__ others,
                                                         int V, const int32_t *__restrict__ nveh, const int32_t *__restrict__ ego,
                                                         const double *__restrict__ ref5, int M, int N,
                                                         int32_t *__restrict__ order, int32_t *__restrict__ tier) {
    __shared__ int s_c0[1024], s_c1[1024], s_c2[1024];
    const int t = threadIdx.x, chunk = (B + 1023) / 1024, lo = min(B, t * chunk), hi = min(B, lo + chunk);
    // tiers first, every thread an interleaved share (neighbouring threads read neighbouring instances), kept in scratch
    for (int i = t; i < B; i += 1024) tier[i] = order_tier(is_collide, state, others, V, nveh, ego, ref5, M, N, i);
    __syncthreads();
    int n0 = 0, n1 = 0, n2 = 0;
    for (int i = lo; i < hi; ++i) {
        const int k = tier[i];
        n0 += k == 0;
        n1 += k == 1;
        n2 += k == 2;
    }
    s_c0[t] = n0;
    s_c1[t] = n1;
    s_c2[t] = n2;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {          // inclusive scans of the tier counts
        const int v0 = t >= d ? s_c0[t - d] : 0, v1 = t >= d ? s_c1[t - d] : 0, v2 = t >= d ? s_c2[t - d] : 0;
        __syncthreads();
        s_c0[t] += v0;
        s_c1[t] += v1;
        s_c2[t] += v2;
        __syncthreads();
    }
    const int tot0 = s_c0[1023], tot1 = s_c1[1023], tot2 = s_c2[1023];
    const int b0 = s_c0[t] - n0, b1 = s_c1[t] - n1, b2 = s_c2[t] - n2;     // members of tier 0 / 1 / 2 before this chunk
    int p0 = b0, p1 = tot0 + b1, p2 = tot0 + tot1 + b2, p3 = tot0 + tot1 + tot2 + (lo - b0 - b1 - b2);
    for (int i = lo; i < hi; ++i) {
        const int k = tier[i];
        if (k == 0) order[p0++] = i;
        else if (k == 1) order[p1++] = i;
        else if (k == 2) order[p2++] = i;
        else order[p3++] = i;
    }
}

// diagnostics (mpc_eval_nlp): the NLP's objective and model step at given points, through Solver::evaluate() of the build
// for runtime horizons (the same class template as every solve kernel); one wave per point
template <bool CC>
__global__ __launch_bounds__(kBlock) void mpc_eval_kernel(mpc::SolveParams P, int B, const double *__restrict__ ref5, int M,
                                                          const int32_t *__restrict__ ego_index, const double *__restrict__ vref,
                                                          const double *__restrict__ weights, const uint8_t *__restrict__ is_collide,
                                                          const double *__restrict__ others, int Vin, double w_collision,
                                                          const double *__restrict__ X, const double *__restrict__ U,
                                                          double *__restrict__ f_out, double *__restrict__ xnext_out) {
    extern __shared__ double smem[];
    const int N = P.N, b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    constexpr int SL = mpc::wave::stage_slots(CC);
    WaveCtx<0, 0> ctx((mpc::wave::lds_double_t *)smem, ref5, ego_index[b], M);
    stage_problem<CC>(ctx, P, b, lane, N, SL, ref5, M, vref, others, Vin);
    for (int node = lane; node <= N; node += kBlock) {
        for (int i = 0; i < 4; ++i) ctx.st(node * SL + mpc::wave::W_X + i, X[((size_t)b * (N + 1) + node) * 4 + i]);
        if (node < N)
            for (int i = 0; i < 2; ++i) ctx.st(node * SL + mpc::wave::W_U + i, U[((size_t)b * N + node) * 2 + i]);
    }
    __syncthreads();
    double x0[4];
    for (int i = 0; i < 4; ++i) x0[i] = X[(size_t)b * (N + 1) * 4 + i];
    const bool collide = is_collide[b] != 0;
    const double ws_ = collide ? 100.0 : weights[(size_t)b * 3 + 0];
    const double wcoll = (CC && collide) ? 3000.0 * w_collision : 0.0;
    mpc::wave::Solver<CC, WaveCtx<0, 0>> solver(P, ctx, x0, ws_, weights[(size_t)b * 3 + 1], weights[(size_t)b * 3 + 2], wcoll);
    mpc::wave::PerLane<double> xn[4];
    const double f = solver.evaluate(xn);
    if (lane == 0) f_out[b] = f;
    if (lane < N)
        for (int i = 0; i < 4; ++i) xnext_out[((size_t)b * N + lane) * 4 + i] = xn[i].at(lane);
}

// ---------------------------------------------------------------------------------------------------
// iterative-linear MPC (mpc_ltv.hpp; reference agents/pure_mpc_linear.py): ONE wave64 per instance, grid = B.
// The ego state comes either from `state` ([B][4] x, y, v, yaw) or straight from the observation (`obs`, row 0 parsed
// as agents/base_agent.py:96-102 does).  U holds the stored control profile (oa, od) on entry and, where the QP was
// solved, the new one on exit; elsewhere it is left alone and the action is (0, 0) (pure_mpc_linear.py:193-196).
// ---------------------------------------------------------------------------------------------------
// Two builds, as for the solve kernel: OCC 3 (<= 168 registers, nothing spilled; LDS admits 12 instances per CU at N = 20
// anyway) keeps the role tables and uniform constants out of registers by recomputing them (opaque / fresh); OCC 2 with
// both relaxed lets the compiler hoist them - 5 % less time per solve when the batch leaves the SIMDs that empty.
template <int OCC, int RELAX>
__global__ __launch_bounds__(kBlock, OCC) void mpc_ltv_kernel(
    mpc::ltv::LtvParams P, int B, const double *__restrict__ ref5, int M, const double *__restrict__ state,
    const float *__restrict__ obs, int rows, double *U, double *__restrict__ u0_out, double *__restrict__ X_out,
    int32_t *__restrict__ status_out, int32_t *__restrict__ iters_out, int32_t *__restrict__ target_out) {
    namespace ltv = mpc::ltv;
    extern __shared__ double smem[];
    const int N = P.N;
    const int b = blockIdx.x;
    if (b >= B) return;
    const int lane = threadIdx.x;
    double x0[4];   // x, y, yaw, v
    if (obs) {
        const float *ob = obs + (size_t)b * rows * mpc::pre::kObsCols;
        x0[0] = (double)ob[1];
        x0[1] = (double)ob[2];
        x0[2] = (double)mpc::pre::normalize_angle_f32(ob[5]);
        x0[3] = (double)mpc::pre::speed_f32(ob[3], ob[4]);
    } else {
        x0[0] = state[(size_t)b * 4 + 0];
        x0[1] = state[(size_t)b * 4 + 1];
        x0[2] = state[(size_t)b * 4 + 3];
        x0[3] = state[(size_t)b * 4 + 2];
    }
    WaveCtx<0, RELAX> ctx((mpc::wave::lds_double_t *)smem, ref5, 0, M);
    // nearest reference point, first minimum of the squared distance (calc_nearest_index_in_direction, :38-60);
    // products and sum rounded separately like the Python expression
    {
        double best = INFINITY;
        int bidx = 0;
        for (int i = lane; i < M; i += kBlock) {
            const double dx = ref5[i * mpc::REF_COLS + mpc::R_X] - x0[0], dy = ref5[i * mpc::REF_COLS + mpc::R_Y] - x0[1];
            const double d = mpc::pre::f64add(mpc::pre::f64mul(dx, dx), mpc::pre::f64mul(dy, dy));
            if (d < best) {
                best = d;
                bidx = i;
            }
        }
        mpc::wave::PerLane<double> pv, pi;
        pv.v = best;
        const double m = ctx.wave_min(pv);
        pi.v = best == m ? (double)bidx : 1e9;
        ctx.e0 = (int)ctx.wave_min(pi);
    }
    if (lane < N) {
        ctx.st(lane * ltv::L_SLOTS + ltv::L_U + 0, U[((size_t)b * N + lane) * 2 + 0]);
        ctx.st(lane * ltv::L_SLOTS + ltv::L_U + 1, U[((size_t)b * N + lane) * 2 + 1]);
    }
    __syncthreads();
    ltv::Solver<WaveCtx<0, RELAX>> solver(P, ctx, x0);
    // the loop of :189: every pass re-simulates the profile the previous one stored and solves the QP linearised
    // about it; the first pass that fails ends the call with the action (0, 0) and the profile stored so far
    int status = ltv::ST_MAX_ITER, iters = 0;
    bool ok = false;
#pragma unroll 1
    for (int pass = 0; pass < P.passes; ++pass) {
        int it = 0;
        solver.solve(status, it);
        iters += it;
        __syncthreads();
        ok = status == ltv::ST_CONVERGED;
        if (!ok) break;
        const int ul = ctx.opaque(lane);   // the row address is formed here, not carried in registers through the solve
        if (ul < N) {
            U[((size_t)b * N + ul) * 2 + 0] = ctx.ld(ul * ltv::L_SLOTS + ltv::L_U + 0);
            U[((size_t)b * N + ul) * 2 + 1] = ctx.ld(ul * ltv::L_SLOTS + ltv::L_U + 1);
        }
    }
    if (lane < 2) u0_out[(size_t)b * 2 + lane] = ok ? ctx.ld(ltv::L_U + lane) : 0.0;
    if (X_out && status != ltv::ST_INFEASIBLE)
        for (int node = lane; node <= N; node += kBlock) {
            double *o = X_out + ((size_t)b * (N + 1) + node) * 4;
            o[0] = ctx.ld(node * ltv::L_SLOTS + ltv::L_X + 0);
            o[1] = ctx.ld(node * ltv::L_SLOTS + ltv::L_X + 1);
            o[2] = ctx.ld(node * ltv::L_SLOTS + ltv::L_X + 3);
            o[3] = ctx.ld(node * ltv::L_SLOTS + ltv::L_X + 2);
        }
    if (lane == 0) {
        if (status_out) status_out[b] = status;
        if (iters_out) iters_out[b] = iters;
        if (target_out) target_out[b] = ctx.e0;
    }
}

// ---------------------------------------------------------------------------------------------------
// observation -> problem data: ONE wave per environment (mpc_preamble_wave.hpp says which lanes do what).  Workgroup = one
// wave, grid = B, 20 KB of LDS (predicted paths, running arc lengths - later the sorted nodes and middles of collinear stretches -, crossing candidates), nothing in scratch.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 2) void mpc_preamble_kernel(
    int B, const float *__restrict__ obs, int rows, const double *__restrict__ ref5, int M, int N, double dt,
    const double *__restrict__ ref_speed, mpc::pre::EnvState *__restrict__ env, double *__restrict__ state,
    int32_t *__restrict__ ego_index, double *__restrict__ vref, uint8_t *__restrict__ is_collide,
    double *__restrict__ others, int Vslots, int32_t *__restrict__ nveh, int advance, double *__restrict__ dbg_ego,
    int32_t *__restrict__ dbg_len, float *__restrict__ dbg_agents) {
    namespace pre = mpc::pre;
    __shared__ double s_words[pre::preamble_wave_lds_doubles()];
    __shared__ int32_t s_conf[pre::kMaxOthers];
    __shared__ pre::P2 s_cpt[pre::kMaxOthers];
    const int b = blockIdx.x;
    if (b >= B) return;
    WaveCtx<0, 3> ctx((mpc::wave::lds_double_t *)s_words, ref5, 0, M);
    const pre::RefTable R{ref5, M};
    constexpr size_t P = pre::kPredHorizon + 1;
    const pre::PreDiag diag{dbg_len ? dbg_ego + (size_t)b * P * 2 : nullptr, dbg_len ? dbg_len + b : nullptr,
                            dbg_len ? dbg_agents + (size_t)b * Vslots * P * 2 : nullptr, Vslots};
    pre::preamble_env_wave(ctx, obs + (size_t)b * rows * pre::kObsCols, rows, R, N, dt, ref_speed ? ref_speed + b : nullptr,
                           env[b], state + (size_t)b * 4, ego_index[b], vref + (size_t)b * (N + 1), is_collide[b],
                           others + (size_t)b * Vslots * 4, Vslots, nveh[b], advance != 0, s_conf, s_cpt, diag);
}

__global__ void mpc_env_reset_kernel(int n, const int32_t *__restrict__ ids, const uint8_t *__restrict__ mask,
                                     mpc::pre::EnvState *__restrict__ env, uint8_t *__restrict__ warm_valid, int cap,
                                     int warm_only, double *__restrict__ ltv_u, int ltv_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = ids ? ids[i] : i;
    if (b < 0 || b >= cap) return;
    if (mask && !mask[i]) return;
    if (!warm_only) env[b] = mpc::pre::EnvState{};
    warm_valid[b] = 0;
    if (ltv_u)                      // stored profile of the iterative-linear agent: oa = od = None -> zeros
        for (int i = 0; i < ltv_row; ++i) ltv_u[(size_t)b * ltv_row + i] = 0.0;
}

// ---------------------------------------------------------------------------------------------------
// synthetic intersection environment (mpc_synth_env.hpp): step + terminal observation + auto-reset + next observation in ONE
// launch (the torch implementation of the same step is ~100 small kernels)
// ---------------------------------------------------------------------------------------------------
// SIXTEEN lanes per environment (round 5; four environments per wave): lane j of a group owns vehicle j
// (model step, respawn draws, crash test, its row of the observation - its place among the rows by counting the vehicles that
// are nearer, which is the stable insertion sort of env::observe), the 85 route points of the lane-centring term are scanned
// 16 at a time with a min-reduction over the group (lowest index among equal distances, as the serial scan keeps the first),
// what concerns the ego alone is computed by every lane of the group.  Statement by statement the arithmetic of
// env::step_env (the host build of that function stays the reference of the tests); one thread per environment spent 30 us
// per step on 256 environments - a chain of ~2000 dependent operations in 4 of the GPU's 1024 SIMDs.
__device__ inline void synth_observe_rows(int q, int K, double x, double y, double th, double sp, double px, double py,
                                          double ps, double ph, bool active, float *s_obs, float *out, bool live) {
    namespace env = mpc::env;
    for (int i = q; i < env::kRows * env::kCols; i += 16) s_obs[i] = 0.0f;
    const double dx = px - x, dy = py - y;
    const double d = active ? sqrt(dx * dx + dy * dy) : INFINITY;
    int rank = 0;
    for (int k = 0; k < K; ++k) {
        const double dk = __shfl(d, k, 16);
        const int ak = __shfl((int)active, k, 16);
        rank += (ak && (dk < d || (dk == d && k < q))) ? 1 : 0;
    }
    __syncthreads();
    if (q == 15) {
        const double s = sin(th), c = cos(th);
        s_obs[0] = 1.0f;
        s_obs[1] = (float)x;
        s_obs[2] = (float)y;
        s_obs[3] = (float)(sp * c);
        s_obs[4] = (float)(sp * s);
        s_obs[5] = (float)th;
        s_obs[6] = (float)s;
        s_obs[7] = (float)c;
    }
    if (active) {
        float *row = s_obs + (1 + rank) * env::kCols;
        const double sh = sin(ph), ch = cos(ph);
        row[0] = 1.0f;
        row[1] = (float)px;
        row[2] = (float)py;
        row[3] = (float)(ps * ch);
        row[4] = (float)(ps * sh);
        row[5] = (float)ph;
        row[6] = (float)sh;
        row[7] = (float)ch;
    }
    __syncthreads();
    if (live)
        for (int i = q; i < env::kRows * env::kCols; i += 16) out[i] = s_obs[i];
}

__global__ __launch_bounds__(64) void mpc_synth_env_rows_kernel(
    int B, int K, double dt, double spawn_probability, uint64_t seed, int env_offset, const double *__restrict__ ref_xy,
    int M, const double *__restrict__ action, double *__restrict__ ego, double *__restrict__ opos,
    double *__restrict__ ospeed, double *__restrict__ ohead, uint8_t *__restrict__ oactive, int32_t *__restrict__ t,
    int64_t *__restrict__ ctr, float *__restrict__ obs, float *__restrict__ terminal_obs, float *__restrict__ reward,
    uint8_t *__restrict__ done, uint8_t *__restrict__ truncated, uint8_t *__restrict__ crashed,
    uint8_t *__restrict__ arrived, int reset_all) {
    namespace env = mpc::env;
    constexpr int kObs = env::kRows * env::kCols;
    __shared__ double s_ref[2 * 128];      // M <= 128 (the launch falls back to one thread per environment otherwise)
    __shared__ float s_obs[4][kObs];
    for (int i = threadIdx.x; i < 2 * M; i += blockDim.x) s_ref[i] = ref_xy[i];
    __syncthreads();
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int b_ = blockIdx.x * 4 + g;
    const bool live = b_ < B;              // a group past the end computes on the last environment and writes nothing
    const int b = live ? b_ : B - 1;
    const int Ks = K > 0 ? K : 1;
    const bool mine = q < K;
    const int j = mine ? q : 0;
    double *eg = ego + (size_t)b * 4;
    const size_t vo = (size_t)b * Ks + j;
    float *o = obs + (size_t)b * kObs;
    const int64_t c0 = ctr[b];
    const env::Rng r(seed, env_offset + b, c0);
    if (reset_all) {
        double px = 0.0, py = 0.0, ps = 0.0, ph = 0.0;
        if (mine) env::spawn_other(r, env::kSlotReset + 4 * j, 5.0, 60.0, px, py, ps, ph);
        const double ex = 2.0, ey = 45.0 + (-5.0 + 10.0 * r.u01(env::kSlotEgo)), eth = -env::kPiE / 2, esp = 10.0;
        synth_observe_rows(q, K, ex, ey, eth, esp, px, py, ps, ph, mine, s_obs[g], o, live);
        if (live && mine) {
            opos[2 * vo] = px;
            opos[2 * vo + 1] = py;
            ospeed[vo] = ps;
            ohead[vo] = ph;
            oactive[vo] = 1;
        }
        if (live && q == 0) {
            eg[0] = ex; eg[1] = ey; eg[2] = eth; eg[3] = esp;
            t[b] = 0;
            ctr[b] = c0 + 1;
        }
        return;
    }
    // ---- ego: the MPC's own vehicle model with the action limits of the environment
    double a = action[(size_t)b * 2], delta = action[(size_t)b * 2 + 1];
    a = a < -5.0 ? -5.0 : (a > 5.0 ? 5.0 : a);
    delta = delta < -env::kPiE / 4 ? -env::kPiE / 4 : (delta > env::kPiE / 4 ? env::kPiE / 4 : delta);
    const double x = eg[0], y = eg[1], th = eg[2], sp = eg[3];
    const double beta = atan(0.5 * tan(delta));
    double ex = x + sp * cos(th + beta) * dt;
    double ey = y + sp * sin(th + beta) * dt;
    double eth = th + sp / env::kWheelbase * sin(beta) * dt;
    const double nv = sp + a * dt;
    double esp = nv < 0.0 ? 0.0 : (nv > 30.0 ? 30.0 : nv);
    // ---- vehicle j
    double px = 0.0, py = 0.0, ps = 0.0, ph = 0.0;
    bool act = false, hit = false;
    if (mine) {
        px = opos[2 * vo]; py = opos[2 * vo + 1]; ps = ospeed[vo]; ph = ohead[vo];
        act = oactive[vo] != 0;
        px += ps * dt * cos(ph);
        py += ps * dt * sin(ph);
        const double ax = fabs(px), ay = fabs(py);
        const bool gone = (ax > ay ? ax : ay) > 65.0 || !act;
        if (gone) {
            const bool respawn = r.u01(env::kSlotRespawn + 5 * j) < spawn_probability;
            if (respawn) env::spawn_other(r, env::kSlotRespawn + 5 * j + 1, 40.0, 60.0, px, py, ps, ph);
            act = respawn;
        }
        if (act) {
            const double dx = px - ex, dy = py - ey;
            hit = sqrt(dx * dx + dy * dy) < env::kCrashDistance;
        }
    }
    const unsigned long long hits = __ballot(hit);
    const bool crash = ((hits >> (16 * g)) & 0xffffull) != 0;
    // ---- nearest route point: the first of the nearest, as the serial scan
    double lateral = INFINITY;
    int idx = 0;
    for (int i = q; i < M; i += 16) {
        const double dx = s_ref[2 * i] - ex, dy = s_ref[2 * i + 1] - ey;
        const double d = sqrt(dx * dx + dy * dy);
        if (d < lateral) {
            lateral = d;
            idx = i;
        }
    }
    for (int off = 8; off >= 1; off >>= 1) {
        const double od = __shfl_xor(lateral, off, 16);
        const int oi = __shfl_xor(idx, off, 16);
        const bool take = od < lateral || (od == lateral && oi < idx);
        lateral = take ? od : lateral;
        idx = take ? oi : idx;
    }
    const bool on_road = lateral <= env::kLaneHalfWidth;
    const bool arr = idx >= M - 3 && on_road;
    double cen = lateral / env::kLaneHalfWidth;
    cen = 1.0 - (cen > 1.0 ? 1.0 : cen);
    const double rew = env::kRewardCollision * (crash ? 1.0 : 0.0) + env::kRewardHighSpeed * (esp / 10.0) +
                       env::kRewardArrived * (arr ? 1.0 : 0.0) + (on_road ? env::kRewardCenter * cen : env::kRewardOffRoad);
    const int tn = t[b] + 1;
    const bool terminated = crash || arr;
    const bool trunc = tn >= env::kEpisodeSteps && !terminated;
    const bool fin = terminated || trunc;
    synth_observe_rows(q, K, ex, ey, eth, esp, px, py, ps, ph, act, s_obs[g], terminal_obs + (size_t)b * kObs, live);
    if (fin) {      // uniform over the group
        ex = 2.0;
        ey = 45.0 + (-5.0 + 10.0 * r.u01(env::kSlotEgo));
        eth = -env::kPiE / 2;
        esp = 10.0;
        if (mine) env::spawn_other(r, env::kSlotReset + 4 * j, 5.0, 60.0, px, py, ps, ph);
        act = mine;
    }
    // (a group that goes on rebuilds the same observation: the four groups of a wave stay in step for the barriers inside)
    synth_observe_rows(q, K, ex, ey, eth, esp, px, py, ps, ph, act, s_obs[g], o, live);
    if (live && mine) {
        opos[2 * vo] = px;
        opos[2 * vo + 1] = py;
        ospeed[vo] = ps;
        ohead[vo] = ph;
        oactive[vo] = act ? 1 : 0;
    }
    if (live && q == 0) {
        eg[0] = ex; eg[1] = ey; eg[2] = eth; eg[3] = esp;
        t[b] = fin ? 0 : tn;
        ctr[b] = c0 + 1;
        reward[b] = (float)rew;
        done[b] = fin;
        truncated[b] = trunc;
        crashed[b] = crash;
        arrived[b] = arr;
    }
}

// ---------------------------------------------------------------------------------------------------
// rollout glue (mpc_rollout_glue.hpp): the policy forward + sample + MPC inputs, and the buffer row / carry-over /
// counters of a step.  One workgroup per environment.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(mpc::glue::kMaxHidden2) void mpc_policy_act_kernel(
    int B, int A, int H2, const float *__restrict__ obs, mpc::glue::PolicyWeights W, float *noise,
    unsigned long long noise_seed, int env_offset, const long long *__restrict__ noise_step,
    int version_v1, int clip, float *__restrict__ actions, float *__restrict__ values, float *__restrict__ log_probs,
    double *__restrict__ mpc_weights, double *__restrict__ mpc_ref_speed) {
    namespace glue = mpc::glue;
    __shared__ float s_x[glue::kObsDim];
    __shared__ float s_h1[glue::kMaxHidden2], s_h2[glue::kMaxHidden2];
    __shared__ float s_head[glue::kMaxAction + 1];
    const int b = blockIdx.x, j = threadIdx.x;
    if (b >= B) return;
    if (j < glue::kObsDim) s_x[j] = obs[(size_t)b * glue::kObsDim + j];
    // the sample's noise: drawn here (and left in `noise` for the caller to see) or handed in
    if (noise_step && j < A) noise[(size_t)b * A + j] = glue::policy_noise(noise_seed, env_offset + b, *noise_step, j);
    __syncthreads();
    if (j < H2) s_h1[j] = glue::layer1_unit(W, H2, s_x, j);
    __syncthreads();
    if (j < H2) s_h2[j] = glue::layer2_unit(W, H2, s_h1, j);
    __syncthreads();
    if (j <= A) s_head[j] = glue::head_unit(W, H2, A, s_h2, j);
    __syncthreads();
    if (j == 0)
        glue::finish_action(W, A, s_head, noise + (size_t)b * A, version_v1, clip, nullptr, actions + (size_t)b * A, values + b,
                            log_probs + b, mpc_weights ? mpc_weights + (size_t)b * 3 : nullptr,
                            mpc_ref_speed ? mpc_ref_speed + b : nullptr);
}

__global__ __launch_bounds__(128) void mpc_rollout_record_kernel(mpc::glue::RecordArgs R, int T, long long *__restrict__ pos_dev,
                                                                 int32_t *__restrict__ ticket,
                                                                 unsigned long long *__restrict__ counts,
                                                                 long long *__restrict__ step_counter) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (b >= R.B) return;
    const long long pos = *pos_dev;
    // a step past the end of the buffer writes no row (the torch path raises there): refused and counted
    const bool inside = pos >= 0 && pos < (long long)T;
    const int bits = mpc::glue::record_thread(R, pos, b, j, inside);
    if (!inside && b == 0 && j == 0) atomicAdd(counts + 4, 1ull);
    if (bits & 1) atomicAdd(counts + 0, 1ull);
    if (bits & 2) atomicAdd(counts + 1, 1ull);
    if (bits & 4) atomicAdd(counts + 2, 1ull);
    if (bits & 8) atomicAdd(counts + 3, 1ull);
    // the last workgroup to finish advances the buffer position (every workgroup has read it by then)
    __syncthreads();
    if (j == 0) {
        __threadfence();
        if (atomicAdd(ticket, 1) == R.B - 1) {
            *ticket = 0;
            *pos_dev = pos + 1;
            if (step_counter) *step_counter += 1;      // policy steps taken so far: keys the next step's noise
        }
    }
}

// end of a rollout (mpc_rollout_glue.hpp: truncation bootstrap + GAE): one workgroup per environment; thread t computes the
// terms of step t (strided over T), thread 0 runs the reversed recurrence over the LDS copies, all threads store
__global__ __launch_bounds__(256) void mpc_rollout_finish_kernel(mpc::glue::GaeArgs g) {
    extern __shared__ float gae_lds[];           // [T] delta -> advantage, [T] coefficient
    const int b = blockIdx.x;
    float *delta = gae_lds, *coef = gae_lds + g.T;
    for (int t = threadIdx.x; t < g.T; t += blockDim.x) mpc::glue::gae_terms(g, b, t, delta + t, coef + t);
    __syncthreads();
    if (threadIdx.x == 0) {
        float gae = 0.0f;
        for (int t = g.T - 1; t >= 0; --t) {
            gae = mpc::glue::gae_step(delta[t], coef[t], gae);
            delta[t] = gae;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < g.T; t += blockDim.x) mpc::glue::gae_store(g, b, t, delta[t]);
}

}  // namespace

struct mpc_handle {
    mpc_config cfg;
    int device = 0;
    double *d_ref = nullptr;  // [M][REF_COLS] x y heading sin cos, followed by the speed column [M]
    int M = 0;
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
    // staging buffers for host-pointer calls
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    // observation-level path (mpc_predict_batch): per-environment detector state and the problem data the
    // preamble kernel writes for the solve kernel
    mpc::pre::EnvState *d_env = nullptr;
    int32_t *d_ids = nullptr;   // scratch of mpc_reset_env_state
    int ids_cap = 0;
    int env_cap = 0;
    double *d_warm = nullptr;        // [env_cap][N][2] last control sequence per environment (MPC_FLAG_WARM_START)
    uint8_t *d_warm_valid = nullptr; // [env_cap] 1 = d_warm holds a solution of the current episode
    double *d_ltv_u = nullptr;       // [env_cap][N][2] stored profile (oa, od) of the iterative-linear agent per environment
    void *d_pre = nullptr;
    size_t pre_bytes = 0;
    int pre_B = 0, pre_V = 0;   // shape of the last preamble output (for mpc_get_last_inputs)
    double *p_state = nullptr, *p_vref = nullptr, *p_others = nullptr;
    int32_t *p_ego = nullptr, *p_nveh = nullptr;
    uint8_t *p_coll = nullptr;
    // diagnostics (mpc_set_diagnostics): polylines of the last preamble launch
    bool diag = false;
    void *d_diag = nullptr;
    size_t diag_bytes = 0;
    int diag_B = 0, diag_V = 0;
    double *g_ego = nullptr;
    int32_t *g_len = nullptr;
    float *g_agents = nullptr;
    // launch order of a batch (mpc_order_kernel): sized by ensure_order / mpc_reserve_envs
    int32_t *d_order = nullptr;
    int order_cap = 0;
};

namespace {

size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// next 16-byte aligned segment of a packed device allocation: returns its offset and advances `off`
size_t carve(size_t &off, size_t bytes) {
    const size_t o = off;
    off += round_up(bytes, 16);
    return o;
}

template <bool CC, int NC, int OCC, int RELAX>
int launch_wave(const mpc_handle *h, const mpc::SolveParams &P, int B, int V, hipStream_t stream,
                const double *d_state, const int32_t *d_ego, const double *d_vref, const double *d_weights,
                const uint8_t *d_coll, const double *d_others, const int32_t *d_nveh, const double *d_uinit, int u_shift,
                uint8_t *d_uvalid, double *d_u0, double *d_U, double *d_X, int32_t *d_status, int32_t *d_iters,
                const int32_t *d_order) {
    auto kern = mpc_solve_wave_kernel<CC, NC, OCC, RELAX>;
    const size_t lds = (size_t)mpc::wave::lds_doubles(CC, P.N, P.V) * sizeof(double);
    // raised once per (kernel, device): the attribute call is not a stream operation and must stay out of a stream
    // capture (hipGraph) of the launch
    static std::atomic<size_t> lds_set[kMaxDevices];      // zero-initialised; concurrent callers at worst both set it
    if (h->device >= kMaxDevices || lds_set[h->device].load(std::memory_order_acquire) < lds) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        if (h->device < kMaxDevices) {
            size_t cur = lds_set[h->device].load(std::memory_order_relaxed);
            while (cur < lds && !lds_set[h->device].compare_exchange_weak(cur, lds, std::memory_order_release)) {}
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(kBlock), lds, stream, P, B, h->d_ref, h->M, d_state, d_ego,
                       d_vref, d_weights, d_coll, d_others, V, d_nveh, h->cfg.w_collision, d_uinit, u_shift, d_uvalid,
                       d_u0, d_U, d_X, d_status, d_iters, d_order);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

// the buffer of the launch order is allocated ONCE, at mpc_create, for the largest batch that is ever ordered (64 waves per SIMD:
// 512 KB on an MI355X) and never moves: a captured hipGraph keeps its address in two kernels (ADVICE r4 - until round 4 it was
// grown on demand, and a later, larger call on the same handle would have left such a graph replaying against freed memory)
int ensure_order(mpc_handle *, int, hipStream_t) { return MPC_OK; }

// staging buffer for host-pointer calls: grown on demand, reused across calls
int ensure_stage(mpc_handle *h, size_t bytes) {
    if (h->stage_bytes >= bytes) return MPC_OK;
    HIP_TRY(hipDeviceSynchronize());     // host-pointer calls are synchronous, but be safe against work of other streams
    if (h->d_stage) HIP_TRY(hipFree(h->d_stage));
    h->d_stage = nullptr;
    h->stage_bytes = 0;
    HIP_TRY(hipMalloc(&h->d_stage, bytes));
    h->stage_bytes = bytes;
    return MPC_OK;
}

// Launch of the solve kernel for B instances whose data already sits in device memory (shared by mpc_solve_batch and
// mpc_predict_batch).  d_nveh: vehicles present per instance or nullptr (= V for all).
mpc::SolveParams solve_params(const mpc_handle *h, int Vuse) {
    mpc::SolveParams P;
    P.N = h->cfg.horizon;
    P.V = Vuse;
    P.max_iter = h->cfg.max_iter;
    P.dt = h->cfg.dt;
    P.tol = h->cfg.tol;
    P.mu_init = 0.1;
    P.w_distance = h->cfg.w_distance;
    P.stall_window = h->cfg.stall_window;
    P.strict_kink = 0;
    return P;
}

int dispatch_solve(const mpc_handle *h, int B, bool cc, int V, uint32_t flags, hipStream_t stream, const double *d_state,
                   const int32_t *d_ego, const double *d_vref, const double *d_weights, const uint8_t *d_coll,
                   const double *d_others, const int32_t *d_nveh, const double *d_uinit, int u_shift, uint8_t *d_uvalid,
                   double *d_u0, double *d_U, double *d_X, int32_t *d_status, int32_t *d_iters) {
    const int N = h->cfg.horizon;
    const int Vuse = cc ? V : 0;
    mpc::SolveParams P = solve_params(h, Vuse);
    P.strict_kink = (flags & MPC_FLAG_STRICT_DISCONTINUITY) ? 1 : 0;
    const bool throughput = (flags & MPC_FLAG_THROUGHPUT) != 0;

    static_assert(MPC_MAX_HORIZON <= mpc::wave::kMaxHorizon, "lane k = stage k needs the horizon to fit a wave");
    const size_t wlds = (size_t)mpc::wave::lds_doubles(cc, N, Vuse) * sizeof(double);
    if (wlds > h->lds_per_cu)
        return fail(MPC_ERR_INVALID_ARG, "horizon / vehicle count too large for the LDS workspace of one instance");
    int rc;
#define MPC_LAUNCH_W(CCV, NCV, OCCV, RLX)                                                                              \
    rc = launch_wave<CCV, NCV, OCCV, RLX>(h, P, (int)B, (int)V, stream, d_state, d_ego, d_vref, d_weights, d_coll, \
                                          d_others, d_nveh, d_uinit, u_shift, d_uvalid, d_u0, d_U, d_X, d_status, d_iters, d_order)
    // which build: by how deep the batch fills the SIMDs (see kWaveOccLat above)
    const int simds = 4 * h->num_cu;
    // launch order (mpc_order_kernel): pays as soon as waves share a SIMD.  Not with MPC_FLAG_THROUGHPUT: batches in flight on
    // several streams may share this handle, and the order buffer is the handle's.  Until round 5 only up to 8 waves per SIMD
    // ("beyond that the order changed nothing" - measured with round 4's kernel); a bulk launch ends with whatever started
    // last running alone, up to 100 iterations x 28 us = 13 % of a launch of 65 536, and the tiers start 85 - 90 % of the
    // instances with >= 60 iterations in the first 40 % of the launch: 21.2 / 21.3 / 21.4 / 22.2 -> 20.3 / 19.8 / 20.6 / 22.6 ms
    // on four draws (profiles/r06_bulk_order.txt; tools/sim_schedule.py's occupancy model predicts each to 0.2 ms)
    const int32_t *d_order = nullptr;
    if (!throughput && B > simds && h->d_order && h->order_cap >= B) {
        hipLaunchKernelGGL(mpc_order_kernel, dim3(1), dim3(1024), 0, stream, (int)B, d_coll, d_state, cc ? d_others : nullptr,
                           (int)Vuse, d_nveh, d_ego, h->d_ref, h->M, N, h->d_order, h->d_order + h->order_cap);
        HIP_TRY(hipGetLastError());
        d_order = h->d_order;
    }
    const bool lat = !throughput && B <= kLatDepth * simds;
#define MPC_LAUNCH_N(CCV, NCV)                                          \
    if (lat) MPC_LAUNCH_W(CCV, NCV, kWaveOccLat, kRelaxLat);            \
    else MPC_LAUNCH_W(CCV, NCV, kWaveOcc, 0)
    if (cc) {
        if (N == 20) { MPC_LAUNCH_N(true, 20); }       /* BASELINE horizon */
        else if (N == 16) { MPC_LAUNCH_N(true, 16); }  /* reference cfg.yaml default */
        else MPC_LAUNCH_W(true, 0, kWaveOccGenericCC, 0);
    } else {
        if (N == 20) { MPC_LAUNCH_N(false, 20); }
        else if (N == 16) { MPC_LAUNCH_N(false, 16); }
        else MPC_LAUNCH_W(false, 0, kWaveOccGeneric, 0);
    }
#undef MPC_LAUNCH_N
#undef MPC_LAUNCH_W
    return rc;
}

}  // namespace

extern "C" {

int mpc_version(void) { return MPC_ABI_VERSION; }

const char *mpc_last_error(void) { return g_last_error.c_str(); }

void mpc_default_config(mpc_config *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(mpc_config);
    cfg->horizon = 20;
    cfg->dt = 0.1;
    cfg->max_iter = 100;
    cfg->device = 0;
    cfg->tol = 1e-8;
    cfg->w_distance = 10.0;
    cfg->w_collision = 1.0;
    cfg->ltv_passes = 1;
}

int mpc_default_config_sized(mpc_config *cfg, int32_t size) {
    if (!cfg) return fail(MPC_ERR_INVALID_ARG, "mpc_default_config_sized: null argument");
    if (size != (int32_t)sizeof(mpc_config))
        return fail(MPC_ERR_INVALID_ARG, "mpc_default_config_sized: the caller's mpc_config has " + std::to_string(size) +
                                         " bytes, this library's has " + std::to_string(sizeof(mpc_config)));
    mpc_default_config(cfg);
    return MPC_OK;
}

int mpc_create(const mpc_config *cfg, mpc_handle **out) {
    if (!cfg || !out) return fail(MPC_ERR_INVALID_ARG, "mpc_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(mpc_config))
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: mpc_config.struct_size mismatch");
    if (cfg->horizon < 1 || cfg->horizon > MPC_MAX_HORIZON)
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: horizon out of range");
    if (!(cfg->dt > 0.0) || cfg->max_iter < 0 || !(cfg->tol > 0.0))
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: dt, max_iter and tol must be positive");
    if (cfg->ltv_passes < 1 || cfg->ltv_passes > 16 || cfg->stall_window < 0)
        return fail(MPC_ERR_INVALID_ARG, "mpc_create: ltv_passes must be 1..16 and stall_window non-negative");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MPC_ERR_NO_DEVICE,
                    std::string("mpc_create: no HIP device available (the engine has no CPU path): ") +
                        (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (cfg->device < 0 || cfg->device >= count) return fail(MPC_ERR_INVALID_ARG, "mpc_create: bad device ordinal");
    HIP_TRY(hipSetDevice(cfg->device));
    mpc_handle *h = new (std::nothrow) mpc_handle();
    if (!h) return fail(MPC_ERR_HIP, "mpc_create: out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) h->num_cu = prop.multiProcessorCount;
            if (prop.maxSharedMemoryPerMultiProcessor > 0) h->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
        }
    }
    {
        const int cap = 64 * 4 * h->num_cu;       // dispatch_solve orders batches of up to 64 waves per SIMD (65 536 instances: 512 KB)
        if (hipMalloc(reinterpret_cast<void **>(&h->d_order), (size_t)cap * 2 * sizeof(int32_t)) == hipSuccess) h->order_cap = cap;
        else h->d_order = nullptr;                 // (unordered launches are correct, only slower)
    }
    *out = h;
    g_last_error.clear();
    return MPC_OK;
}

void mpc_destroy(mpc_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_ref) (void)hipFree(h->d_ref);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->d_env) (void)hipFree(h->d_env);
    if (h->d_warm) (void)hipFree(h->d_warm);
    if (h->d_warm_valid) (void)hipFree(h->d_warm_valid);
    if (h->d_ltv_u) (void)hipFree(h->d_ltv_u);
    if (h->d_ids) (void)hipFree(h->d_ids);
    if (h->d_pre) (void)hipFree(h->d_pre);
    if (h->d_diag) (void)hipFree(h->d_diag);
    if (h->d_order) (void)hipFree(h->d_order);
    delete h;
}

int mpc_set_reference(mpc_handle *h, const double *ref, int32_t M) {
    if (!h || !ref || M < 1 || M > 4096) return fail(MPC_ERR_INVALID_ARG, "mpc_set_reference: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    std::string buf((size_t)M * (mpc::REF_COLS + 1) * sizeof(double), '\0');
    double *r6 = reinterpret_cast<double *>(&buf[0]);
    for (int i = 0; i < M; ++i) {
        r6[i * mpc::REF_COLS + mpc::R_X] = ref[i * 4 + 0];
        r6[i * mpc::REF_COLS + mpc::R_Y] = ref[i * 4 + 1];
        r6[i * mpc::REF_COLS + mpc::R_H] = ref[i * 4 + 3];
        r6[i * mpc::REF_COLS + mpc::R_SIN] = sin(ref[i * 4 + 3]);
        r6[i * mpc::REF_COLS + mpc::R_COS] = cos(ref[i * 4 + 3]);
        r6[M * mpc::REF_COLS + i] = ref[i * 4 + 2];
    }
    if (h->d_ref) HIP_TRY(hipFree(h->d_ref));
    h->d_ref = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_ref), buf.size()));
    HIP_TRY(hipMemcpy(h->d_ref, r6, buf.size(), hipMemcpyHostToDevice));
    h->M = M;
    return MPC_OK;
}

int64_t mpc_workspace_bytes(const mpc_handle *h, int32_t B, int32_t V) {
    if (!h || B < 0 || V < 0 || V > MPC_MAX_OTHERS) return -1;
    // every build of the solve kernel has the same LDS layout since round 5 (mpc_wave.hpp: lds_doubles)
    const int N = h->cfg.horizon;
    (void)B;
    return (int64_t)mpc::wave::lds_doubles(V > 0, N, V) * (int64_t)sizeof(double);
}

int mpc_solve_batch(mpc_handle *h, int32_t B, const double *state, const int32_t *ego_index, const double *vref,
                    const double *weights, const uint8_t *is_collide, const double *others, int32_t V,
                    uint32_t flags, double *u0, double *U, double *X, int32_t *status, int32_t *iters,
                    void *stream_) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: null handle");
    if (B < 0 || !state || !ego_index || !weights || !is_collide || !u0)
        return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: null required pointer or negative batch");
    if (V < 0 || V > MPC_MAX_OTHERS) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: V out of range");
    const bool cc = (flags & MPC_FLAG_COLLISION_COST) != 0;
    if (cc && V > 0 && !others) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: collision cost needs `others`");
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_solve_batch: call mpc_set_reference first");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int N = h->cfg.horizon;
    const size_t N1 = (size_t)N + 1;

    // ---- device views of the arguments
    const double *d_state = state, *d_vref = vref, *d_weights = weights, *d_others = others;
    const int32_t *d_ego = ego_index;
    const uint8_t *d_coll = is_collide;
    double *d_u0 = u0, *d_U = U, *d_X = X;
    int32_t *d_status = status, *d_iters = iters;
    const bool dev = (flags & MPC_FLAG_DEVICE_PTRS) != 0;
    size_t off_u0 = 0, off_U = 0, off_X = 0, off_st = 0, off_it = 0;
    if (!dev) {
        // pack everything into one staging allocation (8-byte aligned segments)
        size_t off = 0;
        auto seg = [&](size_t bytes) { return carve(off, bytes); };
        const size_t o_state = seg((size_t)B * 4 * 8), o_ego = seg((size_t)B * 4), o_w = seg((size_t)B * 3 * 8);
        const size_t o_c = seg((size_t)B), o_vref = vref ? seg((size_t)B * N1 * 8) : 0;
        const size_t o_oth = (cc && V > 0) ? seg((size_t)B * V * 4 * 8) : 0;
        off_u0 = seg((size_t)B * 2 * 8);
        off_U = U ? seg((size_t)B * N * 2 * 8) : 0;
        off_X = X ? seg((size_t)B * N1 * 4 * 8) : 0;
        off_st = seg((size_t)B * 4);
        off_it = seg((size_t)B * 4);
        if (int rc = ensure_stage(h, off)) return rc;
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(sb + o_state, state, (size_t)B * 4 * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_ego, ego_index, (size_t)B * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_w, weights, (size_t)B * 3 * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_c, is_collide, (size_t)B, hipMemcpyHostToDevice, stream));
        d_state = reinterpret_cast<double *>(sb + o_state);
        d_ego = reinterpret_cast<int32_t *>(sb + o_ego);
        d_weights = reinterpret_cast<double *>(sb + o_w);
        d_coll = reinterpret_cast<uint8_t *>(sb + o_c);
        if (vref) {
            HIP_TRY(hipMemcpyAsync(sb + o_vref, vref, (size_t)B * N1 * 8, hipMemcpyHostToDevice, stream));
            d_vref = reinterpret_cast<double *>(sb + o_vref);
        }
        if (cc && V > 0) {
            HIP_TRY(hipMemcpyAsync(sb + o_oth, others, (size_t)B * V * 4 * 8, hipMemcpyHostToDevice, stream));
            d_others = reinterpret_cast<double *>(sb + o_oth);
        }
        d_u0 = reinterpret_cast<double *>(sb + off_u0);
        d_U = U ? reinterpret_cast<double *>(sb + off_U) : nullptr;
        d_X = X ? reinterpret_cast<double *>(sb + off_X) : nullptr;
        d_status = reinterpret_cast<int32_t *>(sb + off_st);
        d_iters = reinterpret_cast<int32_t *>(sb + off_it);
    }

    const bool warm = (flags & MPC_FLAG_WARM_START) != 0;
    if (warm && !U) return fail(MPC_ERR_INVALID_ARG, "mpc_solve_batch: MPC_FLAG_WARM_START needs U (initial controls in, solution out)");
    if (warm && !dev)
        HIP_TRY(hipMemcpyAsync(d_U, U, (size_t)B * N * 2 * 8, hipMemcpyHostToDevice, stream));
    if (int rc = ensure_order(h, B, stream)) return rc;
    if (int rc = dispatch_solve(h, B, cc, V, flags, stream, d_state, d_ego, d_vref, d_weights, d_coll, d_others, nullptr,
                                warm ? d_U : nullptr, 0, nullptr, d_u0, d_U, d_X, d_status, d_iters))
        return rc;

    if (!dev) {
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(u0, sb + off_u0, (size_t)B * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (U) HIP_TRY(hipMemcpyAsync(U, sb + off_U, (size_t)B * N * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (X) HIP_TRY(hipMemcpyAsync(X, sb + off_X, (size_t)B * N1 * 4 * 8, hipMemcpyDeviceToHost, stream));
        if (status) HIP_TRY(hipMemcpyAsync(status, sb + off_st, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        if (iters) HIP_TRY(hipMemcpyAsync(iters, sb + off_it, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    } else if (!(flags & MPC_FLAG_NO_SYNC)) {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return MPC_OK;
}

// grow the per-environment state array to at least B records (new records start as fresh episodes).  The old buffers
// are released, so this must not happen while earlier work may still use them: not inside a stream capture (a captured
// graph has the addresses baked in - size the handle first with mpc_reserve_envs), and only after the device is idle.
static int ensure_env(mpc_handle *h, int B, hipStream_t stream) {
    if (B <= h->env_cap) return MPC_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(MPC_ERR_INVALID_ARG, "per-environment buffers must grow, which is not possible inside a stream capture: "
                                         "call mpc_reserve_envs before capturing");
    int cap = h->env_cap > 0 ? h->env_cap : 256;
    while (cap < B) cap *= 2;
    const size_t wrow = (size_t)h->cfg.horizon * 2 * sizeof(double);
    constexpr int NB = 4;
    void *nb[NB] = {nullptr, nullptr, nullptr, nullptr};
    const size_t bytes[NB] = {(size_t)cap * sizeof(mpc::pre::EnvState), (size_t)cap * wrow, (size_t)cap, (size_t)cap * wrow};
    void *old[NB] = {h->d_env, h->d_warm, h->d_warm_valid, h->d_ltv_u};
    const size_t old_bytes[NB] = {(size_t)h->env_cap * sizeof(mpc::pre::EnvState), (size_t)h->env_cap * wrow,
                                  (size_t)h->env_cap, (size_t)h->env_cap * wrow};
    hipError_t e = hipDeviceSynchronize();       // in-flight MPC_FLAG_NO_SYNC work on any stream still uses the old buffers
    for (int i = 0; i < NB && e == hipSuccess; ++i) {
        e = hipMalloc(&nb[i], bytes[i]);
        if (e == hipSuccess) e = hipMemset(nb[i], 0, bytes[i]);
        if (e == hipSuccess && old[i]) e = hipMemcpy(nb[i], old[i], old_bytes[i], hipMemcpyDeviceToDevice);
    }
    if (e != hipSuccess) {
        for (int i = 0; i < NB; ++i)
            if (nb[i]) (void)hipFree(nb[i]);
        return fail(MPC_ERR_HIP, std::string("ensure_env: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < NB; ++i)
        if (old[i]) (void)hipFree(old[i]);
    h->d_env = static_cast<mpc::pre::EnvState *>(nb[0]);
    h->d_warm = static_cast<double *>(nb[1]);
    h->d_warm_valid = static_cast<uint8_t *>(nb[2]);
    h->d_ltv_u = static_cast<double *>(nb[3]);
    h->env_cap = cap;
    return MPC_OK;
}

int mpc_reserve_envs(mpc_handle *h, int32_t B) {
    if (!h || B < 0) return fail(MPC_ERR_INVALID_ARG, "mpc_reserve_envs: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if (int rc = ensure_order(h, B, nullptr)) return rc;
    return ensure_env(h, B, nullptr);
}

int mpc_predict_batch(mpc_handle *h, int32_t B, const float *obs, int32_t vehicles_count, const double *weights,
                      const double *ref_speed, uint32_t flags, double *act, int32_t *status, int32_t *iters,
                      void *stream_) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_predict_batch: null handle");
    const bool detect_only = (flags & MPC_FLAG_DETECT_ONLY) != 0, detected = (flags & MPC_FLAG_DETECTED) != 0;
    if (detect_only && detected)
        return fail(MPC_ERR_INVALID_ARG, "mpc_predict_batch: MPC_FLAG_DETECT_ONLY and MPC_FLAG_DETECTED exclude each other");
    if (B < 0 || !obs || (!detect_only && (!weights || !act)))
        return fail(MPC_ERR_INVALID_ARG, "mpc_predict_batch: null required pointer or negative batch");
    if (vehicles_count < 1 || vehicles_count > MPC_MAX_OTHERS + 1)
        return fail(MPC_ERR_INVALID_ARG, "mpc_predict_batch: vehicles_count out of range");
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_predict_batch: call mpc_set_reference first");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int N = h->cfg.horizon, rows = vehicles_count, V = vehicles_count - 1;
    const size_t N1 = (size_t)N + 1;
    const bool cc = (flags & MPC_FLAG_COLLISION_COST) != 0 && V > 0;
    const bool dev = (flags & MPC_FLAG_DEVICE_PTRS) != 0;
    if (int rc = ensure_env(h, B, stream)) return rc;

    // problem-data buffers written by the preamble kernel
    {
        size_t off = 0;
        auto seg = [&](size_t bytes) { return carve(off, bytes); };
        const size_t o_state = seg((size_t)B * 4 * 8), o_vref = seg((size_t)B * N1 * 8);
        const size_t o_oth = seg((size_t)B * (V > 0 ? V : 1) * 4 * 8), o_ego = seg((size_t)B * 4);
        const size_t o_nv = seg((size_t)B * 4), o_c = seg((size_t)B);
        if (h->pre_bytes < off) {
            // the old buffer may still be in use by enqueued (MPC_FLAG_NO_SYNC) work, and a captured graph holds its
            // address: never inside a capture, and only once the device is idle
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return fail(MPC_ERR_INVALID_ARG, "mpc_predict_batch: the problem-data buffer must grow, which is not possible "
                                                 "inside a stream capture: run one step of this size before capturing");
            HIP_TRY(hipDeviceSynchronize());
            if (h->d_pre) HIP_TRY(hipFree(h->d_pre));
            h->d_pre = nullptr;
            h->pre_bytes = 0;
            HIP_TRY(hipMalloc(&h->d_pre, off));
            h->pre_bytes = off;
        }
        char *pb = static_cast<char *>(h->d_pre);
        h->p_state = reinterpret_cast<double *>(pb + o_state);
        h->p_vref = reinterpret_cast<double *>(pb + o_vref);
        h->p_others = reinterpret_cast<double *>(pb + o_oth);
        h->p_ego = reinterpret_cast<int32_t *>(pb + o_ego);
        h->p_nveh = reinterpret_cast<int32_t *>(pb + o_nv);
        h->p_coll = reinterpret_cast<uint8_t *>(pb + o_c);
        h->pre_B = B;
        h->pre_V = V;
    }

    const float *d_obs = obs;
    const double *d_weights = weights, *d_rs = ref_speed;
    double *d_act = act;
    int32_t *d_status = status, *d_iters = iters;
    size_t off_act = 0, off_st = 0, off_it = 0;
    if (!dev) {
        size_t off = 0;
        auto seg = [&](size_t bytes) { return carve(off, bytes); };
        const size_t o_obs = seg((size_t)B * rows * mpc::pre::kObsCols * 4), o_w = seg((size_t)B * 3 * 8);
        const size_t o_rs = ref_speed ? seg((size_t)B * 8) : 0;
        off_act = seg((size_t)B * 2 * 8);
        off_st = seg((size_t)B * 4);
        off_it = seg((size_t)B * 4);
        if (int rc = ensure_stage(h, off)) return rc;
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(sb + o_obs, obs, (size_t)B * rows * mpc::pre::kObsCols * 4, hipMemcpyHostToDevice,
                               stream));
        if (weights) HIP_TRY(hipMemcpyAsync(sb + o_w, weights, (size_t)B * 3 * 8, hipMemcpyHostToDevice, stream));
        d_obs = reinterpret_cast<float *>(sb + o_obs);
        d_weights = reinterpret_cast<double *>(sb + o_w);
        if (ref_speed) {
            HIP_TRY(hipMemcpyAsync(sb + o_rs, ref_speed, (size_t)B * 8, hipMemcpyHostToDevice, stream));
            d_rs = reinterpret_cast<double *>(sb + o_rs);
        }
        d_act = reinterpret_cast<double *>(sb + off_act);
        d_status = reinterpret_cast<int32_t *>(sb + off_st);
        d_iters = reinterpret_cast<int32_t *>(sb + off_it);
    }

    if (h->diag) {
        size_t off = 0;
        const size_t P = mpc::pre::kPredHorizon + 1, Vs = (size_t)(V > 0 ? V : 1);
        const size_t o_ego = carve(off, (size_t)B * P * 2 * 8), o_len = carve(off, (size_t)B * 4);
        const size_t o_ag = carve(off, (size_t)B * Vs * P * 2 * 4);
        if (h->diag_bytes < off) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return fail(MPC_ERR_INVALID_ARG, "mpc_predict_batch: diagnostics buffers cannot grow inside a stream capture");
            HIP_TRY(hipDeviceSynchronize());
            if (h->d_diag) HIP_TRY(hipFree(h->d_diag));
            h->d_diag = nullptr;
            h->diag_bytes = 0;
            HIP_TRY(hipMalloc(&h->d_diag, off));
            h->diag_bytes = off;
        }
        char *db = static_cast<char *>(h->d_diag);
        h->g_ego = reinterpret_cast<double *>(db + o_ego);
        h->g_len = reinterpret_cast<int32_t *>(db + o_len);
        h->g_agents = reinterpret_cast<float *>(db + o_ag);
        h->diag_B = B;
        h->diag_V = (int)Vs;
        HIP_TRY(hipMemsetAsync(h->d_diag, 0, off, stream));
    }
    hipLaunchKernelGGL(mpc_preamble_kernel, dim3((unsigned)B), dim3(kBlock), 0, stream,
                       (int)B, d_obs, rows, h->d_ref, h->M, N, h->cfg.dt, d_rs, h->d_env, h->p_state, h->p_ego,
                       h->p_vref, h->p_coll, h->p_others, V > 0 ? V : 1, h->p_nveh, detected ? 0 : 1,
                       h->diag ? h->g_ego : nullptr, h->diag ? h->g_len : nullptr, h->diag ? h->g_agents : nullptr);
    HIP_TRY(hipGetLastError());
    const bool warm = (flags & MPC_FLAG_WARM_START) != 0;
    if (detect_only) {
        // _check_collision on its own (agents/pure_mpc.py:552-676): the records are advanced, nothing is solved
        if (!dev || !(flags & MPC_FLAG_NO_SYNC)) HIP_TRY(hipStreamSynchronize(stream));
        return MPC_OK;
    }
    if (int rc = ensure_order(h, B, stream)) return rc;
    if (int rc = dispatch_solve(h, B, cc, V, flags, stream, h->p_state, h->p_ego, h->p_vref, d_weights, h->p_coll,
                                h->p_others, h->p_nveh, warm ? h->d_warm : nullptr, 1, warm ? h->d_warm_valid : nullptr,
                                d_act, warm ? h->d_warm : nullptr, nullptr, d_status, d_iters))
        return rc;

    if (!dev) {
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(act, sb + off_act, (size_t)B * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (status) HIP_TRY(hipMemcpyAsync(status, sb + off_st, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        if (iters) HIP_TRY(hipMemcpyAsync(iters, sb + off_it, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    } else if (!(flags & MPC_FLAG_NO_SYNC)) {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return MPC_OK;
}

// ---- iterative-linear MPC (reference agents/pure_mpc_linear.py) ---------------------------------------------
// the two builds of mpc_ltv_kernel: waves per SIMD, and what the latency build may keep in registers (fresh / opaque the
// identity, residuals kept: mpc_ltv.hpp relax_bits).  Measured on one box (tools/gpu_ltv_ab.py, round 4): recomputed residuals
// cost a lone wave 2 %.
constexpr int kLtvOcc = 3, kLtvOccLat = 2, kLtvRelaxLat = 1 | 2 | 8;
// ... used up to FOUR waves per SIMD of batch depth, like the solve kernel's (round 6: with the 4x4 factorisation a lone wave is 15 %
// faster than in round 5 and a batch of 4096 takes 1.50 ms in this build against 1.61 ms in the other, 16 384 the same in both)
constexpr int kLtvLatDepth = 4;
static int launch_ltv(mpc_handle *h, int B, hipStream_t stream, const double *d_state, const float *d_obs, int rows,
                      double *d_U, double *d_u0, double *d_X, int32_t *d_status, int32_t *d_iters, int32_t *d_target) {
    mpc::ltv::LtvParams P;
    P.N = h->cfg.horizon;
    P.max_iter = h->cfg.max_iter;
    P.passes = h->cfg.ltv_passes;
    P.dt = h->cfg.dt;
    const size_t lds = (size_t)mpc::ltv::lds_doubles(P.N) * sizeof(double);
    static std::atomic<size_t> lds_set[kMaxDevices];
    if (h->device >= kMaxDevices || lds_set[h->device].load(std::memory_order_acquire) < lds) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mpc_ltv_kernel<kLtvOcc, 0>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mpc_ltv_kernel<kLtvOccLat, kLtvRelaxLat>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (h->device < kMaxDevices) {
            size_t cur = lds_set[h->device].load(std::memory_order_relaxed);
            while (cur < lds && !lds_set[h->device].compare_exchange_weak(cur, lds, std::memory_order_release)) {}
        }
    }
    // which build: by how deep the batch fills the SIMDs
    if (B <= kLtvLatDepth * 4 * h->num_cu)
        hipLaunchKernelGGL((mpc_ltv_kernel<kLtvOccLat, kLtvRelaxLat>), dim3((unsigned)B), dim3(kBlock), lds, stream, P, B,
                           h->d_ref, h->M, d_state, d_obs, rows, d_U, d_u0, d_X, d_status, d_iters, d_target);
    else
        hipLaunchKernelGGL((mpc_ltv_kernel<kLtvOcc, 0>), dim3((unsigned)B), dim3(kBlock), lds, stream, P, B, h->d_ref, h->M,
                           d_state, d_obs, rows, d_U, d_u0, d_X, d_status, d_iters, d_target);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_ltv_solve_batch(mpc_handle *h, int32_t B, const double *state, uint32_t flags, double *u0, double *U, double *X,
                        int32_t *status, int32_t *iters, int32_t *target_index, void *stream_) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_ltv_solve_batch: null handle");
    if (B < 0 || !state || !u0 || !U)
        return fail(MPC_ERR_INVALID_ARG, "mpc_ltv_solve_batch: null required pointer or negative batch");
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_ltv_solve_batch: call mpc_set_reference first");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int N = h->cfg.horizon;
    const size_t N1 = (size_t)N + 1;
    const bool dev = (flags & MPC_FLAG_DEVICE_PTRS) != 0;
    const double *d_state = state;
    double *d_u0 = u0, *d_U = U, *d_X = X;
    int32_t *d_status = status, *d_iters = iters, *d_target = target_index;
    size_t o_u0 = 0, o_U = 0, o_X = 0, o_st = 0, o_it = 0, o_tg = 0;
    if (!dev) {
        size_t off = 0;
        auto seg = [&](size_t bytes) { return carve(off, bytes); };
        const size_t o_state = seg((size_t)B * 4 * 8);
        o_u0 = seg((size_t)B * 2 * 8);
        o_U = seg((size_t)B * N * 2 * 8);
        o_X = X ? seg((size_t)B * N1 * 4 * 8) : 0;
        o_st = seg((size_t)B * 4);
        o_it = seg((size_t)B * 4);
        o_tg = seg((size_t)B * 4);
        if (int rc = ensure_stage(h, off)) return rc;
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(sb + o_state, state, (size_t)B * 4 * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(sb + o_U, U, (size_t)B * N * 2 * 8, hipMemcpyHostToDevice, stream));
        d_state = reinterpret_cast<double *>(sb + o_state);
        d_u0 = reinterpret_cast<double *>(sb + o_u0);
        d_U = reinterpret_cast<double *>(sb + o_U);
        d_X = X ? reinterpret_cast<double *>(sb + o_X) : nullptr;
        d_status = reinterpret_cast<int32_t *>(sb + o_st);
        d_iters = reinterpret_cast<int32_t *>(sb + o_it);
        d_target = reinterpret_cast<int32_t *>(sb + o_tg);
    }
    if (int rc = launch_ltv(h, B, stream, d_state, nullptr, 0, d_U, d_u0, d_X, d_status, d_iters, d_target)) return rc;
    if (!dev) {
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(u0, sb + o_u0, (size_t)B * 2 * 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(U, sb + o_U, (size_t)B * N * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (X) HIP_TRY(hipMemcpyAsync(X, sb + o_X, (size_t)B * N1 * 4 * 8, hipMemcpyDeviceToHost, stream));
        if (status) HIP_TRY(hipMemcpyAsync(status, sb + o_st, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        if (iters) HIP_TRY(hipMemcpyAsync(iters, sb + o_it, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        if (target_index) HIP_TRY(hipMemcpyAsync(target_index, sb + o_tg, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    } else if (!(flags & MPC_FLAG_NO_SYNC)) {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return MPC_OK;
}

int mpc_ltv_predict_batch(mpc_handle *h, int32_t B, const float *obs, int32_t vehicles_count, uint32_t flags, double *act,
                          int32_t *status, int32_t *iters, void *stream_) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_ltv_predict_batch: null handle");
    if (B < 0 || !obs || !act)
        return fail(MPC_ERR_INVALID_ARG, "mpc_ltv_predict_batch: null required pointer or negative batch");
    if (vehicles_count < 1 || vehicles_count > MPC_MAX_OTHERS + 1)
        return fail(MPC_ERR_INVALID_ARG, "mpc_ltv_predict_batch: vehicles_count out of range");
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_ltv_predict_batch: call mpc_set_reference first");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int rows = vehicles_count;
    const bool dev = (flags & MPC_FLAG_DEVICE_PTRS) != 0;
    if (int rc = ensure_env(h, B, stream)) return rc;
    const float *d_obs = obs;
    double *d_act = act;
    int32_t *d_status = status, *d_iters = iters;
    size_t o_act = 0, o_st = 0, o_it = 0;
    if (!dev) {
        size_t off = 0;
        auto seg = [&](size_t bytes) { return carve(off, bytes); };
        const size_t o_obs = seg((size_t)B * rows * mpc::pre::kObsCols * 4);
        o_act = seg((size_t)B * 2 * 8);
        o_st = seg((size_t)B * 4);
        o_it = seg((size_t)B * 4);
        if (int rc = ensure_stage(h, off)) return rc;
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(sb + o_obs, obs, (size_t)B * rows * mpc::pre::kObsCols * 4, hipMemcpyHostToDevice, stream));
        d_obs = reinterpret_cast<float *>(sb + o_obs);
        d_act = reinterpret_cast<double *>(sb + o_act);
        d_status = reinterpret_cast<int32_t *>(sb + o_st);
        d_iters = reinterpret_cast<int32_t *>(sb + o_it);
    }
    if (int rc = launch_ltv(h, B, stream, nullptr, d_obs, rows, h->d_ltv_u, d_act, nullptr, d_status, d_iters, nullptr))
        return rc;
    if (!dev) {
        char *sb = static_cast<char *>(h->d_stage);
        HIP_TRY(hipMemcpyAsync(act, sb + o_act, (size_t)B * 2 * 8, hipMemcpyDeviceToHost, stream));
        if (status) HIP_TRY(hipMemcpyAsync(status, sb + o_st, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        if (iters) HIP_TRY(hipMemcpyAsync(iters, sb + o_it, (size_t)B * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    } else if (!(flags & MPC_FLAG_NO_SYNC)) {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return MPC_OK;
}

int mpc_reset_env_state(mpc_handle *h, const int32_t *env_ids, int32_t n, void *stream_) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_reset_env_state: null handle");
    if (!h->d_env) return MPC_OK;   // nothing allocated yet: every environment is fresh
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!env_ids || n < 0) {
        HIP_TRY(hipMemsetAsync(h->d_env, 0, (size_t)h->env_cap * sizeof(mpc::pre::EnvState), stream));
        HIP_TRY(hipMemsetAsync(h->d_warm_valid, 0, (size_t)h->env_cap, stream));
        HIP_TRY(hipMemsetAsync(h->d_ltv_u, 0, (size_t)h->env_cap * h->cfg.horizon * 2 * sizeof(double), stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return MPC_OK;
    }
    if (n == 0) return MPC_OK;
    if (n > h->ids_cap) {                       // handle-owned, grow-only (no allocation per call)
        HIP_TRY(hipStreamSynchronize(stream));
        if (h->d_ids) HIP_TRY(hipFree(h->d_ids));
        h->d_ids = nullptr;
        h->ids_cap = 0;
        int cap = 256;
        while (cap < n) cap *= 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_ids), (size_t)cap * 4));
        h->ids_cap = cap;
    }
    int32_t *d_ids = h->d_ids;
    hipError_t e = hipMemcpyAsync(d_ids, env_ids, (size_t)n * 4, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mpc_env_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (int)n, d_ids,
                           (const uint8_t *)nullptr, h->d_env, h->d_warm_valid, h->env_cap, 0, h->d_ltv_u, h->cfg.horizon * 2);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(MPC_ERR_HIP, std::string("mpc_reset_env_state: ") + hipGetErrorString(e));
    return MPC_OK;
}

int mpc_reset_env_mask(mpc_handle *h, int32_t B, const uint8_t *done, uint32_t flags, void *stream_) {
    if (!h || B < 0 || !done) return fail(MPC_ERR_INVALID_ARG, "mpc_reset_env_mask: bad argument");
    if (!h->d_env || B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const uint8_t *d_done = done;
    uint8_t *tmp = nullptr;
    if (!(flags & MPC_FLAG_DEVICE_PTRS)) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)B));
        hipError_t e = hipMemcpyAsync(tmp, done, (size_t)B, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) {
            (void)hipFree(tmp);
            return fail(MPC_ERR_HIP, std::string("mpc_reset_env_mask: ") + hipGetErrorString(e));
        }
        d_done = tmp;
    }
    const int n = B < h->env_cap ? B : h->env_cap;
    hipLaunchKernelGGL(mpc_env_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n,
                       (const int32_t *)nullptr, d_done, h->d_env, h->d_warm_valid, h->env_cap,
                       (flags & MPC_FLAG_WARM_START) ? 1 : 0, (flags & MPC_FLAG_WARM_START) ? nullptr : h->d_ltv_u,
                       h->cfg.horizon * 2);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && (tmp || !(flags & MPC_FLAG_NO_SYNC))) e = hipStreamSynchronize(stream);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return fail(MPC_ERR_HIP, std::string("mpc_reset_env_mask: ") + hipGetErrorString(e));
    return MPC_OK;
}

int mpc_get_env_state(mpc_handle *h, int32_t B, int32_t *is_collide, int32_t *ego_index, int32_t *collision_memory,
                      int32_t *stop_index, int32_t *conflict_index, double *conflict_points) {
    if (!h || B < 0) return fail(MPC_ERR_INVALID_ARG, "mpc_get_env_state: bad argument");
    if (B == 0) return MPC_OK;
    if (!h->d_env || B > h->env_cap) return fail(MPC_ERR_INVALID_ARG, "mpc_get_env_state: no such environments");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<mpc::pre::EnvState> host((size_t)B);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host.data(), h->d_env, (size_t)B * sizeof(mpc::pre::EnvState), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
        const mpc::pre::EnvState &s = host[(size_t)b];
        if (is_collide) is_collide[b] = s.is_collide;
        if (ego_index) ego_index[b] = s.ego_index;
        if (collision_memory) collision_memory[b] = s.collision_memory;
        if (stop_index) stop_index[b] = s.stop_index1 - 1;
        if (conflict_index)
            for (int j = 0; j < MPC_MAX_OTHERS; ++j)
                conflict_index[(size_t)b * MPC_MAX_OTHERS + j] = j < s.n_conflict ? s.conflict[j] : -1;
        if (conflict_points)
            for (int j = 0; j < MPC_MAX_OTHERS; ++j) {
                const bool hit = j < s.n_conflict && s.conflict[j] >= 0;
                conflict_points[((size_t)b * MPC_MAX_OTHERS + j) * 2 + 0] = hit ? s.conflict_pt[j][0] : NAN;
                conflict_points[((size_t)b * MPC_MAX_OTHERS + j) * 2 + 1] = hit ? s.conflict_pt[j][1] : NAN;
            }
    }
    return MPC_OK;
}

int64_t mpc_env_state_bytes(void) { return (int64_t)sizeof(mpc::pre::EnvState); }

int mpc_save_env_state(mpc_handle *h, int32_t B, void *records) {
    if (!h || B < 0 || (!records && B > 0)) return fail(MPC_ERR_INVALID_ARG, "mpc_save_env_state: bad argument");
    if (B == 0) return MPC_OK;
    if (!h->d_env || B > h->env_cap) return fail(MPC_ERR_INVALID_ARG, "mpc_save_env_state: no such environments");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(records, h->d_env, (size_t)B * sizeof(mpc::pre::EnvState), hipMemcpyDeviceToHost));
    return MPC_OK;
}

int mpc_set_env_state(mpc_handle *h, int32_t B, const void *records) {
    if (!h || B < 0 || (!records && B > 0)) return fail(MPC_ERR_INVALID_ARG, "mpc_set_env_state: bad argument");
    if (B == 0) return MPC_OK;
    // the records are opaque to the caller but not trusted: the device code uses the counts as loop bounds and the
    // indices as subscripts of the reference table
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_set_env_state: call mpc_set_reference first (indices are checked against it)");
    {
        const auto *rec = static_cast<const mpc::pre::EnvState *>(records);
        const int M = h->M, K = mpc::pre::kMaxOthers;
        for (int b = 0; b < B; ++b) {
            const mpc::pre::EnvState &s = rec[b];
            bool ok = s.collision_memory >= 0 && s.collision_memory <= 10 && (s.has_memorized == 0 || s.has_memorized == 1) &&
                      s.n_memorized >= 0 && s.n_memorized <= K && s.n_conflict >= 0 && s.n_conflict <= K &&
                      (s.is_collide == 0 || s.is_collide == 1) && s.ego_index >= 0 && s.ego_index < M &&
                      s.stop_index1 >= 0 && s.stop_index1 <= M && s.last_valid_stop1 >= 0 && s.last_valid_stop1 <= M;
            for (int j = 0; ok && j < K; ++j)
                ok = s.conflict[j] >= -1 && s.conflict[j] < M && s.memorized[j] >= -1 && s.memorized[j] < M;
            if (!ok) return fail(MPC_ERR_INVALID_ARG, "mpc_set_env_state: record " + std::to_string(b) + " is not a valid detector state");
        }
    }
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = ensure_env(h, B, nullptr)) return rc;
    HIP_TRY(hipMemcpy(h->d_env, records, (size_t)B * sizeof(mpc::pre::EnvState), hipMemcpyHostToDevice));
    // a restored environment is not the episode whose controls the handle remembers: no warm start, no stored LTV profile
    HIP_TRY(hipMemset(h->d_warm_valid, 0, (size_t)B));
    HIP_TRY(hipMemset(h->d_ltv_u, 0, (size_t)B * h->cfg.horizon * 2 * sizeof(double)));
    return MPC_OK;
}

int mpc_streams_overlap(int32_t device, void *stream_a, void *stream_b, int32_t *overlap) {
    if (!overlap) return fail(MPC_ERR_INVALID_ARG, "mpc_streams_overlap: null output pointer");
    HIP_TRY(hipSetDevice(device));
    hipStream_t sa = static_cast<hipStream_t>(stream_a), sb = static_cast<hipStream_t>(stream_b);
    if (sa == sb) {
        *overlap = 0;
        return MPC_OK;
    }
    const long long ticks = 30000;       // of the 100 MHz constant clock: 0.3 ms, long against a launch, short against anything else
    auto timed = [&](bool both, double &ms) -> int {
        HIP_TRY(hipStreamSynchronize(sa));
        HIP_TRY(hipStreamSynchronize(sb));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(mpc_timer_kernel, dim3(1), dim3(64), 0, sa, ticks);
        if (both) hipLaunchKernelGGL(mpc_timer_kernel, dim3(1), dim3(64), 0, sb, ticks);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(sa));
        HIP_TRY(hipStreamSynchronize(sb));
        ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return MPC_OK;
    };
    double one = 0.0, two = 0.0, warm = 0.0;
    if (int rc = timed(true, warm)) return rc;       // first launch of the kernel on these streams: code load, queue wake-up
    if (int rc = timed(false, one)) return rc;
    if (int rc = timed(true, two)) return rc;
    *overlap = two < 1.5 * one ? 1 : 0;
    return MPC_OK;
}

int mpc_synth_env_step(int32_t device, int32_t B, int32_t K, double dt, double spawn_probability, uint64_t seed,
                       int32_t env_offset, const double *ref_xy, int32_t M, const double *action, double *ego, double *opos,
                       double *ospeed, double *ohead, uint8_t *oactive, int32_t *t, int64_t *rng_counter, float *obs,
                       float *terminal_obs, float *reward, uint8_t *done, uint8_t *truncated, uint8_t *crashed,
                       uint8_t *arrived, int32_t reset_all, void *stream_) {
    if (B < 0 || K < 0 || K > mpc::env::kMaxOthers || !(dt > 0.0) || M < 1)
        return fail(MPC_ERR_INVALID_ARG, "mpc_synth_env_step: bad size");
    if (!ref_xy || !ego || !opos || !ospeed || !ohead || !oactive || !t || !rng_counter || !obs)
        return fail(MPC_ERR_INVALID_ARG, "mpc_synth_env_step: null state pointer");
    if (!reset_all && (!action || !terminal_obs || !reward || !done || !truncated || !crashed || !arrived))
        return fail(MPC_ERR_INVALID_ARG, "mpc_synth_env_step: null output pointer");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(device));
    // sixteen lanes per environment (mpc_synth_env_rows_kernel); the one-thread-per-environment kernel it replaced in round 5 is
    // gone (round 6) - its statement env::step_env stays as the host-side reference of the tests
    if (K > 15 || M > 128)
        return fail(MPC_ERR_INVALID_ARG, "mpc_synth_env_step: at most 15 other vehicles and 128 route points");
    hipLaunchKernelGGL(mpc_synth_env_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream_),
                       (int)B, (int)K, dt, spawn_probability, seed, (int)env_offset, ref_xy, (int)M, action, ego, opos, ospeed,
                       ohead, oactive, t, rng_counter, obs, terminal_obs, reward, done, truncated, crashed, arrived,
                       (int)reset_all);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_policy_act(int32_t device, int32_t B, int32_t A, int32_t H2, const float *obs, const float *w1, const float *b1,
                   const float *w2, const float *b2, const float *wh, const float *bh, const float *std_, const float *c0,
                   float *noise, uint64_t noise_seed, int32_t env_offset, const int64_t *noise_step, int32_t version_v1,
                   int32_t clip, float *actions, float *values, float *log_probs, double *mpc_weights, double *mpc_ref_speed,
                   void *stream_) {
    if (B < 0 || A < 1 || A > mpc::glue::kMaxAction || H2 < 2 || H2 > mpc::glue::kMaxHidden2 || (H2 & 1))
        return fail(MPC_ERR_INVALID_ARG, "mpc_policy_act: bad size (action_dim 1..8, 2 x hidden <= 256)");
    if (!obs || !w1 || !b1 || !w2 || !b2 || !wh || !bh || !std_ || !c0 || !noise || !actions || !values || !log_probs)
        return fail(MPC_ERR_INVALID_ARG, "mpc_policy_act: null pointer");
    if (version_v1 ? (!mpc_weights || A < 3) : !mpc_ref_speed)
        return fail(MPC_ERR_INVALID_ARG, "mpc_policy_act: v1 needs mpc_weights and >= 3 action components, v0 needs mpc_ref_speed");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(device));
    const mpc::glue::PolicyWeights W{w1, b1, w2, b2, wh, bh, std_, c0};
    const int threads = H2 > mpc::glue::kObsDim ? H2 : mpc::glue::kObsDim;
    hipLaunchKernelGGL(mpc_policy_act_kernel, dim3((unsigned)B), dim3((unsigned)((threads + 63) / 64 * 64)), 0,
                       reinterpret_cast<hipStream_t>(stream_), (int)B, (int)A, (int)H2, obs, W, noise, (unsigned long long)noise_seed,
                       (int)env_offset, reinterpret_cast<const long long *>(noise_step), (int)version_v1, (int)clip,
                       actions, values, log_probs, version_v1 ? mpc_weights : nullptr, version_v1 ? nullptr : mpc_ref_speed);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_rollout_record(int32_t device, int32_t T, int32_t B, int32_t A, int32_t cols, int32_t keep_terminal, float *row,
                       double *mpc_actions_buf, int64_t *pos_dev, int32_t *ticket, float *last_obs, float *last_starts,
                       const float *actions, const float *values, const float *log_probs, const double *mpc_act,
                       const int32_t *mpc_status, const float *new_obs, const float *reward, const uint8_t *done,
                       const float *terminal_obs, const uint8_t *truncated, const uint8_t *crashed, const uint8_t *arrived,
                       int64_t *counts, uint8_t *dones_out, int64_t *step_counter, void *stream_) {
    constexpr int O = mpc::glue::kObsDim;
    if (T < 1 || B < 0 || A < 1 || A > mpc::glue::kMaxAction || cols != O + A + 4 + (keep_terminal ? O + 1 : 0))
        return fail(MPC_ERR_INVALID_ARG, "mpc_rollout_record: bad size / row layout");
    if (!row || !mpc_actions_buf || !pos_dev || !ticket || !last_obs || !last_starts || !actions || !values || !log_probs ||
        !mpc_act || !mpc_status || !new_obs || !reward || !done || !crashed || !arrived || !counts || !dones_out ||
        (keep_terminal && (!terminal_obs || !truncated)))
        return fail(MPC_ERR_INVALID_ARG, "mpc_rollout_record: null pointer");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(device));
    const mpc::glue::RecordArgs R{(int)B, (int)A, (int)cols, (int)keep_terminal, row, mpc_actions_buf, last_obs, last_starts, actions,
                                 values, log_probs, mpc_act, mpc_status, new_obs, reward, done, terminal_obs, truncated, crashed,
                                 arrived, dones_out};
    hipLaunchKernelGGL(mpc_rollout_record_kernel, dim3((unsigned)B), dim3(128), 0, reinterpret_cast<hipStream_t>(stream_), R, (int)T,
                       reinterpret_cast<long long *>(pos_dev), ticket, reinterpret_cast<unsigned long long *>(counts),
                       reinterpret_cast<long long *>(step_counter));
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_rollout_finish(int32_t device, int32_t T, int32_t B, int32_t A, int32_t cols, int32_t keep_terminal, float *row,
                       const float *last_values, const uint8_t *dones, const float *terminal_values, double gamma,
                       double gae_lambda, float *advantages, float *returns, void *stream_) {
    constexpr int O = mpc::glue::kObsDim;
    if (T < 1 || T > 8192 || B < 0 || A < 1 || A > mpc::glue::kMaxAction || cols != O + A + 4 + (keep_terminal ? O + 1 : 0))
        return fail(MPC_ERR_INVALID_ARG, "mpc_rollout_finish: bad size / row layout (1 <= T <= 8192)");
    if (!row || !last_values || !dones || !advantages || !returns)
        return fail(MPC_ERR_INVALID_ARG, "mpc_rollout_finish: null pointer");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(device));
    const mpc::glue::GaeArgs g{(int)T, (int)B, (int)A, (int)cols, (int)keep_terminal, row, last_values, dones, terminal_values,
                              (float)gamma, (float)(gamma * gae_lambda), advantages, returns};
    hipLaunchKernelGGL(mpc_rollout_finish_kernel, dim3((unsigned)B), dim3(256), (size_t)T * 2 * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream_), g);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
}

int mpc_eval_nlp(mpc_handle *h, int32_t B, const int32_t *ego_index, const double *vref, const double *weights,
                 const uint8_t *is_collide, const double *others, int32_t V, uint32_t flags, const double *X, const double *U,
                 double *f, double *x_next) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_eval_nlp: null handle");
    if (B < 0 || !ego_index || !weights || !is_collide || !X || !U || !f || !x_next)
        return fail(MPC_ERR_INVALID_ARG, "mpc_eval_nlp: null required pointer or negative batch");
    if (V < 0 || V > MPC_MAX_OTHERS) return fail(MPC_ERR_INVALID_ARG, "mpc_eval_nlp: V out of range");
    const bool cc = (flags & MPC_FLAG_COLLISION_COST) != 0;
    if (flags & ~(uint32_t)MPC_FLAG_COLLISION_COST) return fail(MPC_ERR_INVALID_ARG, "mpc_eval_nlp: only MPC_FLAG_COLLISION_COST (host pointers, synchronous)");
    if (cc && V > 0 && !others) return fail(MPC_ERR_INVALID_ARG, "mpc_eval_nlp: collision cost needs `others`");
    if (!h->d_ref) return fail(MPC_ERR_NO_REFERENCE, "mpc_eval_nlp: call mpc_set_reference first");
    if (B == 0) return MPC_OK;
    HIP_TRY(hipSetDevice(h->device));
    const int N = h->cfg.horizon;
    const size_t N1 = (size_t)N + 1;
    size_t off = 0;
    auto seg = [&](size_t bytes) { return carve(off, bytes); };
    const size_t o_ego = seg((size_t)B * 4), o_w = seg((size_t)B * 3 * 8), o_c = seg((size_t)B);
    const size_t o_vref = vref ? seg((size_t)B * N1 * 8) : 0, o_oth = (cc && V > 0) ? seg((size_t)B * V * 4 * 8) : 0;
    const size_t o_X = seg((size_t)B * N1 * 4 * 8), o_U = seg((size_t)B * N * 2 * 8), o_f = seg((size_t)B * 8);
    const size_t o_xn = seg((size_t)B * N * 4 * 8);
    if (int rc = ensure_stage(h, off)) return rc;
    char *sb = static_cast<char *>(h->d_stage);
    hipStream_t stream = nullptr;
    HIP_TRY(hipMemcpyAsync(sb + o_ego, ego_index, (size_t)B * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(sb + o_w, weights, (size_t)B * 3 * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(sb + o_c, is_collide, (size_t)B, hipMemcpyHostToDevice, stream));
    if (vref) HIP_TRY(hipMemcpyAsync(sb + o_vref, vref, (size_t)B * N1 * 8, hipMemcpyHostToDevice, stream));
    if (cc && V > 0) HIP_TRY(hipMemcpyAsync(sb + o_oth, others, (size_t)B * V * 4 * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(sb + o_X, X, (size_t)B * N1 * 4 * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(sb + o_U, U, (size_t)B * N * 2 * 8, hipMemcpyHostToDevice, stream));
    mpc::SolveParams P = solve_params(h, cc ? V : 0);
    const size_t lds = (size_t)mpc::wave::lds_doubles(cc, N, P.V) * sizeof(double);
    auto D = [&](size_t o) { return reinterpret_cast<double *>(sb + o); };
    if (cc)
        hipLaunchKernelGGL(mpc_eval_kernel<true>, dim3((unsigned)B), dim3(kBlock), lds, stream, P, (int)B, h->d_ref, h->M,
                           reinterpret_cast<int32_t *>(sb + o_ego), vref ? D(o_vref) : nullptr, D(o_w),
                           reinterpret_cast<uint8_t *>(sb + o_c), V > 0 ? D(o_oth) : nullptr, (int)V, h->cfg.w_collision, D(o_X),
                           D(o_U), D(o_f), D(o_xn));
    else
        hipLaunchKernelGGL(mpc_eval_kernel<false>, dim3((unsigned)B), dim3(kBlock), lds, stream, P, (int)B, h->d_ref, h->M,
                           reinterpret_cast<int32_t *>(sb + o_ego), vref ? D(o_vref) : nullptr, D(o_w),
                           reinterpret_cast<uint8_t *>(sb + o_c), nullptr, 0, h->cfg.w_collision, D(o_X), D(o_U), D(o_f), D(o_xn));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(f, sb + o_f, (size_t)B * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(x_next, sb + o_xn, (size_t)B * N * 4 * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return MPC_OK;
}

int mpc_set_diagnostics(mpc_handle *h, int32_t on) {
    if (!h) return fail(MPC_ERR_INVALID_ARG, "mpc_set_diagnostics: null handle");
    h->diag = on != 0;
    if (!h->diag) h->diag_B = 0;
    return MPC_OK;
}

int mpc_get_last_paths(mpc_handle *h, int32_t B, double *ego_path, int32_t *ego_len, float *agent_paths) {
    if (!h || B < 0) return fail(MPC_ERR_INVALID_ARG, "mpc_get_last_paths: bad argument");
    if (B == 0) return MPC_OK;
    if (!h->diag || !h->d_diag || B > h->diag_B)
        return fail(MPC_ERR_INVALID_ARG, "mpc_get_last_paths: no diagnostics of that size (mpc_set_diagnostics, then mpc_predict_batch)");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t P = mpc::pre::kPredHorizon + 1;
    if (ego_path) HIP_TRY(hipMemcpy(ego_path, h->g_ego, (size_t)B * P * 2 * 8, hipMemcpyDeviceToHost));
    if (ego_len) HIP_TRY(hipMemcpy(ego_len, h->g_len, (size_t)B * 4, hipMemcpyDeviceToHost));
    if (agent_paths) HIP_TRY(hipMemcpy(agent_paths, h->g_agents, (size_t)B * h->diag_V * P * 2 * 4, hipMemcpyDeviceToHost));
    return MPC_OK;
}

int mpc_get_last_inputs(mpc_handle *h, int32_t B, double *state, int32_t *ego_index, double *vref,
                        uint8_t *is_collide, double *others, int32_t *nveh) {
    if (!h || B < 0) return fail(MPC_ERR_INVALID_ARG, "mpc_get_last_inputs: bad argument");
    if (B == 0) return MPC_OK;
    if (!h->d_pre || B > h->pre_B) return fail(MPC_ERR_INVALID_ARG, "mpc_get_last_inputs: no preamble output of that size");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t N1 = (size_t)h->cfg.horizon + 1, V = (size_t)(h->pre_V > 0 ? h->pre_V : 1);
    if (state) HIP_TRY(hipMemcpy(state, h->p_state, (size_t)B * 4 * 8, hipMemcpyDeviceToHost));
    if (ego_index) HIP_TRY(hipMemcpy(ego_index, h->p_ego, (size_t)B * 4, hipMemcpyDeviceToHost));
    if (vref) HIP_TRY(hipMemcpy(vref, h->p_vref, (size_t)B * N1 * 8, hipMemcpyDeviceToHost));
    if (is_collide) HIP_TRY(hipMemcpy(is_collide, h->p_coll, (size_t)B, hipMemcpyDeviceToHost));
    if (others) HIP_TRY(hipMemcpy(others, h->p_others, (size_t)B * V * 4 * 8, hipMemcpyDeviceToHost));
    if (nveh) HIP_TRY(hipMemcpy(nveh, h->p_nveh, (size_t)B * 4, hipMemcpyDeviceToHost));
    return MPC_OK;
}

}  // extern "C"
