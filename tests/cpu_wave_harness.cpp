// Host build of mpc-rl_for_avs_amd/csrc/mpc_wave.hpp for tests only (-m "not gpu"): the wave-cooperative solver
// with its 64 lanes emulated by loops (each `phase` runs lane 0..63 in turn), compared with the oracle on the CPU.
// Never loaded by the product.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "host_wave_ctx.hpp"

namespace {
int g_stall_window = 0;
template <bool CC>
void run(const mpc::SolveParams &P, HostCtx &ctx, const double *x0, double ws, double wc, double wd, double wcoll,
         int &st, int &it, int &cur, double &e, bool warm) {
    mpc::wave::Solver<CC, HostCtx> s(P, ctx, x0, ws, wc, wd, wcoll);
    s.solve(st, it, cur, e, warm);
}   // mpc_config.stall_window of the calls that follow (wave_set_stall_window)
}  // namespace

extern "C" void wave_set_stall_window(int w) { g_stall_window = w > 0 ? w : 0; }

extern "C" int wave_solve_batch_warm(int B, int N, double dt, const double *ref_table, int M, const double *state,
                                     const int32_t *ego_index, const double *vref, const double *weights,
                                     const uint8_t *is_collide, const double *others, int V, uint32_t flags,
                                     double w_distance, double w_collision, double tol, int max_iter,
                                     const double *u_init, double *u0, double *U, double *X, int32_t *status,
                                     int32_t *iters, double *kkt);

extern "C" int wave_solve_batch(int B, int N, double dt, const double *ref_table, int M, const double *state,
                                const int32_t *ego_index, const double *vref, const double *weights,
                                const uint8_t *is_collide, const double *others, int V, uint32_t flags,
                                double w_distance, double w_collision, double tol, int max_iter, double *u0,
                                double *U, double *X, int32_t *status, int32_t *iters, double *kkt) {
    return wave_solve_batch_warm(B, N, dt, ref_table, M, state, ego_index, vref, weights, is_collide, others, V, flags,
                                 w_distance, w_collision, tol, max_iter, nullptr, u0, U, X, status, iters, kkt);
}

// u_init: [B][N][2] initial controls (warm start) or nullptr (cold start of the reference)
extern "C" int wave_solve_batch_warm(int B, int N, double dt, const double *ref_table, int M, const double *state,
                                     const int32_t *ego_index, const double *vref, const double *weights,
                                     const uint8_t *is_collide, const double *others, int V, uint32_t flags,
                                     double w_distance, double w_collision, double tol, int max_iter,
                                     const double *u_init, double *u0, double *U, double *X, int32_t *status,
                                     int32_t *iters, double *kkt) {
    if (N > mpc::wave::kMaxHorizon) return -1;
    const int cc = (flags & 1u) ? 1 : 0;
    const int Vuse = cc ? V : 0;
    std::vector<double> table((size_t)M * mpc::REF_COLS);
    for (int i = 0; i < M; ++i) {
        table[i * mpc::REF_COLS + mpc::R_X] = ref_table[i * 4 + 0];
        table[i * mpc::REF_COLS + mpc::R_Y] = ref_table[i * 4 + 1];
        table[i * mpc::REF_COLS + mpc::R_H] = ref_table[i * 4 + 3];
        table[i * mpc::REF_COLS + mpc::R_SIN] = std::sin(ref_table[i * 4 + 3]);
        table[i * mpc::REF_COLS + mpc::R_COS] = std::cos(ref_table[i * 4 + 3]);
    }
    mpc::SolveParams P;
    P.N = N; P.V = Vuse; P.max_iter = max_iter; P.dt = dt; P.tol = tol; P.mu_init = 0.1;
    P.w_distance = w_distance;
    P.stall_window = g_stall_window;
    P.strict_kink = 0;
    const int SL = mpc::wave::stage_slots(cc);
    const int nd = mpc::wave::lds_doubles(cc, N, Vuse);
    for (int b = 0; b < B; ++b) {
        std::vector<double> L((size_t)nd, NAN);
        HostCtx ctx{};
        ctx.L = L.data();
        ctx.table = table.data();
        ctx.e0 = ego_index[b];
        ctx.M = M;
        ctx.nwords = nd;   // the kernel source must stay inside lds_doubles() for every horizon / vehicle count
        for (int k = 0; k <= N; ++k) {
            int idx = ego_index[b] + k;
            idx = idx > M - 1 ? M - 1 : idx;
            idx = idx < 0 ? 0 : idx;
            L[k * SL + mpc::wave::W_RV] = vref ? vref[(size_t)b * (N + 1) + k] : ref_table[idx * 4 + 2];
        }
        const int OTH = SL * (N + 1) + mpc::wave::SC_SIZE;
        for (int j = 0; j < Vuse; ++j) {
            const double *ov = others + ((size_t)b * V + j) * 4;
            L[OTH + j * 4 + 0] = ov[0];
            L[OTH + j * 4 + 1] = ov[1];
            L[OTH + j * 4 + 2] = ov[2] * dt * std::cos(ov[3]);
            L[OTH + j * 4 + 3] = ov[2] * dt * std::sin(ov[3]);
        }
        const bool collide = is_collide[b] != 0;
        const double ws_ = collide ? 100.0 : weights[3 * b + 0];
        const double wcoll = (cc && collide) ? 3000.0 * w_collision : 0.0;
        if (u_init)
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < 2; ++i)
                    L[k * SL + mpc::wave::W_U + i] = mpc::warm_clamp(u_init[((size_t)b * N + k) * 2 + i], i);
        int st, it, cur;
        double e;
        if (cc)
            run<true>(P, ctx, state + 4 * (size_t)b, ws_, weights[3 * b + 1], weights[3 * b + 2], wcoll, st, it, cur, e,
                      u_init != nullptr);
        else
            run<false>(P, ctx, state + 4 * (size_t)b, ws_, weights[3 * b + 1], weights[3 * b + 2], wcoll, st, it, cur, e,
                       u_init != nullptr);
        const int CB = cur * 6;
        u0[2 * b + 0] = L[0 * SL + CB + mpc::wave::W_U + 0];
        u0[2 * b + 1] = L[0 * SL + CB + mpc::wave::W_U + 1];
        if (U)
            for (int k = 0; k < N; ++k)
                for (int i = 0; i < 2; ++i) U[((size_t)b * N + k) * 2 + i] = L[k * SL + CB + mpc::wave::W_U + i];
        if (X)
            for (int k = 0; k <= N; ++k)
                for (int i = 0; i < 4; ++i) X[((size_t)b * (N + 1) + k) * 4 + i] = L[k * SL + CB + mpc::wave::W_X + i];
        status[b] = st;
        iters[b] = it;
        if (kkt) kkt[b] = e;
    }
    return 0;
}

// mpc_eval_nlp on the host: Solver::evaluate() at given points z = (X [B][N+1][4], U [B][N][2]) -> f [B], x_next [B][N][4]
extern "C" int wave_eval_batch(int B, int N, double dt, const double *ref_table, int M, const int32_t *ego_index,
                               const double *vref, const double *weights, const uint8_t *is_collide, const double *others, int V,
                               uint32_t flags, double w_distance, double w_collision, const double *X, const double *U, double *f,
                               double *x_next) {
    if (N > mpc::wave::kMaxHorizon) return -1;
    const int cc = (flags & 1u) ? 1 : 0;
    const int Vuse = cc ? V : 0;
    std::vector<double> table((size_t)M * mpc::REF_COLS);
    for (int i = 0; i < M; ++i) {
        table[i * mpc::REF_COLS + mpc::R_X] = ref_table[i * 4 + 0];
        table[i * mpc::REF_COLS + mpc::R_Y] = ref_table[i * 4 + 1];
        table[i * mpc::REF_COLS + mpc::R_H] = ref_table[i * 4 + 3];
        table[i * mpc::REF_COLS + mpc::R_SIN] = std::sin(ref_table[i * 4 + 3]);
        table[i * mpc::REF_COLS + mpc::R_COS] = std::cos(ref_table[i * 4 + 3]);
    }
    mpc::SolveParams P;
    P.N = N; P.V = Vuse; P.max_iter = 0; P.dt = dt; P.tol = 1e-8; P.mu_init = 0.1;
    P.w_distance = w_distance;
    const int SL = mpc::wave::stage_slots(cc);
    const int nd = mpc::wave::lds_doubles(cc, N, Vuse);
    for (int b = 0; b < B; ++b) {
        std::vector<double> L((size_t)nd, NAN);
        HostCtx ctx{};
        ctx.L = L.data();
        ctx.table = table.data();
        ctx.e0 = ego_index[b];
        ctx.M = M;
        ctx.nwords = nd;
        for (int k = 0; k <= N; ++k) {
            int idx = ego_index[b] + k;
            idx = idx > M - 1 ? M - 1 : idx;
            idx = idx < 0 ? 0 : idx;
            L[k * SL + mpc::wave::W_RV] = vref ? vref[(size_t)b * (N + 1) + k] : ref_table[idx * 4 + 2];
            for (int i = 0; i < 4; ++i) L[k * SL + mpc::wave::W_X + i] = X[((size_t)b * (N + 1) + k) * 4 + i];
            if (k < N)
                for (int i = 0; i < 2; ++i) L[k * SL + mpc::wave::W_U + i] = U[((size_t)b * N + k) * 2 + i];
        }
        const int OTH = SL * (N + 1) + mpc::wave::SC_SIZE;
        for (int j = 0; j < Vuse; ++j) {
            const double *ov = others + ((size_t)b * V + j) * 4;
            L[OTH + j * 4 + 0] = ov[0];
            L[OTH + j * 4 + 1] = ov[1];
            L[OTH + j * 4 + 2] = ov[2] * dt * std::cos(ov[3]);
            L[OTH + j * 4 + 3] = ov[2] * dt * std::sin(ov[3]);
        }
        const bool collide = is_collide[b] != 0;
        const double ws_ = collide ? 100.0 : weights[3 * b + 0];
        const double wcoll = (cc && collide) ? 3000.0 * w_collision : 0.0;
        mpc::wave::PerLane<double> xn[4];
        const double *x0 = X + (size_t)b * (N + 1) * 4;
        if (cc) {
            mpc::wave::Solver<true, HostCtx> s(P, ctx, x0, ws_, weights[3 * b + 1], weights[3 * b + 2], wcoll);
            f[b] = s.evaluate(xn);
        } else {
            mpc::wave::Solver<false, HostCtx> s(P, ctx, x0, ws_, weights[3 * b + 1], weights[3 * b + 2], wcoll);
            f[b] = s.evaluate(xn);
        }
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < 4; ++i) x_next[((size_t)b * N + k) * 4 + i] = xn[i].at(k);
    }
    return 0;
}
