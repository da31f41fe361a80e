// Host build of mpc-rl_for_avs_amd/csrc/mpc_wave.hpp for tests only (-m "not gpu"): the wave-cooperative solver
// with its 64 lanes emulated by loops (each `phase` runs lane 0..63 in turn), compared with the oracle on the CPU.
// Never loaded by the product.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../mpc-rl_for_avs_amd/csrc/mpc_wave.hpp"

namespace {
struct HostCtx {
    static constexpr int kN = 0;
    double *L;
    const double *table;  // [M][REF_COLS]
    int e0, M;
    double ld(int i) const { return L[i]; }
    void st(int i, double v) { L[i] = v; }
    template <class F>
    void phase(F &&f) {
        for (int lane = 0; lane < mpc::wave::kLanes; ++lane) f(lane);
    }
    void tick(int) const {}
    template <class F>
    void lanes(F &&f) {
        for (int lane = 0; lane < mpc::wave::kLanes; ++lane) f(lane);
    }
    // v_mfma_f64_4x4x4f64: lane l = 16 hi + 4 blk + lo; D_blk[hi][lo] = C + sum_k A_blk[hi][k] B_blk[k][lo] with
    // A_blk[row][k] in lane 16 k + 4 blk + row and B_blk[k][col] in lane 16 k + 4 blk + col (probed on MI355X,
    // tools/ubench/mfma_f64_probe.hip)
    void mfma(mpc::wave::PerLane<double> &a, mpc::wave::PerLane<double> &b, mpc::wave::PerLane<double> &cd) const {
        double out[mpc::wave::kLanes];
        for (int l = 0; l < mpc::wave::kLanes; ++l) {
            const int hi = l >> 4, blk = (l >> 2) & 3, lo = l & 3;
            double acc = cd.v[l];
            for (int k = 0; k < 4; ++k) acc = std::fma(a.v[16 * k + 4 * blk + hi], b.v[16 * k + 4 * blk + lo], acc);
            out[l] = acc;
        }
        for (int l = 0; l < mpc::wave::kLanes; ++l) cd.v[l] = out[l];
    }
    void take(mpc::wave::PerLane<double> &dst, mpc::wave::PerLane<double> &src, mpc::wave::PerLane<int> &from) const {
        double out[mpc::wave::kLanes];
        for (int l = 0; l < mpc::wave::kLanes; ++l) out[l] = src.v[from.v[l]];
        for (int l = 0; l < mpc::wave::kLanes; ++l) dst.v[l] = out[l];
    }
    double lane_get(mpc::wave::PerLane<double> &p, int lane) const { return p.v[lane]; }
    double wave_sum(mpc::wave::PerLane<double> &p) const {
        return mpc::wave::host_reduce(p, [](double a, double b) { return a + b; });
    }
    double wave_max(mpc::wave::PerLane<double> &p) const {
        return mpc::wave::host_reduce(p, [](double a, double b) { return mpc::fmax2(a, b); });
    }
    double wave_min(mpc::wave::PerLane<double> &p) const {
        return mpc::wave::host_reduce(p, [](double a, double b) { return mpc::fmin2(a, b); });
    }
    void wave_sum2(mpc::wave::PerLane<double> &p, double &lo, double &hi) const {
        mpc::wave::host_row_reduce(p, [](double a, double b) { return a + b; });
        lo = p.v[0] + p.v[16];
        hi = p.v[32] + p.v[48];
    }
    int wave_bcast(mpc::wave::PerLane<int> &p, int lane) const { return p.v[lane]; }
    void wave_max_ratio(mpc::wave::PerLane<double> &pn, mpc::wave::PerLane<double> &pd, double &rn, double &rd) const {
        for (int step = 0; step < 4; ++step) {
            double nn[mpc::wave::kLanes], nd[mpc::wave::kLanes];
            for (int l = 0; l < mpc::wave::kLanes; ++l) {
                const int q = mpc::wave::row_partner(l, step);
                const bool take = mpc::wave::ratio_greater(pn.v[q], pd.v[q], pn.v[l], pd.v[l]);
                nn[l] = take ? pn.v[q] : pn.v[l];
                nd[l] = take ? pd.v[q] : pd.v[l];
            }
            for (int l = 0; l < mpc::wave::kLanes; ++l) {
                pn.v[l] = nn[l];
                pd.v[l] = nd[l];
            }
        }
        rn = pn.v[0];
        rd = pd.v[0];
        for (int r = 16; r < mpc::wave::kLanes; r += 16)
            if (mpc::wave::ratio_greater(pn.v[r], pd.v[r], rn, rd)) {
                rn = pn.v[r];
                rd = pd.v[r];
            }
    }
    double ref(int k, int c) const {
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        return table[idx * mpc::REF_COLS + c];
    }
};

template <bool CC>
void run(const mpc::SolveParams &P, HostCtx &ctx, const double *x0, double ws, double wc, double wd, double wcoll,
         int &st, int &it, int &cur, double &e, bool warm) {
    mpc::wave::Solver<CC, HostCtx> s(P, ctx, x0, ws, wc, wd, wcoll);
    s.solve(st, it, cur, e, warm);
}
}  // namespace

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
    const int SL = mpc::wave::stage_slots(cc);
    const int nd = mpc::wave::lds_doubles(cc, N, Vuse);
    for (int b = 0; b < B; ++b) {
        std::vector<double> L((size_t)nd, NAN);
        HostCtx ctx{L.data(), table.data(), ego_index[b], M};
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
