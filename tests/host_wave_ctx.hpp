// Host model of the CTX interface of mpc-rl_for_avs_amd/csrc/mpc_wave.hpp for tests only (-m "not gpu"): the 64
// lanes of a wave emulated by loops (each `phase` runs lane 0..63 in turn), the FP64 matrix core, the lane
// permutations and the wave reductions modelled as the device executes them.  Shared by the host harnesses of the
// solvers (cpu_wave_harness.cpp, cpu_ltv_harness.cpp).  Never loaded by the product.
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../mpc-rl_for_avs_amd/csrc/mpc_wave.hpp"

struct HostCtx {
    static constexpr int kN = 0;
    double *L;
    const double *table;  // [M][REF_COLS]
    int e0, M;
    const double *speeds = nullptr;  // [M] speed column of the reference table (refv)
    int nwords = 0;                  // size of L in doubles (lds_doubles()): every access is checked against it when set
    void check(int i) const {
        if (nwords > 0 && (i < 0 || i >= nwords)) {
            std::fprintf(stderr, "host_wave_ctx: LDS word %d outside the instance's %d words\n", i, nwords);
            std::abort();
        }
    }
    double ld(int i) const {
        check(i);
        return L[i];
    }
    void st(int i, double v) {
        check(i);
        L[i] = v;
    }
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
    double lane_get(mpc::wave::PerLane<double> &p, int lane) const { return p.v[lane]; }
    // v_mov_b64_dpp row_newbcast:J - every lane takes lane J of its own 16-lane row
    template <int J>
    void row_bcast(mpc::wave::PerLane<double> &dst, mpc::wave::PerLane<double> &src) const {
        double out[mpc::wave::kLanes];
        for (int l = 0; l < mpc::wave::kLanes; ++l) out[l] = src.v[(l & ~15) + J];
        for (int l = 0; l < mpc::wave::kLanes; ++l) dst.v[l] = out[l];
    }
    // lanes 0..7 of a row take lane J, lanes 8..15 lane J + 8 of their row
    template <int J>
    void row_bcast2(mpc::wave::PerLane<double> &dst, mpc::wave::PerLane<double> &src) const {
        double out[mpc::wave::kLanes];
        for (int l = 0; l < mpc::wave::kLanes; ++l) out[l] = src.v[(l & ~15) + J + ((l & 8) ? 8 : 0)];
        for (int l = 0; l < mpc::wave::kLanes; ++l) dst.v[l] = out[l];
    }
    int hide(int v) const { return v; }
    template <int PRIO>
    void set_priority() const {}
    double bit_select(int m, double a, double b) const {      // m ? a : b bit by bit, m = 0 or ~0
        unsigned long long ua, ub;
        __builtin_memcpy(&ua, &a, 8);
        __builtin_memcpy(&ub, &b, 8);
        const unsigned long long mm = ((unsigned long long)(unsigned)m << 32) | (unsigned)m;
        const unsigned long long ur = (ua & mm) | (ub & ~mm);
        double r;
        __builtin_memcpy(&r, &ur, 8);
        return r;
    }
    int opaque(int v) const { return v; }
    int opaque_shared(int v) const { return v; }
    double fresh(double v) const { return v; }
    void sched_fence() const {}
    double keep(double v) const { return v; }
    double uni(double v) const { return v; }
    double wave_sum(mpc::wave::PerLane<double> &p) const {
        return mpc::wave::host_reduce(p, [](double a, double b) { return a + b; });
    }
    double wave_max(mpc::wave::PerLane<double> &p) const {
        return mpc::wave::host_reduce(p, [](double a, double b) { return mpc::fmax2(a, b); });
    }
    double wave_min(mpc::wave::PerLane<double> &p) const {
        return mpc::wave::host_reduce(p, [](double a, double b) { return mpc::fmin2(a, b); });
    }
    // same order of additions as the device: shifts by 1, 2, 4, 8 inside each 16-lane row, then the rows above
    void wave_suffix_sum(mpc::wave::PerLane<double> &p) const {
        for (int sh = 1; sh <= 8; sh *= 2) {
            double nv[mpc::wave::kLanes];
            for (int l = 0; l < mpc::wave::kLanes; ++l) nv[l] = p.v[l] + (((l & 15) + sh < 16) ? p.v[l + sh] : 0.0);
            for (int l = 0; l < mpc::wave::kLanes; ++l) p.v[l] = nv[l];
        }
        const double t1 = p.v[16], t2 = p.v[32], t3 = p.v[48];
        for (int l = 0; l < mpc::wave::kLanes; ++l) {
            const int row = l >> 4;
            p.v[l] += row == 0 ? (t1 + (t2 + t3)) : (row == 1 ? (t2 + t3) : (row == 2 ? t3 : 0.0));
        }
    }
    void wave_sum2(mpc::wave::PerLane<double> &p, double &lo, double &hi) const {
        mpc::wave::host_row_reduce(p, [](double a, double b) { return a + b; });
        lo = p.v[0] + p.v[16];
        hi = p.v[32] + p.v[48];
    }
    int wave_bcast(mpc::wave::PerLane<int> &p, int lane) const { return p.v[lane]; }
    unsigned long long ballot(mpc::wave::PerLane<int> &p) const {
        unsigned long long m = 0;
        for (int l = 0; l < mpc::wave::kLanes; ++l) m |= (unsigned long long)(p.v[l] != 0) << l;
        return m;
    }
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
    double refv(int k) const {
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        return speeds[idx];
    }
    double ref(int k, int c) const {
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        return table[idx * mpc::REF_COLS + c];
    }
};


// mpc_ltv.hpp's code path of the latency build (residuals kept in registers: relax_bits)
struct HostCtxLtvRelaxed : HostCtx {
    static constexpr int kRelax = 8;
};
