// sweep_probe.hip - round 6: the Riccati sweep of mpc_wave.hpp ALONE, one wave, synthetic positive definite stage data, timed
// with the wall clock (no section hooks: the compiler schedules as in the product).  Variants are compile-time macros of
// mpc_wave.hpp; build one binary per variant:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMPC_ROWQUU=0 ...] -o tools/ubench/sweep_probe tools/ubench/sweep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../mpc-rl_for_avs_amd/csrc/mpc_wave_dev.hpp"
#ifndef PROBE_RELAX
#define PROBE_RELAX 7
#endif
namespace {
struct Ctx : mpc::wave::WaveOpsT<PROBE_RELAX> {
    static constexpr int kN = 20;
    __device__ __forceinline__ Ctx(mpc::wave::lds_double_t *l) : mpc::wave::WaveOpsT<PROBE_RELAX>{l} {}
    __device__ __forceinline__ void tick(int) const {}
    __device__ __forceinline__ double ref(int, int) const { return 0.0; }
};
__global__ __launch_bounds__(64, 2) void probe(double *out, long long *ticks, int reps) {
    using namespace mpc::wave;
    extern __shared__ double smem[];
    constexpr int N = 20;
    mpc::SolveParams P;
    P.N = N; P.V = 8; P.max_iter = 100; P.dt = 0.1; P.tol = 1e-8; P.mu_init = 0.1; P.stall_window = 0; P.strict_kink = 0; P.w_distance = 10.0;
    Ctx ctx((lds_double_t *)smem);
    double x0[4] = {2.0, 40.0, -1.5, 8.0};
    Solver<true, Ctx> s(P, ctx, x0, 1.0, 1.0, 1.0, 0.0);
    s.init_tables();
    s.sc(SC_K + K_RD, 0.02);
    const int lane = threadIdx.x;
    long long acc = 0;
    double chk = 0.0;
    for (int r = 0; r < reps; ++r) {
        ctx.phase([&](int ln) {
            if (ln > N) return;
            const int k = ln;
            const double e = 1e-3 * k + 1e-4 * r;
            for (int i = 0; i < 12; ++i) s.S(k, W_LIN + i, 0.0);
            s.S(k, W_LIN + LIN_A02, -0.05 + e); s.S(k, W_LIN + LIN_A12, 0.08 - e); s.S(k, W_LIN + LIN_A03, 0.07); s.S(k, W_LIN + LIN_A13, 0.07 + e);
            s.S(k, W_LIN + LIN_A23, 0.01); s.S(k, W_LIN + LIN_B01, -0.02 - e); s.S(k, W_LIN + LIN_B11, 0.03); s.S(k, W_LIN + LIN_B21, 0.05 + e);
            s.S(k, A_L00, 12.0 + e); s.S(k, A_L01, 0.5); s.S(k, A_L11, 10.0 - e); s.S(k, A_H22, 5.0); s.S(k, A_H23, 0.1 + e); s.S(k, A_H33, 20.0);
            s.S(k, A_WTD, 0.01); s.S(k, A_WVD, 0.02 - e); s.S(k, A_H66, 0.05 + e); s.S(k, A_H77, 0.06);
            for (int i = 0; i < 4; ++i) s.S(k, 6 + A_HV0 + i, 0.1 * (i + 1) - e);
            s.S(k, A_HV4, 0.01); s.S(k, 6 + A_HV5, -0.01); s.S(k, A_HV6, 0.02 + e); s.S(k, A_HV7, -0.03);
            s.S(k, W_X + 2, -1.5); s.S(k, W_X + 3, 8.0); s.S(k, W_ZXL + 0, 1.0); s.S(k, W_ZXL + 1, 1.0); s.S(k, W_ZXU + 0, 1.0); s.S(k, W_ZXU + 1, 1.0);
        });
        s.set_roles4(6);
        double dV1 = 0.0;
        const long long t0 = wall_clock64();
        const bool ok = s.sweep4(0, 6, 0.1, 0.0, false, dV1);
        const long long t1 = wall_clock64();
        acc += t1 - t0;
        chk += dV1 + (ok ? 1.0 : 1000.0);
    }
    out[lane] = chk + s.S(lane % N, W_KX + (lane & 7));
    if (lane == 0) ticks[0] = acc;
}
}  // namespace
int main() {
    double *out; long long *tk;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&tk, 8);
    const size_t lds = (size_t)mpc::wave::lds_doubles(true, 20, 8) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int reps = 200;
    long long best = 1ll << 60;
    double h[64];
    for (int r = 0; r < 5; ++r) {
        hipLaunchKernelGGL(probe, 1, 64, lds, 0, out, tk, reps);
        long long t; (void)hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
        if (t < best) best = t;
    }
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("sweep: %8.1f ns per sweep = %6.1f ns per stage   (check %.12g %.12g)\n", best * 10.0 / reps, best * 10.0 / reps / 20.0, h[0], h[17]);
    return 0;
}
