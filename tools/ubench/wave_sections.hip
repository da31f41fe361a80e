// wave_sections.hip - where does an iteration of the wave-cooperative solver spend its cycles?
// Development tool (not part of the product): the same mpc_wave.hpp solver with a context whose `tick(section)`
// accumulates s_memtime deltas per section, run for B instances (one wave each), cycles returned per instance.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libwave_sections.so tools/ubench/wave_sections.hip
// driven by tools/gpu_wave_sections.py.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../mpc-rl_for_avs_amd/csrc/mpc_wave_dev.hpp"

namespace {

#ifndef PROF_FINE
#define PROF_FINE 1
#endif
#ifndef PROF_RELAX
#define PROF_RELAX 0      // 15: the latency build's configuration (everything hoisted, fused linear step, precomputed trial bounds)
#endif
struct ProfCtx : mpc::wave::WaveOpsT<PROF_RELAX> {
    static constexpr int kN = 20;
    static constexpr bool kFine = PROF_FINE != 0;     // also attribute the parts of a rollout stage
    const double *table;
    int e0, M;
    unsigned long long *acc;   // [T_COUNT] accumulators of this instance in LDS, behind the solver's words (lane 0 adds)
    unsigned long long last;
    __device__ __forceinline__ ProfCtx(mpc::wave::lds_double_t *l, const double *t, int e, int m, unsigned long long *a)
        : mpc::wave::WaveOpsT<PROF_RELAX>{l}, table(t), e0(e), M(m), acc(a), last(0ull) {}
    // LDS accumulation: a tick is the counter read plus one LDS add (a global read-modify-write per tick would cost
    // hundreds of cycles and land in whichever section waits for memory next)
    __device__ __forceinline__ void tick(int s) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (threadIdx.x == 0) acc[s] += now - last;
        last = now;
    }
    __device__ __forceinline__ double ref(int k, int c) const {
        int idx = e0 + k;
        idx = idx > M - 1 ? M - 1 : idx;
        idx = idx < 0 ? 0 : idx;
        return table[idx * mpc::REF_COLS + c];
    }
};

template <bool CC>
__global__ __launch_bounds__(64, 2) void prof_kernel(mpc::SolveParams P, int B, const double *ref5, int M,
                                                     const double *state, const int32_t *ego_index, const double *vref,
                                                     const double *weights, const uint8_t *is_collide,
                                                     const double *others, int Vin, double *u0_out,
                                                     int32_t *status_out, int32_t *iters_out,
                                                     unsigned long long *cycles) {
    extern __shared__ double smem[];
    constexpr int N = 20;
    const int b = blockIdx.x, lane = threadIdx.x;
    constexpr bool kPre = (PROF_RELAX & 8) != 0;
    constexpr int SL = mpc::wave::stage_slots(CC);
    unsigned long long *lacc = reinterpret_cast<unsigned long long *>(smem + mpc::wave::lds_doubles(CC, N, P.V));
    if (lane < mpc::wave::T_COUNT) lacc[lane] = 0ull;
    ProfCtx ctx((mpc::wave::lds_double_t *)smem, ref5, ego_index[b], M, lacc);
    const int OTH = SL * (N + 1) + mpc::wave::SC_SIZE;
    if (lane <= N) ctx.st(lane * SL + mpc::wave::W_RV, vref[(size_t)b * (N + 1) + lane]);
    if (CC && lane < P.V) {
        const double *ov = others + ((size_t)b * Vin + lane) * 4;
        const double sp = ov[2] * P.dt, hh = ov[3];
        ctx.st(OTH + lane * 4 + 0, ov[0]);
        ctx.st(OTH + lane * 4 + 1, ov[1]);
        ctx.st(OTH + lane * 4 + 2, sp * cos(hh));
        ctx.st(OTH + lane * 4 + 3, sp * sin(hh));
    }
    __syncthreads();
    double x0[4];
    for (int i = 0; i < 4; ++i) x0[i] = state[(size_t)b * 4 + i];
    const bool collide = is_collide[b] != 0;
    const double ws_ = collide ? 100.0 : weights[(size_t)b * 3 + 0];
    const double wcoll = (CC && collide) ? 3000.0 : 0.0;
    mpc::wave::Solver<CC, ProfCtx> solver(P, ctx, x0, ws_, weights[(size_t)b * 3 + 1], weights[(size_t)b * 3 + 2], wcoll);
    int status, iters, cur;
    double kkt;
    ctx.last = __builtin_readcyclecounter();
    solver.solve(status, iters, cur, kkt);
    __syncthreads();
    if (lane < 2) u0_out[(size_t)b * 2 + lane] = ctx.ld(cur * 6 + mpc::wave::W_U + lane);
    if (lane == 0) {
        status_out[b] = status;
        iters_out[b] = iters;
    }
    if (lane < mpc::wave::T_COUNT) cycles[(size_t)b * mpc::wave::T_COUNT + lane] = lacc[lane];
}
}  // namespace

// all pointers are DEVICE pointers (the driver passes torch tensors); cycles: [B][T_COUNT] uint64, zeroed by the caller
extern "C" int wave_sections(int B, int V, int cc, int max_iter, const double *ref5, int M, const double *state,
                             const int32_t *ego_index, const double *vref, const double *weights,
                             const uint8_t *is_collide, const double *others, double *u0, int32_t *status,
                             int32_t *iters, unsigned long long *cycles) {
    mpc::SolveParams P;
    P.N = 20;
    P.V = cc ? V : 0;
    P.max_iter = max_iter;
    P.dt = 0.1;
    P.tol = 1e-8;
    P.mu_init = 0.1;
    P.stall_window = 0;
    P.strict_kink = 0;
    P.w_distance = 10.0;
    const size_t lds = (size_t)(mpc::wave::lds_doubles(cc != 0, 20, P.V) + mpc::wave::T_COUNT) * sizeof(double);
    if (cc) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(prof_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(prof_kernel<true>, dim3(B), dim3(64), lds, 0, P, B, ref5, M, state, ego_index, vref, weights,
                           is_collide, others, V, u0, status, iters, cycles);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(prof_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(prof_kernel<false>, dim3(B), dim3(64), lds, 0, P, B, ref5, M, state, ego_index, vref, weights,
                           is_collide, others, V, u0, status, iters, cycles);
    }
    return (int)hipDeviceSynchronize();
}

extern "C" int wave_sections_count(void) { return mpc::wave::T_COUNT; }
