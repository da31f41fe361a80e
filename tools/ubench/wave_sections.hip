// wave_sections.hip - where does an iteration of the wave-cooperative solver spend its cycles?
// Development tool (not part of the product): the same mpc_wave.hpp solver with a context whose `tick(section)`
// accumulates s_memtime deltas per section, run for B instances (one wave each), cycles returned per instance.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libwave_sections.so tools/ubench/wave_sections.hip
// driven by tools/gpu_wave_sections.py.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../mpc-rl_for_avs_amd/csrc/mpc_wave.hpp"

namespace {
typedef __attribute__((address_space(3))) double lds_double;

struct ProfCtx {
    static constexpr int kN = 20;
    lds_double *L;
    const double *table;
    int e0, M;
    unsigned long long *acc;   // [T_COUNT] of this instance (lane 0 writes)
    unsigned long long last;
    __device__ __forceinline__ double ld(int i) const { return L[i]; }
    __device__ __forceinline__ void st(int i, double v) { L[i] = v; }
    template <class F>
    __device__ __forceinline__ void phase(F &&f) {
        f((int)threadIdx.x);
        __syncthreads();
    }
    __device__ __forceinline__ void tick(int s) {
        const unsigned long long now = __builtin_readcyclecounter();
        if (threadIdx.x == 0) acc[s] += now - last;
        last = now;
    }
    // register-only per-lane work, the FP64 matrix core, lane permutation, lane broadcast
    template <class F>
    __device__ __forceinline__ void lanes(F &&f) {
        f((int)threadIdx.x);
    }
    __device__ __forceinline__ void mfma(mpc::wave::PerLane<double> &a, mpc::wave::PerLane<double> &b,
                                         mpc::wave::PerLane<double> &cd) const {
        cd.v = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v, b.v, cd.v, 0, 0, 0);
    }
    __device__ __forceinline__ void take(mpc::wave::PerLane<double> &dst, mpc::wave::PerLane<double> &src,
                                         mpc::wave::PerLane<int> &from) const {
        dst.v = __shfl(src.v, from.v);
    }
    __device__ __forceinline__ double lane_get(mpc::wave::PerLane<double> &p, int lane) const {
        const long long bits = __double_as_longlong(p.v);
        const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), lane);
        const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), lane);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    // wave reductions: xor-butterfly of lane shuffles, every lane ends with the result
    __device__ __forceinline__ double wave_sum(mpc::wave::PerLane<double> &p) const {
        double v = p.v;
#pragma unroll
        for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
        return v;
    }
    __device__ __forceinline__ double wave_max(mpc::wave::PerLane<double> &p) const {
        double v = p.v;
#pragma unroll
        for (int off = 32; off; off >>= 1) v = mpc::fmax2(v, __shfl_xor(v, off));
        return v;
    }
    __device__ __forceinline__ double wave_min(mpc::wave::PerLane<double> &p) const {
        double v = p.v;
#pragma unroll
        for (int off = 32; off; off >>= 1) v = mpc::fmin2(v, __shfl_xor(v, off));
        return v;
    }
    __device__ __forceinline__ void wave_sum2(mpc::wave::PerLane<double> &p, double &lo, double &hi) const {
        double v = p.v;
#pragma unroll
        for (int off = 16; off; off >>= 1) v += __shfl_xor(v, off);
        lo = __shfl(v, 0);
        hi = __shfl(v, 32);
    }
    __device__ __forceinline__ int wave_bcast(mpc::wave::PerLane<int> &p, int lane) const { return __shfl(p.v, lane); }
    __device__ __forceinline__ void wave_max_ratio(mpc::wave::PerLane<double> &pn, mpc::wave::PerLane<double> &pd,
                                                   double &rn, double &rd) const {
        double n = pn.v, d = pd.v;
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const double n2 = __shfl_xor(n, off), d2 = __shfl_xor(d, off);
            const bool take = mpc::wave::ratio_greater(n2, d2, n, d);
            n = take ? n2 : n;
            d = take ? d2 : d;
        }
        rn = n;
        rd = d;
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
    constexpr int SL = mpc::wave::stage_slots(CC);
    ProfCtx ctx{(lds_double *)smem, ref5, ego_index[b], M, cycles + (size_t)b * mpc::wave::T_COUNT, 0ull};
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
    P.w_distance = 10.0;
    const size_t lds = (size_t)mpc::wave::lds_doubles(cc != 0, 20, P.V) * sizeof(double);
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
