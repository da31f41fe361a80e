// quu_probe.hip - round 6: what does it cost to get an entry of a matrix-core result to all lanes?  Dependent chains of
//   (0) mfma -> mfma (B operand)                         the floor
//   (1) mfma -> v_readlane x2 -> v_add (scalar operand) -> v_mul -> mfma          the round-5 Riccati stage
//   (2) mfma -> v_mov_b64_dpp row_newbcast -> v_add -> v_mul -> mfma              the per-lane form
//   (3) mfma -> 5 x row_newbcast -> 5 adds -> fma -> mul -> mfma                  the whole 2x2 read-out, per lane
//   (4) mfma -> 10 x v_readlane -> 5 adds -> fma -> mul -> mfma                   the whole 2x2 read-out, scalar
//   (5) v_fma -> row_newbcast -> v_fma                                            (VALU result, not a matrix-core result)
//   (6) mfma -> v_add (plain VALU read of the result) -> v_mul -> mfma
// timed with the 100 MHz wall clock over many steps; prints ns per step (x 2.4 = shader cycles at 2.4 GHz)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/quu_probe tools/ubench/quu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double mf(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int J>
__device__ __forceinline__ double bc(double v) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + J, 0xf, 0xf, true));
}
__device__ __forceinline__ double rl(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kSteps = 4096;
template <int MODE>
__global__ void probe(double *out, long long *ticks) {
    double a = 1e-3 * (1 + (threadIdx.x & 3)), b = 0.5 + 1e-9 * threadIdx.x, w = 1.0000001, p0 = 1e-12, p1 = 2e-12;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(w), "+v"(p0), "+v"(p1));
    const long long t0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < kSteps; ++i) {
        if (MODE == 0) b = mf(a, b, 0.0);
        if (MODE == 1) { const double m = mf(a, b, 0.0); b = (rl(m, 0) + p0) * w; }
        if (MODE == 2) { const double m = mf(a, b, 0.0); b = (bc<0>(m) + p0) * w; }
        if (MODE == 3) {
            const double m = mf(a, b, 0.0);
            const double ha = bc<0>(m) + p0, hb = bc<1>(m) + p1, hc = bc<5>(m) + p0, h0 = bc<2>(m) + p1, h1 = bc<6>(m) + p0;
            b = (ha * hc - hb * hb) * w + (h0 - h1);
        }
        if (MODE == 4) {
            const double m = mf(a, b, 0.0);
            const double ha = rl(m, 0) + p0, hb = rl(m, 1) + p1, hc = rl(m, 17) + p0, h0 = rl(m, 2) + p1, h1 = rl(m, 18) + p0;
            b = (ha * hc - hb * hb) * w + (h0 - h1);
        }
        if (MODE == 5) { const double m = __builtin_fma(b, w, p0); b = __builtin_fma(bc<0>(m), w, p1); }
        if (MODE == 6) { const double m = mf(a, b, 0.0); b = (m + p0) * w; }
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = b;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double *out; long long *tk;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&tk, 8);
    const char *names[7] = {"mfma -> mfma (B operand)", "mfma -> readlane x2 -> add -> mul -> mfma", "mfma -> row_newbcast -> add -> mul -> mfma",
                            "mfma -> 5 bcast -> 2x2 det -> mfma", "mfma -> 10 readlane -> 2x2 det -> mfma", "fma -> row_newbcast -> fma",
                            "mfma -> add -> mul -> mfma"};
    for (int m = 0; m < 7; ++m) {
        long long best = 1ll << 60;
        for (int r = 0; r < 5; ++r) {
            if (m == 0) hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, out, tk);
            if (m == 1) hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, out, tk);
            if (m == 2) hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, out, tk);
            if (m == 3) hipLaunchKernelGGL(probe<3>, 1, 64, 0, 0, out, tk);
            if (m == 4) hipLaunchKernelGGL(probe<4>, 1, 64, 0, 0, out, tk);
            if (m == 5) hipLaunchKernelGGL(probe<5>, 1, 64, 0, 0, out, tk);
            if (m == 6) hipLaunchKernelGGL(probe<6>, 1, 64, 0, 0, out, tk);
            long long h; (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
            if (h < best) best = h;
        }
        printf("%-48s %7.2f ns per step  (~%5.1f cycles at 2.4 GHz)\n", names[m], best * 10.0 / kSteps, best * 10.0 / kSteps * 2.4);
    }
    return 0;
}
