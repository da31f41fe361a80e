// issue_probe.hip - round 6: issue / latency figures of the FP64 pipes on gfx950, unrolled chains (64 per trip), one wave
//   0 dependent v_fma_f64                      1 two independent fma chains (per pair)       2 four independent (per 4)
//   3 dependent mfma_f64_4x4x4 (B operand)     4 two independent mfma chains (per pair)      5 four independent (per 4)
//   6 one mfma + 4 independent fmas per step (does the VALU work hide under the matrix core?)    7 one mfma + 8 fmas
//   8 dependent: mfma -> fma -> mfma            9 v_rcp_f64 dependent                         10 ds_write -> ds_read round trip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double mf(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
constexpr int kOuter = 256, kIn = 64;
template <int MODE>
__global__ void probe(double *out, long long *ticks) {
    __shared__ double lds[64];
    double a = 1e-3 * (1 + (threadIdx.x & 3)), w = 1.0000001, e = 1e-9;
    double b0 = 0.5 + 1e-9 * threadIdx.x, b1 = 0.6, b2 = 0.7, b3 = 0.8, b4 = 0.9, b5 = 1.0, b6 = 1.1, b7 = 1.2, b8 = 1.3;
    asm volatile("" : "+v"(a), "+v"(w), "+v"(e), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    const long long t0 = wall_clock64();
#pragma unroll 1
    for (int o = 0; o < kOuter; ++o) {
#pragma unroll
        for (int i = 0; i < kIn; ++i) {
            if (MODE == 0) b0 = __builtin_fma(b0, w, e);
            if (MODE == 1) { b0 = __builtin_fma(b0, w, e); b1 = __builtin_fma(b1, w, e); }
            if (MODE == 2) { b0 = __builtin_fma(b0, w, e); b1 = __builtin_fma(b1, w, e); b2 = __builtin_fma(b2, w, e); b3 = __builtin_fma(b3, w, e); }
            if (MODE == 3) b0 = mf(a, b0, 0.0);
            if (MODE == 4) { b0 = mf(a, b0, 0.0); b1 = mf(a, b1, 0.0); }
            if (MODE == 5) { b0 = mf(a, b0, 0.0); b1 = mf(a, b1, 0.0); b2 = mf(a, b2, 0.0); b3 = mf(a, b3, 0.0); }
            if (MODE == 6) { b0 = mf(a, b0, 0.0); b1 = __builtin_fma(b1, w, e); b2 = __builtin_fma(b2, w, e); b3 = __builtin_fma(b3, w, e); b4 = __builtin_fma(b4, w, e); }
            if (MODE == 7) { b0 = mf(a, b0, 0.0); b1 = __builtin_fma(b1, w, e); b2 = __builtin_fma(b2, w, e); b3 = __builtin_fma(b3, w, e); b4 = __builtin_fma(b4, w, e);
                             b5 = __builtin_fma(b5, w, e); b6 = __builtin_fma(b6, w, e); b7 = __builtin_fma(b7, w, e); b8 = __builtin_fma(b8, w, e); }
            if (MODE == 6 || MODE == 7) __builtin_amdgcn_sched_barrier(0);
            if (MODE == 8) { b0 = mf(a, b0, 0.0); b0 = __builtin_fma(b0, w, e); }
            if (MODE == 9) b0 = __builtin_amdgcn_rcp(b0);
            if (MODE == 10) { lds[threadIdx.x] = b0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); b0 = lds[threadIdx.x ^ 1] + e; }
        }
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + b8;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
#define RUN(M) case M: hipLaunchKernelGGL(probe<M>, 1, 64, 0, 0, out, tk); break;
int main() {
    double *out; long long *tk;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&tk, 8);
    const char *names[11] = {"dependent v_fma_f64", "2 independent fma chains (per pair)", "4 independent fma chains (per 4)", "dependent mfma (B operand)",
                             "2 independent mfma chains (per pair)", "4 independent mfma chains (per 4)", "mfma chain + 4 independent fma per step",
                             "mfma chain + 8 independent fma per step", "mfma -> fma -> mfma (per pair)", "dependent v_rcp_f64", "ds_write -> ds_read round trip (+ add)"};
    for (int m = 0; m < 11; ++m) {
        long long best = 1ll << 60;
        for (int r = 0; r < 3; ++r) {
            switch (m) { RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) }
            long long h; (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
            if (h < best) best = h;
        }
        const double ns = best * 10.0 / (kOuter * kIn);
        printf("%-48s %7.2f ns per step = %6.1f cycles at 2.4 GHz\n", names[m], ns, ns * 2.4);
    }
    return 0;
}
