// mfma_chain_probe.hip - dependent-issue latencies that bound one Riccati stage of a wave alone on its SIMD (development tool):
//   (a) v_mfma_f64_4x4x4 whose B operand is the previous product's result          [T1 -> M, Ya -> Pxx', ...]
//   (b) the same through the accumulator (C operand)                                [M += ..., twice]
//   (c) mfma -> v_readlane -> scalar-in-VALU multiply -> mfma                       [M -> 2x2 block -> operand build]
//   (d) v_rcp_f64 + one third-order Newton step + multiply, dependent              [reciprocal of the determinant]
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_chain_probe tools/ubench/mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d1 __attribute__((ext_vector_type(1)));
__device__ __forceinline__ double mf(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ void probe(double *out, long long *cyc) {
    double a = 1.0 + 1e-9 * threadIdx.x, b = 0.5 + 1e-9 * threadIdx.x, c = 0.0, w = 1.0000001;
    asm volatile("" : "+v"(a), "+v"(b));
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 128; ++i) {
        if (MODE == 0) b = mf(a, b, 0.0) * 1.0;                 // folded: plain dependent through B
        if (MODE == 1) c = mf(a, b, c);                         // dependent through C
        if (MODE == 2) {
            const double m = mf(a, b, 0.0);
            const double s = __builtin_amdgcn_readlane((int)__double_as_longlong(m), 0) == 12345 ? 2.0 : 1.0;   // low word read
            b = s * m;
        }
        if (MODE == 3) {
            double r = __builtin_amdgcn_rcp(b);
            const double e = __builtin_fma(-b, r, 1.0);
            r = __builtin_fma(r, __builtin_fma(e, e, e), r);
            b = r * w + 0.5;
        }
        if (MODE == 4) b = __builtin_fma(b, w, 1e-9);           // plain dependent FMA
    }
    asm volatile("" ::"v"(b), "v"(c));
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = b + c;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 8);
    const char *names[5] = {"mfma -> B operand of the next mfma", "mfma -> C operand of the next mfma", "mfma -> readlane -> multiply -> mfma", "rcp + Newton + multiply (4 dependent ops)", "dependent v_fma_f64"};
    for (int m = 0; m < 5; ++m) {
        long long best = 1ll << 60;
        for (int r = 0; r < 5; ++r) {
            if (m == 0) hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, out, cyc);
            if (m == 1) hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, out, cyc);
            if (m == 2) hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, out, cyc);
            if (m == 3) hipLaunchKernelGGL(probe<3>, 1, 64, 0, 0, out, cyc);
            if (m == 4) hipLaunchKernelGGL(probe<4>, 1, 64, 0, 0, out, cyc);
            long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            if (h < best) best = h;
        }
        printf("%-45s %7.1f counter ticks per step (128 steps)\n", names[m], best / 128.0);
    }
    // the counter's rate: ticks of a 1 ms wall interval are not measurable here; s_memtime counts at 100 MHz on gfx950
    return 0;
}
