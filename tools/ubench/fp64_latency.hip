// fp64_latency.hip - dependent-issue latency of FP64 operations for a lone wave on MI355X, and the raw accuracy of
// v_rcp_f64 / v_rsq_f64 (how many Newton steps the lean math needs).  Development tool.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>

template <int OP>
__global__ void chain(double *out, long long *cycles, double seed) {
    double d = seed + 1e-9 * threadIdx.x, a = 1.0000001, b = 1e-9;
    asm volatile("" : "+v"(d));
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        if (OP == 0) d = __builtin_fma(d, a, b);
        if (OP == 1) d = d * a;
        if (OP == 2) d = d + b;
        if (OP == 3) d = __builtin_amdgcn_rcp(d) + 1.5;
        if (OP == 4) d = __builtin_amdgcn_rsq(d) + 1.5;
        if (OP == 5) d = __builtin_fmax(d, b) ;
        if (OP == 6) d = __builtin_rint(d * a);
    }
    asm volatile("" ::"v"(d));
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = d;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
// two independent chains: does a lone wave overlap them?
__global__ void chain2(double *out, long long *cycles) {
    double d = 1.0 + 1e-9 * threadIdx.x, e = 2.0 + 1e-9 * threadIdx.x, a = 1.0000001, b = 1e-9;
    asm volatile("" : "+v"(d), "+v"(e));
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 256; ++i) {
        d = __builtin_fma(d, a, b);
        e = __builtin_fma(e, a, b);
    }
    asm volatile("" ::"v"(d), "v"(e));
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = d + e;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
__global__ void accuracy(double *err) {
    double worst_rcp = 0, worst_rsq = 0;
    for (int i = 0; i < 4096; ++i) {
        const double x = 0.37 + 1.731e-3 * (i * 64 + threadIdx.x);
        const double r = __builtin_amdgcn_rcp(x), q = __builtin_amdgcn_rsq(x);
        worst_rcp = fmax(worst_rcp, fabs(r * x - 1.0));
        worst_rsq = fmax(worst_rsq, fabs(q * q * x - 1.0) * 0.5);
    }
    err[2 * threadIdx.x] = worst_rcp;
    err[2 * threadIdx.x + 1] = worst_rsq;
}
int main() {
    double *out, *err;
    long long *cyc, c;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&err, 128 * 8);
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64 + add", "v_rsq_f64 + add", "v_max_f64", "v_rndne(mul)"};
#define RUN(OP) hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.25); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); \
    printf("dependent %-18s %.1f cycles per op\n", names[OP], (double)c / 256);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    hipLaunchKernelGGL(chain2, dim3(1), dim3(64), 0, 0, out, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("two independent fma chains: %.1f cycles per pair\n", (double)c / 256);
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, err);
    double h[128]; hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost);
    double wr = 0, wq = 0; for (int i = 0; i < 64; ++i) { wr = fmax(wr, h[2*i]); wq = fmax(wq, h[2*i+1]); }
    printf("raw relative error: v_rcp_f64 %.3g (2^%.1f), v_rsq_f64 %.3g (2^%.1f)\n", wr, log2(wr), wq, log2(wq));
    return 0;
}
