// clock_probe.hip - shader clock seen by a lone wave: s_memtime (shader cycles) against s_memrealtime (100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long *o, int n) {
    double b = 1.0 + threadIdx.x * 1e-9, w = 1.0000001;
    const long long c0 = __builtin_readcyclecounter(), t0 = wall_clock64();
    for (int i = 0; i < n; ++i) b = __builtin_fma(b, w, 1e-9);
    const long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    if (threadIdx.x == 0) { o[0] = c1 - c0; o[1] = t1 - t0; o[2] = (long long)b; }
}
int main() {
    long long *d, h[3];
    (void)hipMalloc(&d, 24);
    for (int n : {100000, 1000000, 4000000}) {
        for (int grid : {1, 2048}) {
            hipLaunchKernelGGL(probe, grid, 64, 0, 0, d, n);
            (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            printf("n %8d grid %4d: %lld counter ticks in %.1f us -> %.3f GHz;  %.2f ns per dependent fma\n", n, grid, h[0], h[1] * 0.01, h[0] / (h[1] * 10.0), h[1] * 10.0 / n);
        }
    }
    return 0;
}
