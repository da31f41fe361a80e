// Micro-benchmark: how does the time of a dependent FP64 chain scale with the number of waves per CU and with
// the number of active lanes?  Also reports the shader clock (s_memtime / s_memrealtime).
// (development aid for DESIGN.md section 4; one wave64 per workgroup, grid = waves_per_cu * 256)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void chain(double *out, unsigned long long *clk, int iters, int active, int stride) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const bool on = (lane % stride == 0) && (lane / stride < active);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (on) {
        double a = 1.0 + 1e-9 * lane, b = 0.999999, c = 1e-7;
        lds[lane] = a;
        for (int i = 0; i < iters; ++i) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a = fma(a, b, c);
            } else {
                double a1 = a + 1, a2 = a + 2, a3 = a + 3;
#pragma unroll
                for (int j = 0; j < 2; ++j) { a = fma(a, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); }
                a += 1e-30 * (a1 + a2 + a3);
            }
        }
        out[blockIdx.x * 64 + lane] = a;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
    double *d;
    unsigned long long *clk, h[2];
    hipMalloc(&d, sizeof(double) * 64 * 256 * 32);
    hipMalloc(&clk, 16);
    const int iters = 100000;
    for (int mode = 0; mode < 2; ++mode)
        for (int cfg = 0; cfg < 8; ++cfg) {
            const int actives[8] = {1, 2, 4, 8, 16, 16, 32, 64};
            const int strides[8] = {1, 1, 1, 1, 1, 4, 1, 1};
            const int active = actives[cfg], stride = strides[cfg];
            printf("mode %d active %2d stride %d:", mode, active, stride);
            for (int wpc : {1, 2, 4, 8, 16}) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(256 * wpc), dim3(64), 8192, 0, d, clk, iters, active, stride);
                    if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(256 * wpc), dim3(64), 8192, 0, d, clk, iters, active, stride);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
                printf("  w%-2d %6.1f ns @%4.0f MHz", wpc, ms * 1e6 / iters, (double)h[0] / (double)h[1] * 100.0);
            }
            printf("\n");
        }
    return 0;
}
