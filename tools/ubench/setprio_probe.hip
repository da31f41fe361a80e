// setprio_probe.hip - does s_setprio let one wave of a SIMD run at lone-wave speed while its SIMD-mate takes the rest?
// Development tool.  One workgroup of 8 waves on one CU (2 per SIMD); every wave runs the same FP64 chain mix (two
// independent FMA chains + scalar ops, roughly the issue profile of the solver's serial loops).  Variant A: all waves
// priority 0.  Variant B: waves 0..3 raise their priority to 3.  Prints the time each wave took.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/setprio_probe tools/ubench/setprio_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void probe(long long *ticks, int *simd, double *sink, int iters, int prio_mode, int active_waves) {
    const int wave = threadIdx.x >> 6;
    if (wave >= active_waves) return;
    if (prio_mode == 1 && wave < 4) __builtin_amdgcn_s_setprio(3);
    if (prio_mode == 2 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5, c = 1.0000001, d = 0.25;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a = __builtin_fma(a, c, b);
        d = __builtin_fma(d, c, a * 1e-30);
        a = __builtin_fma(a, 0.999999, -b);
        d = __builtin_fma(d, 0.999999, b);
    }
    const long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        ticks[wave] = t1 - t0;
        simd[wave] = (int)((id >> 4) & 3);
    }
    sink[threadIdx.x] = a + d;
}

int main() {
    long long *d_t; int *d_s; double *d_sink;
    (void)hipMalloc(&d_t, 8 * sizeof(long long)); (void)hipMalloc(&d_s, 8 * sizeof(int)); (void)hipMalloc(&d_sink, 512 * sizeof(double));
    const int iters = 200000;
    for (int active = 4; active <= 8; active += 4)
        for (int mode = 0; mode < 3; ++mode) {
            if (active == 4 && mode > 0) continue;
            (void)hipMemset(d_t, 0, 8 * sizeof(long long));
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d_t, d_s, d_sink, iters, mode, active);
            (void)hipDeviceSynchronize();
            long long t[8]; int s[8];
            (void)hipMemcpy(t, d_t, sizeof(t), hipMemcpyDeviceToHost); (void)hipMemcpy(s, d_s, sizeof(s), hipMemcpyDeviceToHost);
            printf("%d waves, %s:", active, mode == 0 ? "all priority 0" : (mode == 1 ? "waves 0-3 priority 3" : "waves 4-7 priority 3"));
            for (int w = 0; w < active; ++w) printf("  w%d(simd %d) %.0f us", w, s[w], t[w] / 100.0);
            printf("\n");
        }
    return 0;
}
