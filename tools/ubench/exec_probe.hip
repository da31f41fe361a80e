// exec_probe.hip - round 6: does a vector instruction cost less when only part of the wave is active (EXEC)?
// 4 independent v_fma_f64 chains, unrolled, with the lanes [0, ACTIVE) enabled; also v_mov_b64_dpp and a 32-bit op.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kOuter = 256, kIn = 64;
template <int OP>
__global__ void probe(double *out, long long *ticks, int active) {
    double w = 1.0000001, e = 1e-9;
    double b0 = 0.5 + 1e-9 * threadIdx.x, b1 = 0.6, b2 = 0.7, b3 = 0.8;
    int i0 = threadIdx.x, i1 = 3, i2 = 5, i3 = 7;
    asm volatile("" : "+v"(w), "+v"(e), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = wall_clock64();
#pragma unroll 1
        for (int o = 0; o < kOuter; ++o) {
#pragma unroll
            for (int i = 0; i < kIn; ++i) {
                if (OP == 0) { b0 = __builtin_fma(b0, w, e); b1 = __builtin_fma(b1, w, e); b2 = __builtin_fma(b2, w, e); b3 = __builtin_fma(b3, w, e); }
                if (OP == 1) { i0 = i0 * 3 + i1; i1 = i1 * 5 + i2; i2 = i2 * 7 + i3; i3 = i3 * 9 + i0; }
                if (OP == 2) {
                    b0 = __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(b1), 0x150 + 1, 0xf, 0xf, true)) + e;
                    b1 = __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(b0), 0x150 + 2, 0xf, 0xf, true)) + e;
                }
            }
        }
        t1 = wall_clock64();
    }
    out[threadIdx.x] = b0 + b1 + b2 + b3 + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    double *out; long long *tk;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&tk, 8);
    const char *names[3] = {"4 x v_fma_f64", "4 x (v_mul_lo + v_add) int32", "2 x (v_mov_b64_dpp + v_add_f64)"};
    for (int op = 0; op < 3; ++op)
        for (int active : {64, 48, 32, 20, 16, 8, 1}) {
            long long best = 1ll << 60;
            for (int r = 0; r < 3; ++r) {
                if (op == 0) hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, out, tk, active);
                if (op == 1) hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, out, tk, active);
                if (op == 2) hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, out, tk, active);
                long long h; (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
                if (h < best) best = h;
            }
            const double ns = best * 10.0 / (kOuter * kIn);
            printf("%-34s active lanes %2d: %6.2f ns per step = %5.1f cycles\n", names[op], active, ns, ns * 2.4);
        }
    return 0;
}
