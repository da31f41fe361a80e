// dpp_chain_probe.hip - dependent-issue cost of the building blocks of the row-cooperative rollout for a lone wave on MI355X:
// v_mov_b64_dpp row_newbcast feeding an FMA, selects, LDS round trips.  Development tool (round 5).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dpp_chain_probe tools/ubench/dpp_chain_probe.hip && tools/ubench/dpp_chain_probe
#include <hip/hip_runtime.h>

#include <cstdio>

template <int J>
__device__ __forceinline__ double bc(double v) {
    return __longlong_as_double(__builtin_amdgcn_update_dpp((long long)0, __double_as_longlong(v), 0x150 + J, 0xf, 0xf, true));
}
template <int OP>
__global__ void chain(double *out, long long *cycles, double seed) {
    __shared__ double lds[128];
    double d = seed + 1e-9 * threadIdx.x, e = seed * 0.5, a = 1.0000001, b = 1e-9, s = seed + 0.25 * threadIdx.x;
    lds[threadIdx.x] = s;
    lds[64 + threadIdx.x] = s;
    __syncthreads();
    asm volatile("" : "+v"(d), "+v"(e), "+v"(s));
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 128; ++i) {
        if (OP == 0) d = __builtin_fma(d, a, b);                                  // FMA -> FMA
        if (OP == 1) d = __builtin_fma(bc<3>(d), a, b);                           // FMA -> bcast -> FMA
        if (OP == 2) d = __builtin_fma(bc<3>(s), a, d);                           // bcast of a ready value, FMA chain
        if (OP == 3) { d = __builtin_fma(d, a, b); e = __builtin_fma(bc<5>(e), a, b); }   // two chains, one with bcasts
        if (OP == 4) d = (threadIdx.x & 1) ? __builtin_fma(d, a, b) : d + b;      // select of two results per lane
        if (OP == 5) { lds[threadIdx.x] = d; asm volatile("" ::: "memory"); d = lds[threadIdx.x ^ 1] + b; asm volatile("" ::: "memory"); }   // LDS round trip
        if (OP == 6) d = __builtin_fmax(__builtin_fmin(d * a, 1e300), -1e300);    // mul -> min -> max
        if (OP == 7) { d = __builtin_fma(d, a, b); e = __builtin_fma(e, a, b); }  // two independent FMA chains
        if (OP == 8) { d = __builtin_fma(d, a, b); e = __builtin_fma(e, a, b); s = __builtin_fma(s, a, b); }   // three
    }
    asm volatile("" ::"v"(d), "v"(e), "v"(s));
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = d + e + s;
    if (threadIdx.x == 0) cycles[0] = t1 - t0;
}
int main() {
    double *out;
    long long *cyc, c;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    const char *names[] = {"FMA -> FMA", "FMA -> row_newbcast -> FMA", "bcast(ready) + FMA chain", "FMA chain || bcast-FMA chain (per pair)",
                           "per-lane select of FMA / add", "LDS store -> load round trip", "mul -> min -> max", "two independent FMA chains (per pair)",
                           "three independent FMA chains (per triple)"};
#define RUN(OP)                                                                  \
    for (int rep = 0; rep < 2; ++rep) {                                          \
        hipLaunchKernelGGL(chain<OP>, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);  \
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                            \
    }                                                                            \
    printf("%-48s %.1f counter ticks per step\n", names[OP], (double)c / 128);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    // the counter's rate against the shader clock: a chain of 4-cycle-issue instructions of known count
    return 0;
}
