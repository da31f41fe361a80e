// dispatch_skew.hip - how long does the GPU take to start the 4096 one-wave workgroups of a batch?
// Development tool: blocks of 64 threads with the solve kernel's footprint (dynamic LDS, a register-count bound) record
// the device clock when they start and spin for a fixed time; prints when the n-th block started relative to the first.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/dispatch_skew tools/ubench/dispatch_skew.hip && tools/ubench/dispatch_skew
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int OCC>
__global__ __launch_bounds__(64, OCC) void probe(long long *start, long long *stop, int *cu, long long spin_ticks) {
    extern __shared__ double smem[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        start[blockIdx.x] = t0;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        cu[blockIdx.x] = (int)id;
    }
    smem[threadIdx.x] = (double)t0;
    while (wall_clock64() - t0 < spin_ticks) {}
    if (threadIdx.x == 0) stop[blockIdx.x] = wall_clock64();
}

template <int OCC>
static void run(int B, size_t lds, double spin_us) {
    long long *d_start, *d_stop;
    int *d_cu;
    hipMalloc(&d_start, B * sizeof(long long));
    hipMalloc(&d_stop, B * sizeof(long long));
    hipMalloc(&d_cu, B * sizeof(int));
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long long ticks = (long long)(spin_us * 100.0);   // wall_clock64 runs at 100 MHz
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<OCC>, dim3(B), dim3(64), lds, 0, d_start, d_stop, d_cu, ticks);
        hipDeviceSynchronize();
    }
    std::vector<long long> s(B), e(B);
    hipMemcpy(s.data(), d_start, B * sizeof(long long), hipMemcpyDeviceToHost);
    hipMemcpy(e.data(), d_stop, B * sizeof(long long), hipMemcpyDeviceToHost);
    const long long t0 = *std::min_element(s.begin(), s.end());
    std::vector<long long> sorted(s);
    std::sort(sorted.begin(), sorted.end());
    printf("occupancy bound %d, LDS %zu B, %d blocks, spin %.0f us: start of block (by rank) 1 %%: %.1f us, 25 %%: %.1f, 50 %%: %.1f, "
           "75 %%: %.1f, 99 %%: %.1f, last: %.1f us; last stop %.1f us\n",
           OCC, lds, B, spin_us, (sorted[B / 100] - t0) / 100.0, (sorted[B / 4] - t0) / 100.0, (sorted[B / 2] - t0) / 100.0,
           (sorted[3 * B / 4] - t0) / 100.0, (sorted[B * 99 / 100] - t0) / 100.0, (sorted[B - 1] - t0) / 100.0,
           (*std::max_element(e.begin(), e.end()) - t0) / 100.0);
    // by block index: mean start of each quarter of the grid
    for (int q = 0; q < 4; ++q) {
        double m = 0;
        for (int i = q * B / 4; i < (q + 1) * B / 4; ++i) m += (s[i] - t0) / 100.0;
        printf("   blocks %d..%d start on average %.1f us after the first\n", q * B / 4, (q + 1) * B / 4 - 1, m / (B / 4));
    }
    hipFree(d_start); hipFree(d_stop); hipFree(d_cu);
}

int main() {
    run<4>(4096, 10152, 2000.0);
    run<4>(4096, 10152, 200.0);
    run<3>(4096, 12888, 2000.0);
    run<4>(3072, 10152, 2000.0);
    run<4>(4096, 1024, 2000.0);
    run<4>(65536, 10152, 100.0);
    return 0;
}
