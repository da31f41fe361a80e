// placement_probe.hip - where do the one-wave workgroups of a small batch land?  (development tool)
// A kernel with the latency build's footprint (64 threads, occupancy 2, 12 KB of LDS) records HW_ID / XCC_ID per workgroup;
// the host counts distinct SIMDs and CUs for B = 256 (config 4), 1024 (config 2), 2048, 4096.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/placement_probe tools/ubench/placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <map>
#include <vector>
__global__ __launch_bounds__(64, 2) void probe(unsigned *out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);     // stay resident while the others arrive
    lds[threadIdx.x] = a;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    if (a == 12345.678) out[0] = 0;
}
int main() {
    unsigned *d;
    (void)hipMalloc(&d, 8192 * 8);
    for (int B : {256, 1024, 2048, 4096}) {
        hipLaunchKernelGGL(probe, dim3(B), dim3(64), 12 * 1024, 0, d, 200000);
        std::vector<unsigned> h(2 * B);
        (void)hipMemcpy(h.data(), d, 8 * B, hipMemcpyDeviceToHost);
        std::set<unsigned long long> simds, cus;
        std::map<unsigned long long, int> per_simd;
        for (int b = 0; b < B; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned long long cuid = ((unsigned long long)xcc << 12) | (se << 8) | (sh << 4) | cu;
            cus.insert(cuid);
            simds.insert(cuid * 4 + simd);
            per_simd[cuid * 4 + simd]++;
        }
        int mx = 0;
        for (auto &kv : per_simd) mx = kv.second > mx ? kv.second : mx;
        printf("B = %4d: %zu distinct CUs, %zu distinct SIMDs, at most %d workgroups on one SIMD\n", B, cus.size(), simds.size(), mx);
    }
    return 0;
}
