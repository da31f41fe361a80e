// mfma_f64_probe.hip - lane layout of v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4, one f64 of A, B, C/D per lane) and the
// cost of a dependent MFMA chain, discovered with one-hot operands.  Development tool.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/ubench/mfma_f64_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(const double *a_in, const double *b_in, double *d_out) {
    const int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], 0.0, 0, 0, 0);
    d_out[l] = d;
}

__global__ void chain(double *out, int n, long long *cycles) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, b = 1.0 - 1e-3 * l, d = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
        a = d * 1e-3;   // dependent: next A operand comes from the result
    }
    long long t1 = __builtin_readcyclecounter();
    out[l] = d;
    if (l == 0) cycles[0] = t1 - t0;
}

__global__ void chain_fma(double *out, int n, long long *cycles) {
    const int l = threadIdx.x;
    double a = 1.0 + 1e-3 * l, d = 0.5;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) d = __builtin_fma(d, a, 1e-3);
    long long t1 = __builtin_readcyclecounter();
    out[l] = d;
    if (l == 0) cycles[0] = t1 - t0;
}

__global__ void chain_bperm(double *out, int n, long long *cycles) {
    const int l = threadIdx.x;
    double d = 1.0 + l;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) d = __shfl(d, (l + 17) & 63) + 1.0;
    long long t1 = __builtin_readcyclecounter();
    out[l] = d;
    if (l == 0) cycles[0] = t1 - t0;
}

__global__ void chain_lds(double *out, int n, long long *cycles) {
    __shared__ double buf[64];
    const int l = threadIdx.x;
    double d = 1.0 + l;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        buf[l] = d;
        __syncthreads();
        d = buf[(l + 17) & 63] + 1.0;
        __syncthreads();
    }
    long long t1 = __builtin_readcyclecounter();
    out[l] = d;
    if (l == 0) cycles[0] = t1 - t0;
}

int main() {
    double *a, *b, *d;
    long long *cyc;
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&d, 64 * 8); hipMalloc(&cyc, 8);
    std::vector<double> ha(64), hb(64), hd(64);
    // A one-hot at lane la, B all ones: the lanes with D != 0 form the row that A element feeds
    printf("A one-hot (B = 1): lanes of D that see A[la]\n");
    for (int la = 0; la < 64; ++la) {
        for (int i = 0; i < 64; ++i) { ha[i] = i == la ? 1.0 : 0.0; hb[i] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("  la=%2d ->", la);
        for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) printf(" %d", i);
        printf("\n");
    }
    printf("B one-hot (A = 1): lanes of D that see B[lb]\n");
    for (int lb = 0; lb < 64; ++lb) {
        for (int i = 0; i < 64; ++i) { hb[i] = i == lb ? 1.0 : 0.0; ha[i] = 1.0; }
        hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
        printf("  lb=%2d ->", lb);
        for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) printf(" %d", i);
        printf("\n");
    }
    // pairing: A one-hot at la and B one-hot at lb contribute iff they share k; print which (la, lb) hit D lane 0..3
    printf("k pairing: for D lane 0, the (la, lb) pairs that contribute\n ");
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            for (int i = 0; i < 64; ++i) { ha[i] = i == la ? 1.0 : 0.0; hb[i] = i == lb ? 1.0 : 0.0; }
            hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a, b, d);
            hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost);
            if (hd[0] != 0.0) printf(" (%d,%d)", la, lb);
            if (hd[21] != 0.0) printf(" [21:(%d,%d)]", la, lb);
        }
    printf("\n");
    const int n = 2000;
    long long c;
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent mfma_f64_4x4x4 + mul chain: %.1f cycles per step\n", (double)c / n);
    hipLaunchKernelGGL(chain_fma, dim3(1), dim3(64), 0, 0, d, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent v_fma_f64 chain: %.1f cycles per step\n", (double)c / n);
    hipLaunchKernelGGL(chain_bperm, dim3(1), dim3(64), 0, 0, d, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent shfl(double)+add chain: %.1f cycles per step\n", (double)c / n);
    hipLaunchKernelGGL(chain_lds, dim3(1), dim3(64), 0, 0, d, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent LDS write/barrier/read chain: %.1f cycles per step\n", (double)c / n);
    return 0;
}
