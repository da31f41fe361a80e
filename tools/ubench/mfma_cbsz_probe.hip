// mfma_cbsz_probe.hip - does v_mfma_f64_4x4x4f64 honour the A-broadcast fields (cbsz / abid) on gfx950, and with what
// meaning?  A = one-hot block indicator (A_blk = (blk + 1) * I4), B = I4 in every block: D_blk = (source block + 1) * I4,
// so the diagonal of each result block names the block its A operand came from.  Development tool.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cbsz_probe tools/ubench/mfma_cbsz_probe.hip && /tmp/cbsz_probe
#include <hip/hip_runtime.h>

#include <cstdio>

template <int CBSZ, int ABID>
__global__ void probe(double *out) {
    const int l = threadIdx.x, hi = l >> 4, blk = (l >> 2) & 3, lo = l & 3;
    const double a = (lo == hi) ? (double)(blk + 1) : 0.0;   // A_blk[row lo][k hi]
    const double b = (hi == lo) ? 1.0 : 0.0;                 // B_blk[k hi][col lo]
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
}

template <int CBSZ, int ABID>
void run(double *d) {
    double h[64];
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("cbsz %d abid %d: A source block of result blocks 0..3 =", CBSZ, ABID);
    for (int blk = 0; blk < 4; ++blk) printf(" %g", h[16 * 0 + 4 * blk + 0] - 1.0);   // element (0, 0) of the block
    printf("\n");
}

int main() {
    double *d;
    hipMalloc(&d, 64 * sizeof(double));
    run<0, 0>(d);
    run<1, 0>(d); run<1, 1>(d);
    run<2, 0>(d); run<2, 1>(d); run<2, 2>(d); run<2, 3>(d);
    hipFree(d);
    return 0;
}
