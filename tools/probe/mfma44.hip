// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand/result lane layout and issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void layout(double *outA, double *outB)
{
    const int l = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        double a = (l == p) ? 1.0 : 0.0, b = 1.0;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        outA[p * 64 + l] = d;
        a = 1.0; b = (l == p) ? 1.0 : 0.0;
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        outB[p * 64 + l] = d;
    }
}
__global__ void timing(double *out, long long *cyc, int n)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
        d4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d4, 0, 0, 0);
        d5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d5, 0, 0, 0);
    }
    long long t1 = clock64();
    out[threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *dA, *dB, *dO; long long *dc;
    hipMalloc(&dA, 64 * 64 * 8); hipMalloc(&dB, 64 * 64 * 8); hipMalloc(&dO, 64 * 8); hipMalloc(&dc, 8);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB);
    std::vector<double> A(4096), B(4096);
    hipMemcpy(A.data(), dA, 4096 * 8, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, 4096 * 8, hipMemcpyDeviceToHost);
    // A lane p feeds output lanes {l}: same block, same row i.  B lane p feeds outputs with the same block, same column j.
    for (int p = 0; p < 64; ++p) { printf("A%02d ->", p); for (int l = 0; l < 64; ++l) if (A[p * 64 + l] != 0) printf(" %d", l); printf("\n"); }
    for (int p = 0; p < 64; ++p) { printf("B%02d ->", p); for (int l = 0; l < 64; ++l) if (B[p * 64 + l] != 0) printf(" %d", l); printf("\n"); }
    const int n = 2000;
    hipLaunchKernelGGL(timing, dim3(1), dim3(64), 0, 0, dO, dc, n);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("clock64 ticks per mfma_f64_4x4x4 (6 independent accumulators): %.2f\n", (double)c / (6.0 * n));
    return 0;
}
