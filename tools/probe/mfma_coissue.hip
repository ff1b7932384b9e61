// Does v_mfma_f64_16x16x4_f64 share its issue / execution slots with the vector FP64 pipe on gfx950?  Per iteration one matrix
// instruction and K independent vector instructions (v_fma_f64, or v_add_u32), W wavefronts per SIMD; wall time by HIP events.
// If the time is that of 16 + K vector slots the two share a pipe; if it is max(16, K) they overlap.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_coissue.hip -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int K, int KIND, int NM>
__global__ __launch_bounds__(256) void k(double *out, int n, double seed)
{
    double d[8]; unsigned a[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + 1e-9 * (threadIdx.x + i) * seed; a[i] = threadIdx.x + i; }
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double x = 1.0000001 * seed, dm = 1.0000001;
    const unsigned m = 3;
    for (int it = 0; it < n; ++it) {
        if (NM >= 1) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[j & 7]) : "v"(dm));
            else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j & 7]) : "v"(m));
        }
        if (NM >= 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc1, 0, 0, 0);
    }
    double r = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[3];
    for (int i = 0; i < 8; ++i) r += d[i] + a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int W = 4;
template <int K, int KIND, int NM> float run(double *out, int n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<K, KIND, NM><<<256 * W, 256>>>(out, n / 10, 1.0);
    hipEventRecord(e0); k<K, KIND, NM><<<256 * W, 256>>>(out, n, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int K> void row(double *out, int n)
{
    const double per = 1e6 / n / W;      // ns per iteration per wavefront-slot of a SIMD (W wavefronts share the SIMD)
    std::printf("K = %2d | no mfma: fma %.1f ns, add_u32 %.1f | 1 mfma + K fma %.1f, + K add_u32 %.1f | 2 mfma + K fma %.1f\n", K,
                run<K, 0, 0>(out, n) * per, run<K, 1, 0>(out, n) * per, run<K, 0, 1>(out, n) * per, run<K, 1, 1>(out, n) * per, run<K, 0, 2>(out, n) * per);
}
int main(int argc, char **argv)
{
    if (argc > 1) W = atoi(argv[1]);
    double *out; hipMalloc(&out, 8 * 256 * 256 * 8);
    const int n = 100000;
    std::printf("%d wavefronts per SIMD; ns per loop iteration and SIMD (one iteration of EVERY resident wavefront: divide by nothing)\n", W);
    row<0>(out, n); row<8>(out, n); row<16>(out, n); row<32>(out, n); row<64>(out, n);
    return 0;
}
