// dd_check.hip -- the double-double primitives of proton_amd/csrc/dd_arith.hpp ON THE DEVICE checked in exact rational arithmetic (tools/probe/dd_check.py)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I proton_amd/csrc -o tools/probe/dd_check tools/probe/dd_check.hip && tools/probe/dd_check && python tools/probe/dd_check.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>
#include "dd_arith.hpp"
using namespace pa;
__global__ void k(int n, const double *ah, const double *al, const double *bh, const double *bl, const double *d, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const dd a{ah[i], al[i]}, b{bh[i], bl[i]};
    dd r[7] = {dd_add(a, b), dd_mul(a, b), dd_mul_d(a, d[i]), dd_rsqrt(dd{fabs(ah[i]) + 0.1, al[i]}), dd_sub(a, b), two_prod(ah[i], bh[i]),
               dd_rsqrt_1(dd{fabs(ah[i]) + 0.1, al[i]})};
    for (int q = 0; q < 7; ++q) { out[(size_t)(2 * q) * n + i] = r[q].hi; out[(size_t)(2 * q + 1) * n + i] = r[q].lo; }
}
int main()
{
    const int n = 1 << 16;
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(-1, 1);
    std::vector<double> ah(n), al(n), bh(n), bl(n), d(n), out(14 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const double x = u(g) * 1e3, y = u(g);
        ah[i] = x; al[i] = x * 1e-17 * u(g); bh[i] = (i % 3 == 0) ? -x * (1 + 1e-9 * u(g)) : y; bl[i] = bh[i] * 1e-17 * u(g); d[i] = u(g);
        // normalise (hi, lo) pairs
        double s = ah[i] + al[i]; al[i] = al[i] - (s - ah[i]); ah[i] = s;
        s = bh[i] + bl[i]; bl[i] = bl[i] - (s - bh[i]); bh[i] = s;
    }
    double *p[6];
    for (int q = 0; q < 5; ++q) (void)hipMalloc((void **)&p[q], n * 8);
    (void)hipMalloc((void **)&p[5], 14 * (size_t)n * 8);
    const double *src[5] = {ah.data(), al.data(), bh.data(), bl.data(), d.data()};
    for (int q = 0; q < 5; ++q) (void)hipMemcpy(p[q], src[q], n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, p[0], p[1], p[2], p[3], p[4], p[5]);
    (void)hipMemcpy(out.data(), p[5], 14 * (size_t)n * 8, hipMemcpyDeviceToHost);
    // inputs and device results to a file; tools/probe/dd_check.py verifies them in exact rational arithmetic
    FILE *f = std::fopen("gpurun_out/dd_check.bin", "wb");
    if (!f) return 1;
    for (int q = 0; q < 5; ++q) std::fwrite(src[q], 8, n, f);
    std::fwrite(out.data(), 8, 14 * (size_t)n, f);
    std::fclose(f);
    return 0;
}
