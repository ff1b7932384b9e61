// solver.hip -- the reference's Jacobi-preconditioned conjugate gradient
// (src/core/core_bits/solver_cg.hpp:63-144, used by run_cuthho_interface cuthho_square.cpp:1737-1743
// and offered by convergence_test) on the device, over the CSR matrix pa_csr_from_triplets builds.
// Same recurrences, exit tests and exit order as the reference; the dot products are block-tree
// reductions (deterministic run to run), not Eigen's sequential sums.
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pa {

constexpr int RB = 256;            // threads per block of the vector kernels
constexpr int ROW_LANES = 16;      // lanes that share a row in the SpMV (HHO rows hold 20-130 entries)

struct CgScalars {                 // device-resident scalars of the iteration
    double rho, dy, nr2, rho_new, alpha, beta;
};

__device__ inline double block_sum(double v, double *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < RB / 64; ++w) s += sh[w];
    __syncthreads();
    return s;
}

// iA = 1 / diag(A)   (solver_cg.hpp:77-80)
__global__ __launch_bounds__(RB) void cg_inv_diag_kernel(size_t n, const int64_t *rowptr, const int32_t *colind, const double *values,
                                                         double *iA)
{
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    double d = 0.0;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if ((size_t)colind[k] == i) d = values[k];
    iA[i] = 1.0 / d;
}

// y = A d and the per-block partial of d . y
__global__ __launch_bounds__(RB) void cg_spmv_kernel(size_t n, const int64_t *rowptr, const int32_t *colind, const double *values,
                                                     const double *d, double *y, double *part_dy)
{
    __shared__ double sh[RB / 64];
    const size_t row = ((size_t)blockIdx.x * RB + threadIdx.x) / ROW_LANES;
    const int sub = threadIdx.x % ROW_LANES;
    double s = 0.0;
    if (row < n)
        for (int64_t k = rowptr[row] + sub; k < rowptr[row + 1]; k += ROW_LANES) s += values[k] * d[colind[k]];
#pragma unroll
    for (int o = ROW_LANES / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, ROW_LANES);
    double dyp = 0.0;
    if (row < n && sub == 0) { y[row] = s; dyp = d[row] * s; }
    const double t = block_sum(dyp, sh);
    if (threadIdx.x == 0) part_dy[blockIdx.x] = t;
}

// r = b - y (first residual, x = 0 means y = 0 is never formed: r = b), d = M^-1 r, partials of r.r and r.M^-1 r
__global__ __launch_bounds__(RB) void cg_init_kernel(size_t n, const double *b, const double *iA, int precond, double *x, double *r,
                                                     double *d, double *part_a, double *part_b)
{
    __shared__ double sh[RB / 64];
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    double rr = 0.0, rz = 0.0;
    if (i < n) {
        const double ri = b[i];
        const double zi = precond ? iA[i] * ri : ri;
        x[i] = 0.0; r[i] = ri; d[i] = zi;
        rr = ri * ri; rz = ri * zi;
    }
    const double t0 = block_sum(rr, sh), t1 = block_sum(rz, sh);
    if (threadIdx.x == 0) { part_a[blockIdx.x] = t0; part_b[blockIdx.x] = t1; }
}

// x += alpha d, r -= alpha y; partials of r.r and r.M^-1 r   (solver_cg.hpp:103-109, 126-127)
__global__ __launch_bounds__(RB) void cg_update_kernel(size_t n, const CgScalars *sc, const double *iA, int precond, const double *d,
                                                       const double *y, double *x, double *r, double *part_a, double *part_b)
{
    __shared__ double sh[RB / 64];
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    const double alpha = sc->alpha;
    double rr = 0.0, rz = 0.0;
    if (i < n) {
        x[i] += alpha * d[i];
        const double ri = r[i] - alpha * y[i];
        r[i] = ri;
        rr = ri * ri; rz = ri * (precond ? iA[i] * ri : ri);
    }
    const double t0 = block_sum(rr, sh), t1 = block_sum(rz, sh);
    if (threadIdx.x == 0) { part_a[blockIdx.x] = t0; part_b[blockIdx.x] = t1; }
}

// d = M^-1 r + beta d   (solver_cg.hpp:128)
__global__ __launch_bounds__(RB) void cg_direction_kernel(size_t n, const CgScalars *sc, const double *iA, int precond, const double *r,
                                                          double *d)
{
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    d[i] = (precond ? iA[i] * r[i] : r[i]) + sc->beta * d[i];
}

// one block: sums up to two arrays of per-block partials, then the scalar algebra of the step
//   mode 0 (after init):   rho = sum(b) [r.M^-1 r], nr2 = sum(a)
//   mode 1 (after spmv):   dy = sum(a); alpha = rho / dy
//   mode 2 (after update): nr2 = sum(a), rho_new = sum(b); beta = rho_new / rho; rho = rho_new
__global__ __launch_bounds__(RB) void cg_reduce_kernel(int mode, size_t nparts, const double *part_a, const double *part_b, CgScalars *sc)
{
    __shared__ double sh[RB / 64];
    double a = 0.0, b = 0.0;
    for (size_t i = threadIdx.x; i < nparts; i += RB) {
        a += part_a[i];
        if (part_b) b += part_b[i];
    }
    const double sa = block_sum(a, sh), sb = block_sum(b, sh);
    if (threadIdx.x == 0) {
        if (mode == 0) { sc->nr2 = sa; sc->rho = sb; }
        else if (mode == 1) { sc->dy = sa; sc->alpha = sc->rho / sa; }
        else { sc->nr2 = sa; sc->rho_new = sb; sc->beta = sb / sc->rho; sc->rho = sb; }
    }
}

// exit_reason: 0 converged, 1 diverged, 2 max_iter reached (cg_exit_reason, solver_cg.hpp:38-43)
hipError_t conjugated_gradient(hipStream_t stream, size_t n, const int64_t *rowptr, const int32_t *colind, const double *values,
                               const double *b, double *x, double convergence_threshold, double divergence_threshold,
                               size_t max_iter, int precond, int *exit_reason, size_t *iterations, double *relative_residual)
{
    hipError_t e = hipSuccess;
    double *r = nullptr, *d = nullptr, *y = nullptr, *iA = nullptr, *pa_ = nullptr, *pb_ = nullptr;
    CgScalars *sc = nullptr;
    const size_t nn = n ? n : 1;
    const unsigned gv = (unsigned)((nn + RB - 1) / RB);
    const unsigned gs = (unsigned)((nn * ROW_LANES + RB - 1) / RB);
    const size_t nparts = gs > gv ? gs : gv;
    auto cleanup = [&]() {
        (void)hipFree(r); (void)hipFree(d); (void)hipFree(y); (void)hipFree(iA); (void)hipFree(pa_); (void)hipFree(pb_); (void)hipFree(sc);
    };
#define CG_TRY(call) do { e = (call); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    CG_TRY(hipMalloc((void **)&r, nn * 8)); CG_TRY(hipMalloc((void **)&d, nn * 8)); CG_TRY(hipMalloc((void **)&y, nn * 8));
    CG_TRY(hipMalloc((void **)&iA, nn * 8)); CG_TRY(hipMalloc((void **)&pa_, nparts * 8)); CG_TRY(hipMalloc((void **)&pb_, nparts * 8));
    CG_TRY(hipMalloc((void **)&sc, sizeof(CgScalars)));
    CgScalars h{};
    size_t iter = 0;
    int reason = 2;
    double rr = 0.0;
    if (n) {
        hipLaunchKernelGGL(cg_inv_diag_kernel, dim3(gv), dim3(RB), 0, stream, n, rowptr, colind, values, iA);
        hipLaunchKernelGGL(cg_init_kernel, dim3(gv), dim3(RB), 0, stream, n, b, iA, precond, x, r, d, pa_, pb_);      // :82-84
        hipLaunchKernelGGL(cg_reduce_kernel, dim3(1), dim3(RB), 0, stream, 0, (size_t)gv, pa_, pb_, sc);
        CG_TRY(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, stream));
        CG_TRY(hipStreamSynchronize(stream));
        const double nr0 = sqrt(h.nr2);
        if (!(nr0 > 0.0)) { reason = 0; }                   // b = 0: x = 0 is the solution (the reference would divide by zero)
        else
            for (;;) {
                hipLaunchKernelGGL(cg_spmv_kernel, dim3(gs), dim3(RB), 0, stream, n, rowptr, colind, values, d, y, pa_);   // :99
                hipLaunchKernelGGL(cg_reduce_kernel, dim3(1), dim3(RB), 0, stream, 1, (size_t)gs, pa_, (const double *)nullptr, sc);   // :101-102
                hipLaunchKernelGGL(cg_update_kernel, dim3(gv), dim3(RB), 0, stream, n, sc, iA, precond, d, y, x, r, pa_, pb_);       // :103-105
                hipLaunchKernelGGL(cg_reduce_kernel, dim3(1), dim3(RB), 0, stream, 2, (size_t)gv, pa_, pb_, sc);
                CG_TRY(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, stream));
                CG_TRY(hipStreamSynchronize(stream));
                rr = sqrt(h.nr2) / nr0;
                if (rr < convergence_threshold) { reason = 0; break; }      // :107-110
                if (iter > max_iter) { reason = 2; break; }                  // :112-115
                if (rr > divergence_threshold) { reason = 1; break; }        // :117-120
                if (!(rr == rr)) { reason = 1; break; }                      // NaN: stop instead of spinning
                hipLaunchKernelGGL(cg_direction_kernel, dim3(gv), dim3(RB), 0, stream, n, sc, iA, precond, r, d);           // :126-128
                iter++;
            }
    } else reason = 0;
    CG_TRY(hipGetLastError());
#undef CG_TRY
    cleanup();
    if (exit_reason) *exit_reason = reason;
    if (iterations) *iterations = iter;
    if (relative_residual) *relative_residual = rr;
    return hipSuccess;
}

}  // namespace pa
