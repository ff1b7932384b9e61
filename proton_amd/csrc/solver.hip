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

// ---- the same solver on a ROW-PARTITIONED system (several GPUs: the face-only condensed system is assembled by rows,
// pa_condensed_csr_fill; every rank solves where it assembled).  This rank holds rows [row_begin, row_end) of the global
// matrix in CSR with GLOBAL column indices.  The search direction lives in a window  [row_begin - need_lo, row_end + need_hi)
// -- the columns this rank's rows read; the matrix is symmetric and banded by the mesh's row structure, so they belong to the
// two neighbouring ranks -- whose two ends are refreshed from the neighbours once per iteration; the dot products are local
// partial sums added over the ranks.  The transport is three callbacks (RCCL: pa_comm_cg_transport; host-staged gloo in the
// tests); without one the call is the one-rank solver above with its recurrences in the same order.
struct CgTransport {
    void *user;
    int (*allreduce_sum)(void *user, double *vals, int n);
    int (*halo)(void *user, const double *send_lo, size_t n_send_lo, const double *send_hi, size_t n_send_hi, double *recv_lo,
                size_t n_recv_lo, double *recv_hi, size_t n_recv_hi, void *stream);
    int (*neighbour_counts)(void *user, int64_t need_lo, int64_t need_hi, int64_t *give_lo, int64_t *give_hi);
};

// smallest and largest column index of the local rows (one block per 256 entries, atomics on two words)
__global__ __launch_bounds__(RB) void cg_col_range_kernel(size_t nnz, const int32_t *colind, int *minmax)
{
    const size_t k = (size_t)blockIdx.x * RB + threadIdx.x;
    if (k >= nnz) return;
    atomicMin(&minmax[0], colind[k]);
    atomicMax(&minmax[1], colind[k]);
}

__global__ __launch_bounds__(RB) void cg_inv_diag_rows_kernel(size_t n, int64_t row_begin, const int64_t *rowptr, const int32_t *colind,
                                                              const double *values, double *iA)
{
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    double d = 0.0;
    for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if ((int64_t)colind[k] == row_begin + (int64_t)i) d = values[k];
    iA[i] = 1.0 / d;
}

// y = A_local dwin (columns shifted into the window) and the per-block partial of d_local . y
__global__ __launch_bounds__(RB) void cg_spmv_rows_kernel(size_t n, int64_t col0, size_t own0, const int64_t *rowptr, const int32_t *colind,
                                                          const double *values, const double *dwin, double *y, double *part_dy)
{
    __shared__ double sh[RB / 64];
    const size_t row = ((size_t)blockIdx.x * RB + threadIdx.x) / ROW_LANES;
    const int sub = threadIdx.x % ROW_LANES;
    double s = 0.0;
    if (row < n)
        for (int64_t k = rowptr[row] + sub; k < rowptr[row + 1]; k += ROW_LANES) s += values[k] * dwin[(int64_t)colind[k] - col0];
#pragma unroll
    for (int o = ROW_LANES / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, ROW_LANES);
    double dyp = 0.0;
    if (row < n && sub == 0) { y[row] = s; dyp = dwin[own0 + row] * s; }
    const double t = block_sum(dyp, sh);
    if (threadIdx.x == 0) part_dy[blockIdx.x] = t;
}

// sums of up to two arrays of per-block partials -> out[0], out[1]
__global__ __launch_bounds__(RB) void cg_sum_kernel(size_t nparts, const double *part_a, const double *part_b, double *out)
{
    __shared__ double sh[RB / 64];
    double a = 0.0, b = 0.0;
    for (size_t i = threadIdx.x; i < nparts; i += RB) {
        a += part_a[i];
        if (part_b) b += part_b[i];
    }
    const double sa = block_sum(a, sh), sb = block_sum(b, sh);
    if (threadIdx.x == 0) { out[0] = sa; out[1] = sb; }
}

__global__ __launch_bounds__(RB) void cg_update_rows_kernel(size_t n, double alpha, const double *iA, int precond, const double *d,
                                                            const double *y, double *x, double *r, double *part_a, double *part_b)
{
    __shared__ double sh[RB / 64];
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
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

__global__ __launch_bounds__(RB) void cg_direction_rows_kernel(size_t n, double beta, const double *iA, int precond, const double *r, double *d)
{
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    d[i] = (precond ? iA[i] * r[i] : r[i]) + beta * d[i];
}

// status: 0 ok; 1 a transport callback failed on THIS rank; 2 this rank's rows read a column no neighbour range covers (or, without a
// transport, beyond its own range); 3 another rank failed (its status says why); 4 a HIP error on this rank.
// EVERY exit is collective: a rank that fails locally -- a callback, a HIP call, the range check -- does not leave the others inside a
// reduction or a neighbour exchange.  Its failure is a flag that rides on the next all-reduce (an extra addend behind the dot products),
// every rank sees the sum and all of them leave together, at the same point.  (What cannot be covered: the all-reduce itself failing
// on one rank only -- then there is nothing left to tell the others with.)
hipError_t conjugated_gradient_rows(hipStream_t stream, const CgTransport *tp, int64_t row_begin, int64_t row_end, const int64_t *rowptr,
                                    const int32_t *colind, const double *values, const double *b, double *x,
                                    double convergence_threshold, double divergence_threshold, size_t max_iter, int precond,
                                    int *exit_reason, size_t *iterations, double *relative_residual, int *transport_status)
{
    const size_t n = (size_t)(row_end - row_begin);
    double *r = nullptr, *dwin = nullptr, *y = nullptr, *iA = nullptr, *pa_ = nullptr, *pb_ = nullptr, *sums = nullptr;
    int *mm = nullptr;
    const size_t nn = n ? n : 1;
    const unsigned gv = (unsigned)((nn + RB - 1) / RB);
    const unsigned gs = (unsigned)((nn * ROW_LANES + RB - 1) / RB);
    const size_t nparts = gs > gv ? gs : gv;
    auto cleanup = [&]() {
        (void)hipFree(r); (void)hipFree(dwin); (void)hipFree(y); (void)hipFree(iA); (void)hipFree(pa_); (void)hipFree(pb_);
        (void)hipFree(sums); (void)hipFree(mm);
    };
    if (transport_status) *transport_status = 0;
    int fail = 0;                          // this rank's status (see above); 0 while everything went well
    hipError_t herr = hipSuccess;
    auto hip_ok = [&](hipError_t e) -> bool { if (e != hipSuccess && !fail) { fail = 4; herr = e; } return e == hipSuccess; };
    auto leave = [&](int status) -> hipError_t {
        if (transport_status) *transport_status = status;
        cleanup();
        return status == 4 ? herr : hipSuccess;
    };
    // all ranks agree on whether anybody failed: the flag summed over the ranks (one rank: the flag itself)
    auto agree = [&]() -> int {
        double f = fail ? 1.0 : 0.0;
        if (tp && tp->allreduce_sum(tp->user, &f, 1) != 0) return fail ? fail : 1;
        return f > 0.0 ? (fail ? fail : 3) : 0;
    };
    // the columns the local rows read
    int64_t need_lo = 0, need_hi = 0;
    {
        int64_t nnz = 0;
        if (n) {
            int64_t ends[2] = {0, 0};
            if (hip_ok(hipMemcpyAsync(&ends[0], rowptr, 8, hipMemcpyDeviceToHost, stream)) &&
                hip_ok(hipMemcpyAsync(&ends[1], rowptr + n, 8, hipMemcpyDeviceToHost, stream)) && hip_ok(hipStreamSynchronize(stream)))
                nnz = ends[1] - ends[0];
        }
        int h[2] = {0x7fffffff, -1};
        if (!fail && hip_ok(hipMalloc((void **)&mm, 8)) && hip_ok(hipMemcpyAsync(mm, h, 8, hipMemcpyHostToDevice, stream))) {
            if (nnz > 0) hipLaunchKernelGGL(cg_col_range_kernel, dim3((unsigned)((nnz + RB - 1) / RB)), dim3(RB), 0, stream, (size_t)nnz, colind, mm);
            if (hip_ok(hipMemcpyAsync(h, mm, 8, hipMemcpyDeviceToHost, stream)) && hip_ok(hipStreamSynchronize(stream)) && nnz > 0) {
                need_lo = h[0] < row_begin ? row_begin - h[0] : 0;
                need_hi = (int64_t)h[1] + 1 > row_end ? (int64_t)h[1] + 1 - row_end : 0;
            }
        }
    }
    int64_t give_lo = 0, give_hi = 0;      // what the neighbours below / above read of MY range (its first / last entries)
    if (tp) {
        // (a rank that already failed still answers its neighbours -- with what it needs, zero -- so that they are not left waiting)
        if (tp->neighbour_counts(tp->user, fail ? 0 : need_lo, fail ? 0 : need_hi, &give_lo, &give_hi) != 0 && !fail) fail = 1;
    } else if (need_lo || need_hi) fail = 2;
    if (!fail && (give_lo > (int64_t)n || give_hi > (int64_t)n)) fail = 2;
    const size_t nwin = n + (size_t)need_lo + (size_t)need_hi;
    if (!fail) {
        (void)(hip_ok(hipMalloc((void **)&r, nn * 8)) && hip_ok(hipMalloc((void **)&dwin, (nwin ? nwin : 1) * 8)) && hip_ok(hipMalloc((void **)&y, nn * 8)) &&
               hip_ok(hipMalloc((void **)&iA, nn * 8)) && hip_ok(hipMalloc((void **)&pa_, nparts * 8)) && hip_ok(hipMalloc((void **)&pb_, nparts * 8)) &&
               hip_ok(hipMalloc((void **)&sums, 16)) && hip_ok(hipMemsetAsync(dwin, 0, (nwin ? nwin : 1) * 8, stream)));
    }
    {
        const int st = agree();            // nobody starts iterating unless everybody can
        if (st) return leave(st);
    }
    double *d = dwin + need_lo;            // the owned part of the window
    double hs[3];
    // local sums -> host -> ranks, the failure flag behind them.  Returns 0, or the status all ranks leave with.
    auto reduce2 = [&](size_t np, const double *pa2, const double *pb2, int nvals) -> int {
        hs[0] = hs[1] = 0.0;
        if (!fail) {
            hipLaunchKernelGGL(cg_sum_kernel, dim3(1), dim3(RB), 0, stream, np, pa2, pb2, sums);
            (void)(hip_ok(hipMemcpyAsync(hs, sums, 16, hipMemcpyDeviceToHost, stream)) && hip_ok(hipStreamSynchronize(stream)));
        }
        hs[nvals] = fail ? 1.0 : 0.0;
        if (tp && tp->allreduce_sum(tp->user, hs, nvals + 1) != 0) return fail ? fail : 1;
        return hs[nvals] > 0.0 ? (fail ? fail : 3) : 0;
    };
    auto refresh = [&]() {                 // the two ends of the window from the neighbours, my ends to them
        if (!tp || fail) return;           // (a failed rank does not post: it tells the others at the reduction that follows the product)
        if (tp->halo(tp->user, d, (size_t)give_lo, d + (n - (size_t)give_hi), (size_t)give_hi, dwin, (size_t)need_lo, d + n, (size_t)need_hi,
                     (void *)stream) != 0)
            fail = 1;
    };
    size_t iter = 0;
    int reason = 2;
    double rr = 0.0;
    {
        if (n) hipLaunchKernelGGL(cg_inv_diag_rows_kernel, dim3(gv), dim3(RB), 0, stream, n, row_begin, rowptr, colind, values, iA);
        // r = b, d = M^-1 r (x = 0), partials of r.r and r.M^-1 r       solver_cg.hpp:73-84
        hipLaunchKernelGGL(cg_init_kernel, dim3(gv), dim3(RB), 0, stream, n, b, iA, precond, x, r, d, pa_, pb_);
        int st = reduce2(gv, pa_, pb_, 2);
        if (st) return leave(st);
        const double nr0 = sqrt(hs[0]);
        double rho = hs[1];
        if (!(nr0 > 0.0)) { reason = 0; }
        else
            for (;;) {
                refresh();
                if (!fail)
                    hipLaunchKernelGGL(cg_spmv_rows_kernel, dim3(gs), dim3(RB), 0, stream, n, row_begin - need_lo, (size_t)need_lo, rowptr, colind,
                                       values, dwin, y, pa_);                                                        // :99
                st = reduce2(gs, pa_, (const double *)nullptr, 1);
                if (st) return leave(st);
                const double alpha = rho / hs[0];                                                                    // :101-102
                hipLaunchKernelGGL(cg_update_rows_kernel, dim3(gv), dim3(RB), 0, stream, n, alpha, iA, precond, d, y, x, r, pa_, pb_);   // :103-105
                st = reduce2(gv, pa_, pb_, 2);
                if (st) return leave(st);
                rr = sqrt(hs[0]) / nr0;
                if (rr < convergence_threshold) { reason = 0; break; }      // :107-110
                if (iter > max_iter) { reason = 2; break; }                  // :112-115
                if (rr > divergence_threshold) { reason = 1; break; }        // :117-120
                if (!(rr == rr)) { reason = 1; break; }
                const double beta = hs[1] / rho;
                rho = hs[1];
                hipLaunchKernelGGL(cg_direction_rows_kernel, dim3(gv), dim3(RB), 0, stream, n, beta, iA, precond, r, d);   // :126-128
                iter++;
            }
    }
    (void)(hip_ok(hipGetLastError()) && hip_ok(hipStreamSynchronize(stream)));
    {
        const int st = agree();            // a rank whose last launches failed says so before anybody reports success
        if (st) return leave(st);
    }
    cleanup();
    if (exit_reason) *exit_reason = reason;
    if (iterations) *iterations = iter;
    if (relative_residual) *relative_residual = rr;
    return hipSuccess;
}

}  // namespace pa
