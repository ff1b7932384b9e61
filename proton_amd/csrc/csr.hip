// csr.hip -- device-side SparseMatrix::setFromTriplets (src/methods/hho_bits/hho.hpp:451-455,
// :746-750; cuthho_square.cpp:1437-1441): COO triplet slots -> CSR with duplicates summed.
// Slots whose row is negative are the ones the assemblers did not push.
//
// Pipeline: 64-bit keys (row << 32 | col) -> stable LSD radix sort of (key, value) pairs (rocPRIM's
// device radix sort: the one library call of this file; everything else is hand-written) -> heads
// of equal-key runs -> exclusive scan of the head flags (three-pass scan of hho_assembly.hpp) ->
// each head sums its run left to right, i.e. in the original push order (the sort is stable), which
// is the order Eigen sums duplicates in -> row pointers by binary search.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

#include "scan.hpp"

namespace pa {

__global__ __launch_bounds__(256) void csr_keys_kernel(size_t n, const int32_t *rows, const int32_t *cols, uint64_t *keys)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t r = rows[t], c = cols[t];
    keys[t] = (r < 0 || c < 0) ? ~(uint64_t)0 : ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
}

// flag[i] = 1 where a run of equal keys starts (empty slots, key = ~0, never start a run)
__global__ __launch_bounds__(256) void csr_heads_kernel(size_t n, const uint64_t *keys, uint8_t *flag)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t k = keys[t];
    flag[t] = (k != ~(uint64_t)0 && (t == 0 || keys[t - 1] != k)) ? 1 : 0;
}

// position among the heads (B_ct of the scan: exclusive count of set flags) -> one output entry per run
__global__ __launch_bounds__(256) void csr_reduce_kernel(size_t n, const uint64_t *keys, const double *vals, const uint8_t *flag,
                                                         const int32_t *pos, int32_t *colind, double *values, uint64_t *ukeys)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || !flag[t]) return;
    const uint64_t k = keys[t];
    double s = vals[t];
    for (size_t u = t + 1; u < n && keys[u] == k; ++u) s += vals[u];       // push order: the sort is stable
    const size_t o = (size_t)pos[t];
    colind[o] = (int32_t)(uint32_t)(k & 0xffffffffu);
    values[o] = s;
    ukeys[o] = k;
}

// rowptr[r] = first entry whose key >= (r << 32): binary search in the sorted unique keys
__global__ __launch_bounds__(256) void csr_rowptr_kernel(size_t nrows, size_t nnz, const uint64_t *ukeys, int64_t *rowptr)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    const uint64_t target = (uint64_t)r << 32;
    size_t lo = 0, hi = nnz;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (ukeys[mid] < target) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = (int64_t)lo;
}

// returns hipSuccess or the failing call's error; *nnz_out on the host
hipError_t csr_from_triplets(hipStream_t stream, size_t n, const int32_t *d_rows, const int32_t *d_cols, const double *d_vals,
                             size_t nrows, int64_t *d_rowptr, int32_t *d_colind, double *d_values, size_t *nnz_out)
{
    if (n >= ((size_t)1 << 31)) return hipErrorInvalidValue;          // positions are int32 (the scan's tables)
    hipError_t e;
    uint64_t *keys = nullptr, *keys2 = nullptr, *ukeys = nullptr;
    double *vals2 = nullptr;
    uint8_t *flag = nullptr;
    int32_t *pos = nullptr, *unused = nullptr;
    uint32_t *counts = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    const size_t nn = n ? n : 1;
    const unsigned grid = (unsigned)((nn + 255) / 256);
    auto cleanup = [&]() {
        (void)hipFree(keys); (void)hipFree(keys2); (void)hipFree(ukeys); (void)hipFree(vals2); (void)hipFree(flag);
        (void)hipFree(pos); (void)hipFree(unused); (void)hipFree(counts); (void)hipFree(tmp);
    };
#define CSR_TRY(call) do { e = (call); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    CSR_TRY(hipMalloc((void **)&keys, nn * 8));
    CSR_TRY(hipMalloc((void **)&keys2, nn * 8));
    CSR_TRY(hipMalloc((void **)&vals2, nn * 8));
    size_t nnz = 0;
    if (n) {
        hipLaunchKernelGGL(csr_keys_kernel, dim3(grid), dim3(256), 0, stream, n, d_rows, d_cols, keys);
        CSR_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, d_vals, vals2, n, 0, 64, stream));
        CSR_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1));
        CSR_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, d_vals, vals2, n, 0, 64, stream));
        CSR_TRY(hipMalloc((void **)&flag, nn));
        CSR_TRY(hipMalloc((void **)&pos, nn * 4));
        CSR_TRY(hipMalloc((void **)&unused, nn * 4));
        hipLaunchKernelGGL(csr_heads_kernel, dim3(grid), dim3(256), 0, stream, n, keys2, flag);
        const uint32_t nblocks = (uint32_t)((n + SCAN_TILE - 1) / SCAN_TILE);
        CSR_TRY(hipMalloc((void **)&counts, (nblocks + 1) * sizeof(uint32_t)));
        hipLaunchKernelGGL(active_count_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, stream, flag, (uint32_t)n, counts);
        hipLaunchKernelGGL(active_block_scan_kernel, dim3(1), dim3(SCAN_BLOCK), 0, stream, counts, nblocks);
        hipLaunchKernelGGL(active_tables_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, stream, flag, (uint32_t)n, counts, unused, pos);
        uint32_t total = 0;
        CSR_TRY(hipMemcpyAsync(&total, counts + nblocks, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        CSR_TRY(hipStreamSynchronize(stream));
        nnz = total;
        CSR_TRY(hipMalloc((void **)&ukeys, (nnz ? nnz : 1) * 8));
        hipLaunchKernelGGL(csr_reduce_kernel, dim3(grid), dim3(256), 0, stream, n, keys2, vals2, flag, pos, d_colind, d_values, ukeys);
    } else {
        CSR_TRY(hipMalloc((void **)&ukeys, 8));
    }
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)((nrows + 1 + 255) / 256)), dim3(256), 0, stream, nrows, nnz, ukeys, d_rowptr);
    CSR_TRY(hipGetLastError());
    CSR_TRY(hipStreamSynchronize(stream));
#undef CSR_TRY
    cleanup();
    if (nnz_out) *nnz_out = nnz;
    return hipSuccess;
}

}  // namespace pa
