// scan.hpp -- exclusive prefix count of byte flags over n items: a three-pass scan (per-block
// counts, scan of the block counts by one block, per-block rescan with the block offset).  Used for
// the compress tables of obstacle_assembler (hho.hpp:538-578) and for the run heads of the device
// CSR build (csr.hip).  Kernels have internal linkage: the header is included in several units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pa {

constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *sh, uint32_t &total)
{
    // wave scan with DPP-free shuffles, then a scan of the 4 wave totals through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        if (w < wave) base += sh[w];
        tot += sh[w];
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

static __global__ __launch_bounds__(SCAN_BLOCK) void active_count_kernel(const uint8_t *in_A, uint32_t n, uint32_t *block_counts)
{
    __shared__ uint32_t sh[SCAN_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) cnt += in_A[base + k] ? 1 : 0;
    uint32_t total;
    block_exclusive_scan(cnt, sh, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// one block: exclusive scan of the block counts in place; the grand total goes to counts[nblocks]
static __global__ __launch_bounds__(SCAN_BLOCK) void active_block_scan_kernel(uint32_t *counts, uint32_t nblocks)
{
    __shared__ uint32_t sh[SCAN_BLOCK / 64];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += SCAN_BLOCK) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < nblocks ? counts[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, sh, total);
        if (i < nblocks) counts[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) counts[nblocks] = carry;
}

static __global__ __launch_bounds__(SCAN_BLOCK) void active_tables_kernel(const uint8_t *in_A, uint32_t n, const uint32_t *block_offsets,
                                                                   int32_t *A_ct, int32_t *B_ct)
{
    __shared__ uint32_t sh[SCAN_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint8_t flag[SCAN_ITEMS];
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        flag[k] = (base + k < n && in_A[base + k]) ? 1 : 0;
        cnt += flag[k];
    }
    uint32_t total;
    uint32_t active_before = block_offsets[blockIdx.x] + block_exclusive_scan(cnt, sh, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint32_t i = base + k;
        if (i < n) {
            B_ct[i] = flag[k] ? (int32_t)active_before : -1;               // hho.hpp:567-578
            A_ct[i] = flag[k] ? -1 : (int32_t)(i - active_before);        // hho.hpp:538-549
        }
        active_before += flag[k];
    }
}

}  // namespace pa
