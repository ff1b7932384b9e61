// condensed.hip -- the face-only global system of the condensed mode.
//
// The reference's assembler numbers cell AND face unknowns (system_size = cbs * ncells + fbs * num_other_faces,
// src/methods/hho_bits/hho.hpp:331) and pushes msize^2 triplets per cell (hho.hpp:391-403).  With the cell unknowns
// eliminated on chip (hho_device.hpp, MODE_COND) what is left is the same numbering without its cell block: unknown k
// of non-Dirichlet face F at compress_table[F] * fbs + k (hho.hpp:305-323, 374-379), Dirichlet columns moved to the
// right-hand side (hho.hpp:381-401).  This file holds
//   * that assembly as triplets in the reference's push order (condensed_triplets_kernel), and
//   * the same matrix built DIRECTLY in CSR from the mesh's face adjacency -- no triplets, no sort: a face's rows hold
//     the unknowns of the (at most 7) non-Dirichlet faces of its (at most 2) cells, in ascending order of their
//     compressed ids; a symbolic phase (once per mesh) records per face which local faces of which cell those are, and
//     the numeric phase is a pure gather: one thread per CSR entry, every entry written once, consecutive threads
//     writing consecutive entries.
// Row partition for the multi-GPU path: a slab owns the faces of its cell rows' blocks (bottom and vertical faces
// of every cell row); the bottom faces of a slab with a slab below also take the contribution of that slab's top
// cells, which arrives as fbs packed rows per cell (condensed_halo_pack_kernel) -- the whole exchange of a step.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <cstdint>

#include "condensed.hpp"

namespace pa {

// entry (i, j) of the packed record's symmetric S; g follows the triangle
__device__ __forceinline__ double cond_S(const double *rec, int i, int j)
{
    const int a = i < j ? i : j, b = i < j ? j : i;
    return rec[b * (b + 1) / 2 + a];
}

// ---- adjacency: the two cells of every local face, lower cell id first ---------------------------------------------
__global__ __launch_bounds__(256) void cond_adj_init_kernel(uint32_t nfaces, int32_t *adj)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nfaces) { adj[2 * t] = 0x7fffffff; adj[2 * t + 1] = -1; }
}

__global__ __launch_bounds__(256) void cond_adj_cells_kernel(uint32_t ncells, const uint32_t *cell_faces, int32_t *adj)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * ncells) return;
    const uint32_t f = cell_faces[t];
    atomicMin(&adj[2 * f], (int32_t)(t / 4));
    atomicMax(&adj[2 * f + 1], (int32_t)(t / 4));
}

// The symbolic record of an owned, non-Dirichlet face: its column faces in ascending compressed order and where each
// of them sits in the face's cells.
//   code[s] bits 0-1: local face index of column face s in cell A, bit 2: present in A; bits 3-4 / 5: the same for B
__device__ inline void cond_describe_face(const CondMesh &m, uint32_t f, int32_t &cA, int32_t &cB, int &rowA, int &rowB,
                                          int32_t (&colcomp)[7], uint8_t (&code)[7], int &ncol, bool &dirichlet_cols)
{
    dirichlet_cols = false;
    int32_t a = m.adj[2 * f], b = m.adj[2 * f + 1];
    if (b == a) b = -1;
    if (a == 0x7fffffff) a = -1;
    // a slab with a slab below: the cells under its bottom faces are remote (their rows arrive packed)
    const bool remote_below = m.structured && m.sm.row0 > 0 && f < 2 * m.sm.Nx && (f % 2 == 0);
    if (remote_below) { b = a; a = -2 - (int32_t)(f / 2); }
    cA = a; cB = b;
    int32_t cand_comp[8]; uint8_t cand_code[8];
    int nc = 0;
    rowA = rowB = 0;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const int32_t c = side == 0 ? a : b;
        if (c == -1) continue;
#pragma unroll
        for (int lf = 0; lf < 4; ++lf) {
            int32_t comp; bool self;
            if (c <= -2) {
                // remote cell (i, row0 - 1): its faces in closed form (bottom, right, top, left)
                const uint32_t i = (uint32_t)(-2 - c), j = m.sm.row0 - 1;
                const uint32_t gid = lf == 0 ? sm_hface(m.sm, i, j) : lf == 1 ? sm_vface(m.sm, i + 1, j)
                                   : lf == 2 ? sm_hface(m.sm, i, j + 1) : sm_vface(m.sm, i, j);
                uint32_t lo, hi; bool d;
                sm_face_decode(m.sm, gid, lo, hi, d, comp);
                self = lf == 2;
            } else {
                const uint32_t fl = m.cell_faces[4 * (uint32_t)c + lf];
                comp = m.face_compress[fl];
                self = fl == f;
            }
            if (self) { if (side == 0) rowA = lf; else rowB = lf; }
            if (comp < 0) { if (c >= 0) dirichlet_cols = true; continue; }      // Dirichlet: no column (hho.hpp:398), its data go to the rhs
            int pos = -1;
            for (int q = 0; q < nc; ++q) if (cand_comp[q] == comp) pos = q;
            if (pos < 0) { pos = nc++; cand_comp[pos] = comp; cand_code[pos] = 0; }
            cand_code[pos] |= side == 0 ? (uint8_t)(lf | 4) : (uint8_t)((lf << 3) | 32);
        }
    }
    // ascending compressed id (insertion sort of at most 7)
    for (int q = 1; q < nc; ++q) {
        const int32_t kc = cand_comp[q]; const uint8_t kd = cand_code[q];
        int r = q - 1;
        while (r >= 0 && cand_comp[r] > kc) { cand_comp[r + 1] = cand_comp[r]; cand_code[r + 1] = cand_code[r]; --r; }
        cand_comp[r + 1] = kc; cand_code[r + 1] = kd;
    }
    ncol = nc;
    for (int q = 0; q < 7; ++q) { colcomp[q] = q < nc ? cand_comp[q] : -1; code[q] = q < nc ? cand_code[q] : 0; }
}

// one thread per local face: owned non-Dirichlet faces write their record at position compress - p0
__global__ __launch_bounds__(256) void cond_symbolic_kernel(CondMesh m, uint32_t nfaces_owned_range, int32_t p0, uint32_t nown,
                                                            CondFace *faces, CondFaceLean *lean, uint32_t *ncols)
{
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces_owned_range) return;
    const int32_t comp = m.face_compress[f];
    if (comp < 0) return;
    const uint32_t q = (uint32_t)(comp - p0);
    if (q >= nown) return;
    CondFace r;
    int rowA, rowB, ncol;
    bool dcols;
    cond_describe_face(m, f, r.cA, r.cB, rowA, rowB, r.colcomp, r.code, ncol, dcols);
    r.rows = (uint8_t)(rowA | (rowB << 2));
    r.ncol = (uint8_t)ncol;
    r.face = f;
    faces[q] = r;
    ncols[q] = (uint32_t)ncol;
    CondFaceLean ln;
    ln.cA = r.cA; ln.cB = r.cB;
    ln.packed = ((uint64_t)r.rows << 42) | ((uint64_t)ncol << 46) | ((uint64_t)(dcols ? 1 : 0) << 49);
    for (int s = 0; s < 7; ++s) ln.packed |= (uint64_t)(r.code[s] & 63) << (6 * s);
    lean[q] = ln;
}

// rowptr / colind of the owned rows for one fbs: row (q, k) starts at fbs^2 prefix[q] + k fbs ncol[q]
__global__ __launch_bounds__(256) void cond_pattern_kernel(uint32_t nown, int fbs, const CondFace *faces, const uint32_t *prefix,
                                                           int64_t *rowptr, int32_t *colind)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = t / (uint32_t)fbs;
    const int k = (int)(t % (uint32_t)fbs);
    if (q > nown || (q == nown && k > 0)) return;
    if (q == nown) { rowptr[(size_t)nown * fbs] = (int64_t)prefix[nown] * fbs * fbs; return; }
    const CondFace &r = faces[q];
    const int64_t start = (int64_t)prefix[q] * fbs * fbs + (int64_t)k * fbs * r.ncol;
    rowptr[(size_t)q * fbs + k] = start;
    if (colind != nullptr)
        for (int s = 0; s < r.ncol; ++s)
            for (int kp = 0; kp < fbs; ++kp) colind[start + s * fbs + kp] = r.colcomp[s] * fbs + kp;
}

// contribution of cell `c` (local; rec = its packed record) to right-hand-side row `row` (a face unknown index in
// 0..nf-1): g_row minus the Dirichlet columns times the boundary data, accumulated in local column order (hho.hpp:401)
__device__ __forceinline__ double cond_rhs_contrib(const CondMesh &m, const double *rec, int32_t c, int row, int fbs, const double *g)
{
    const int nf = 4 * fbs;
    double s = rec[nf * (nf + 1) / 2 + row];
    for (int lf = 0; lf < 4; ++lf) {
        const uint32_t fl = m.cell_faces[4 * (uint32_t)c + lf];
        if (m.face_compress[fl] >= 0) continue;
        for (int kp = 0; kp < fbs; ++kp) {
            const double d = g != nullptr ? g[(size_t)fl * fbs + kp] : 0.0;
            s -= cond_S(rec, row, lf * fbs + kp) * d;
        }
    }
    return s;
}

// numeric phase: one thread per CSR entry of a face's rows -- (row k, column slot s, column k') with k' fastest, so
// that consecutive lanes write consecutive entries.  A wavefront takes U groups of faces at a time (a group: one face of
// 7 fbs^2 <= 112 entries, or 64 / (7 fbs^2) whole faces for fbs <= 2); everything is unrolled over the U groups, so that
// their 16-byte descriptors, then their gathers from the packed records, are in flight together: the kernel is a chain
// of three dependent memory round trips (descriptor, record entries, store) and lives on how many of them overlap
// (one group at a time: 1.8 ms for the 132 M entries of the 1024^2 k = 2 system).
template <int FBS, int U>
__global__ __launch_bounds__(256) void cond_fill_kernel(uint32_t nown, const CondFaceLean *__restrict__ lean,
                                                        const uint32_t *__restrict__ prefix, const double *__restrict__ cond,
                                                        const double *__restrict__ halo, double *__restrict__ values)
{
    constexpr uint32_t per_face = 7u * FBS * FBS;
    constexpr uint32_t FPW = per_face <= 32 ? 64u / per_face : 1u;
    constexpr uint32_t PASSES = (per_face + 63u) / 64u;
    constexpr int nf = 4 * FBS, ncond = nf * (nf + 1) / 2 + nf, hd = FBS * (nf + 1);
    const uint32_t lane = threadIdx.x % 64u, wave = blockIdx.x * (256u / 64u) + threadIdx.x / 64u;
    const uint32_t sub = FPW > 1 ? lane / per_face : 0u;
    const uint32_t e0 = FPW > 1 ? lane - sub * per_face : lane;
    CondFaceLean d[U];
    uint32_t pre[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t q = (wave * U + u) * FPW + sub;
        on[u] = q < nown && sub < FPW;
        const uint32_t qq = on[u] ? q : 0u;
        d[u] = lean[qq];
        pre[u] = prefix[qq];
    }
    const double *pa_[U][PASSES], *pb_[U][PASSES];
    size_t out[U][PASSES];
    bool ok[U][PASSES], two[U][PASSES];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t ncol = (uint32_t)(d[u].packed >> 46) & 7u, rows = (uint32_t)(d[u].packed >> 42) & 15u;
        const uint32_t rowlen = ncol * FBS;
#pragma unroll
        for (uint32_t p = 0; p < PASSES; ++p) {
            const uint32_t e = e0 + 64u * p;
            ok[u][p] = on[u] && e < rowlen * FBS;
            const uint32_t k = (e >= rowlen) + (e >= 2 * rowlen) + (e >= 3 * rowlen);
            const uint32_t er = ok[u][p] ? e - k * rowlen : 0u;
            const uint32_t s = er / (uint32_t)FBS, kp = er % (uint32_t)FBS;
            const uint32_t code = (uint32_t)(d[u].packed >> (6 * s)) & 63u;
            const int rowA = (int)((rows & 3u) * FBS + k), rowB = (int)(((rows >> 2) & 3u) * FBS + k);
            const int colA = (int)((code & 3u) * FBS + kp), colB = (int)(((code >> 3) & 3u) * FBS + kp);
            const bool hasA = code & 4u, hasB = code & 32u;
            const int aA = rowA < colA ? rowA : colA, bA = rowA < colA ? colA : rowA;
            const int aB = rowB < colB ? rowB : colB, bB = rowB < colB ? colB : rowB;
            const double *pA = d[u].cA >= 0 ? cond + (size_t)d[u].cA * ncond + (bA * (bA + 1) / 2 + aA)
                                            : halo + (size_t)(d[u].cA <= -2 ? -2 - d[u].cA : 0) * hd + (k * nf + colA);
            const double *pB = cond + (size_t)(d[u].cB >= 0 ? d[u].cB : 0) * ncond + (bB * (bB + 1) / 2 + aB);
            // one or two addends; a lane without work re-reads entry 0 of the records (never stored)
            const bool useA = ok[u][p] && hasA, useB = ok[u][p] && hasB;
            pa_[u][p] = useA ? pA : (useB ? pB : cond);
            pb_[u][p] = useB ? pB : cond;
            two[u][p] = useA && useB;
            out[u][p] = (size_t)pre[u] * (FBS * FBS) + e;
        }
    }
    double va[U][PASSES], vb[U][PASSES];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (uint32_t p = 0; p < PASSES; ++p) { va[u][p] = *pa_[u][p]; vb[u][p] = *pb_[u][p]; }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (uint32_t p = 0; p < PASSES; ++p)
            if (ok[u][p]) values[out[u][p]] = two[u][p] ? va[u][p] + vb[u][p] : va[u][p];
}

// right-hand side of the owned rows: one thread per (owned face, row k)
__global__ __launch_bounds__(256) void cond_rhs_rows_kernel(CondMesh m, uint32_t nown, int fbs, const CondFaceLean *lean,
                                                            const double *cond, const double *g, const double *halo, double *rhs)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = t / (uint32_t)fbs;
    if (q >= nown) return;
    const int k = (int)(t % (uint32_t)fbs);
    const CondFaceLean r = lean[q];
    const uint32_t rows = (uint32_t)(r.packed >> 42) & 15u;
    const int nf = 4 * fbs, ncond = nf * (nf + 1) / 2 + nf, hd = fbs * (nf + 1);
    const int rowA = (int)(rows & 3u) * fbs + k, rowB = (int)((rows >> 2) & 3u) * fbs + k;
    // (bit 49: some face of its local cells is Dirichlet; otherwise a cell's contribution is its g entry alone)
    const bool dcols = (r.packed >> 49) & 1u;
    const int og = nf * (nf + 1) / 2;
    double b = 0.0;
    bool have = false;
    if (r.cA >= 0) {
        const double *rec = cond + (size_t)r.cA * ncond;
        b = dcols ? cond_rhs_contrib(m, rec, r.cA, rowA, fbs, g) : rec[og + rowA];
        have = true;
    } else if (r.cA <= -2) { b = halo[(size_t)(-2 - r.cA) * hd + fbs * nf + k]; have = true; }
    if (r.cB >= 0) {
        const double *rec = cond + (size_t)r.cB * ncond;
        const double w = dcols ? cond_rhs_contrib(m, rec, r.cB, rowB, fbs, g) : rec[og + rowB];
        b = have ? b + w : w;
    }
    rhs[t] = b;
}

// the rows of the top faces of the slab's top cell row, for the slab above: fbs x nf values, then fbs rhs values
__global__ __launch_bounds__(256) void cond_halo_pack_kernel(CondMesh m, uint32_t first_cell, uint32_t ncells_row, int fbs,
                                                             const double *cond, const double *g, double *halo)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = 4 * fbs, ncond = nf * (nf + 1) / 2 + nf, hd = fbs * (nf + 1);
    if (t >= ncells_row * (uint32_t)hd) return;
    const uint32_t i = t / (uint32_t)hd;
    const int e = (int)(t % (uint32_t)hd);
    const uint32_t c = first_cell + i;
    const double *rec = cond + (size_t)c * ncond;
    double v;
    if (e < fbs * nf) v = cond_S(rec, 2 * fbs + e / nf, e % nf);
    else v = cond_rhs_contrib(m, rec, (int32_t)c, 2 * fbs + (e - fbs * nf), fbs, g);
    halo[t] = v;
}

// assembler::assemble on the condensed block: nf^2 slots per cell in the reference's push order (hho.hpp:391-403)
__global__ __launch_bounds__(256) void condensed_triplets_kernel(CondMesh m, size_t first, size_t n, int fbs, const double *cond,
                                                                 const double *g, int32_t *rows, int32_t *cols, double *vals,
                                                                 int32_t *rhs_rows, double *rhs_vals)
{
    const int nf = 4 * fbs, ncond = nf * (nf + 1) / 2 + nf;
    __shared__ int32_t idx[16];
    __shared__ double dd[16];
    for (size_t c = blockIdx.x; c < n; c += gridDim.x) {
        const size_t cl = first + c;
        if ((int)threadIdx.x < nf) {
            const int lf = threadIdx.x / fbs, k = threadIdx.x % fbs;
            const uint32_t fl = m.cell_faces[4 * cl + lf];
            const int32_t comp = m.face_compress[fl];
            idx[threadIdx.x] = comp < 0 ? -1 : comp * fbs + k;                                   // hho.hpp:374-379 less the cell block
            dd[threadIdx.x] = (comp < 0 && g != nullptr) ? g[(size_t)fl * fbs + k] : 0.0;         // :381-386
        }
        __syncthreads();
        const double *rec = cond + c * (size_t)ncond;
        for (int e = threadIdx.x; e < nf * nf; e += blockDim.x) {
            const int i = e / nf, j = e % nf;
            const bool keep = idx[i] >= 0 && idx[j] >= 0;                                         // :393,398
            const size_t o = c * (size_t)(nf * nf) + e;
            rows[o] = keep ? idx[i] : -1;
            cols[o] = keep ? idx[j] : -1;
            vals[o] = cond_S(rec, i, j);
        }
        if ((int)threadIdx.x < nf) {
            const int i = threadIdx.x;
            double s = rec[nf * (nf + 1) / 2 + i];
            if (idx[i] >= 0)
                for (int j = 0; j < nf; ++j)
                    if (idx[j] < 0) s -= cond_S(rec, i, j) * dd[j];                               // :401
            rhs_rows[c * nf + i] = idx[i];
            rhs_vals[c * nf + i] = idx[i] >= 0 ? s : 0.0;
        }
        __syncthreads();
    }
}

// take_local_data restricted to the faces (hho.hpp:408-449)
__global__ __launch_bounds__(256) void condensed_take_faces_kernel(CondMesh m, size_t first, size_t n, int fbs, const double *solution,
                                                                   const double *g, double *uF)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nf = 4 * fbs;
    if (t >= n * nf) return;
    const size_t cl = first + t / nf;
    const int lf = (int)(t % nf) / fbs, k = (int)(t % nf) % fbs;
    const uint32_t fl = m.cell_faces[4 * cl + lf];
    const int32_t comp = m.face_compress[fl];
    uF[t] = comp < 0 ? (g != nullptr ? g[(size_t)fl * fbs + k] : 0.0) : solution[(size_t)comp * fbs + k];
}

__global__ __launch_bounds__(256) void condensed_expand_kernel(size_t ncells_local, size_t cell_base, size_t ncells_global, int cbs,
                                                               size_t nface_dofs, const double *uT, const double *xF, double *full)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ncd = ncells_local * cbs;
    if (t < ncd) full[cell_base * cbs + t] = uT[t];
    else if (xF != nullptr && t < ncd + nface_dofs) full[ncells_global * cbs + (t - ncd)] = xF[t - ncd];
}

// ---- host side ---------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256 ? (n + 255) / 256 : 1); }

hipError_t cond_build_tables(hipStream_t stream, CondMesh m, uint32_t nfaces_local, uint32_t ncells, uint32_t owned_range,
                             int32_t p0, uint32_t nown, int32_t *adj, CondFace *faces, CondFaceLean *lean, uint32_t *ncols,
                             uint32_t *prefix)
{
    hipLaunchKernelGGL(cond_adj_init_kernel, dim3(blocks_for(nfaces_local)), dim3(256), 0, stream, nfaces_local, adj);
    hipLaunchKernelGGL(cond_adj_cells_kernel, dim3(blocks_for((size_t)4 * ncells)), dim3(256), 0, stream, ncells, m.cell_faces, adj);
    hipError_t e = hipMemsetAsync(ncols, 0, ((size_t)nown + 1) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    m.adj = adj;
    hipLaunchKernelGGL(cond_symbolic_kernel, dim3(blocks_for(owned_range)), dim3(256), 0, stream, m, owned_range, p0, nown, faces, lean, ncols);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tmp_bytes = 0;
    void *tmp = nullptr;
    e = rocprim::exclusive_scan(nullptr, tmp_bytes, ncols, prefix, 0u, (size_t)nown + 1, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(tmp, tmp_bytes, ncols, prefix, 0u, (size_t)nown + 1, rocprim::plus<uint32_t>(), stream);
    hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(tmp);
    return e != hipSuccess ? e : e2;
}

hipError_t cond_pattern(hipStream_t stream, uint32_t nown, int fbs, const CondFace *faces, const uint32_t *prefix, int64_t *rowptr,
                        int32_t *colind)
{
    hipLaunchKernelGGL(cond_pattern_kernel, dim3(blocks_for(((size_t)nown + 1) * fbs)), dim3(256), 0, stream, nown, fbs, faces, prefix,
                       rowptr, colind);
    return hipGetLastError();
}

#ifndef PA_FILL_UNROLL
#define PA_FILL_UNROLL 4
#endif
hipError_t cond_fill(hipStream_t stream, const CondMesh &m, uint32_t nown, int fbs, const CondFaceLean *lean, const uint32_t *prefix,
                     const double *cond, const double *g, const double *halo, double *values, double *rhs)
{
    if (nown == 0) return hipSuccess;
    constexpr int U = PA_FILL_UNROLL;
    const uint32_t per_face = 7u * fbs * fbs, fpw = per_face <= 32 ? 64u / per_face : 1u;
    const uint32_t faces_per_block = 4u * U * fpw;
    const dim3 grid((nown + faces_per_block - 1) / faces_per_block), block(256);
    const double *h = halo ? halo : cond;
    switch (fbs) {
    case 1: hipLaunchKernelGGL((cond_fill_kernel<1, U>), grid, block, 0, stream, nown, lean, prefix, cond, h, values); break;
    case 2: hipLaunchKernelGGL((cond_fill_kernel<2, U>), grid, block, 0, stream, nown, lean, prefix, cond, h, values); break;
    case 3: hipLaunchKernelGGL((cond_fill_kernel<3, U>), grid, block, 0, stream, nown, lean, prefix, cond, h, values); break;
    case 4: hipLaunchKernelGGL((cond_fill_kernel<4, U>), grid, block, 0, stream, nown, lean, prefix, cond, h, values); break;
    default: return hipErrorInvalidValue;
    }
    if (rhs != nullptr)
        hipLaunchKernelGGL(cond_rhs_rows_kernel, dim3(blocks_for((size_t)nown * fbs)), dim3(256), 0, stream, m, nown, fbs, lean, cond, g,
                           h, rhs);
    return hipGetLastError();
}

hipError_t cond_halo_pack(hipStream_t stream, const CondMesh &m, uint32_t first_cell, uint32_t ncells_row, int fbs, const double *cond,
                          const double *g, double *halo)
{
    if (ncells_row == 0) return hipSuccess;
    const size_t total = (size_t)ncells_row * fbs * (4 * fbs + 1);
    hipLaunchKernelGGL(cond_halo_pack_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, m, first_cell, ncells_row, fbs, cond, g, halo);
    return hipGetLastError();
}

hipError_t cond_triplets(hipStream_t stream, const CondMesh &m, int num_cus, size_t first, size_t n, int fbs, const double *cond,
                         const double *g, int32_t *rows, int32_t *cols, double *vals, int32_t *rhs_rows, double *rhs_vals)
{
    if (n == 0) return hipSuccess;
    const size_t resident = (size_t)num_cus * 8;
    hipLaunchKernelGGL(condensed_triplets_kernel, dim3((unsigned)(n < resident ? n : resident)), dim3(256), 0, stream, m, first, n, fbs,
                       cond, g, rows, cols, vals, rhs_rows, rhs_vals);
    return hipGetLastError();
}

hipError_t cond_take_faces(hipStream_t stream, const CondMesh &m, size_t first, size_t n, int fbs, const double *solution,
                           const double *g, double *uF)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(condensed_take_faces_kernel, dim3(blocks_for(n * 4 * fbs)), dim3(256), 0, stream, m, first, n, fbs, solution, g, uF);
    return hipGetLastError();
}

hipError_t cond_expand(hipStream_t stream, size_t ncells_local, size_t cell_base, size_t ncells_global, int cbs, size_t nface_dofs,
                       const double *uT, const double *xF, double *full)
{
    const size_t total = ncells_local * cbs + (xF ? nface_dofs : 0);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(condensed_expand_kernel, dim3(blocks_for(total)), dim3(256), 0, stream, ncells_local, cell_base, ncells_global, cbs,
                       nface_dofs, uT, xF, full);
    return hipGetLastError();
}

}  // namespace pa
