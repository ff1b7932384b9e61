// assembler_csr.hip -- the reference's OWN global system (cell + face unknowns) built directly in CSR.
//
// assembler<Mesh> (src/methods/hho_bits/hho.hpp:252-463) numbers the cell unknowns of cell c at c * cbs + i and the
// unknowns of a non-Dirichlet face F behind them at cbs * ncells + compress_table[F] * fbs + k (:362-379), pushes
// msize^2 triplets per cell in row-major local order (:391-403) and hands them to setFromTriplets (:451-455), which
// sums duplicates and sorts the columns of a row.  That is the span the reference's drivers print as "Matrix assembly"
// (apps/cuthho/cuthho_square.cpp:881-905, apps/convergence_test/convergence_test.cpp:201-217).
//
// Here the same matrix comes out of the mesh's face adjacency without triplets and without a sort (the design of
// condensed.hip, extended by the cell block):
//   * a CELL row (c, i) holds the cbs unknowns of its own cell, then the unknowns of the cell's non-Dirichlet faces in
//     ascending order of their compressed ids: one contribution each, lc_c(i, j);
//   * a FACE row (F, k) holds the cell unknowns of F's (at most two) cells, lower cell id first, then the unknowns of the (at
//     most 7) non-Dirichlet faces of those cells in ascending compressed order: one contribution for the cell columns and
//     for the faces of one cell only, two -- lower cell id first, the order in which setFromTriplets meets the duplicates --
//     for F itself.
// Row pointers are closed forms of two prefix counts (non-Dirichlet faces per cell; cells and column faces per face); the
// symbolic phase runs once per mesh, the numeric phase is a pure gather from lc: one thread per CSR entry, consecutive
// lanes writing consecutive entries.  Structure and values are bit-identical to pa_csr_from_triplets(pa_triplets_batch(..))
// (tests/test_gpu_assembler.py); the right-hand side is the triplet path's per-row sums (:401, :405) added in cell order.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <cstdint>

#include "assembler_csr.hpp"

namespace pa {

static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256 ? (n + 255) / 256 : 1); }

// ---- symbolic: per cell the number of its non-Dirichlet faces, per non-Dirichlet face the number of its cells ------------
__global__ __launch_bounds__(256) void asm_counts_kernel(uint32_t ncells, uint32_t nown, const uint32_t *cell_faces,
                                                         const int32_t *face_compress, const CondFaceLean *lean, uint32_t *nfc,
                                                         uint32_t *nfcell)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncells) {
        const uint4 f = *reinterpret_cast<const uint4 *>(cell_faces + 4 * (size_t)t);
        nfc[t] = (face_compress[f.x] >= 0) + (face_compress[f.y] >= 0) + (face_compress[f.z] >= 0) + (face_compress[f.w] >= 0);
    }
    if (t < nown) nfcell[t] = (lean[t].cA >= 0) + (lean[t].cB >= 0);
    if (t == ncells) nfc[ncells] = 0;
    if (t == nown) nfcell[nown] = 0;
}

hipError_t asm_build_tables(hipStream_t stream, const CondMesh &m, uint32_t ncells, uint32_t nown, const CondFaceLean *lean,
                            uint32_t *nfc, uint32_t *cprefix, uint32_t *nfcell, uint32_t *fprefix)
{
    const uint32_t top = ncells > nown ? ncells : nown;
    hipLaunchKernelGGL(asm_counts_kernel, dim3(blocks_for((size_t)top + 1)), dim3(256), 0, stream, ncells, nown, m.cell_faces,
                       m.face_compress, lean, nfc, nfcell);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t b1 = 0, b2 = 0;
    e = rocprim::exclusive_scan(nullptr, b1, nfc, cprefix, 0u, (size_t)ncells + 1, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(nullptr, b2, nfcell, fprefix, 0u, (size_t)nown + 1, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    const size_t tb = b1 > b2 ? b1 : b2;
    e = hipMalloc(&tmp, tb ? tb : 1);
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(tmp, b1, nfc, cprefix, 0u, (size_t)ncells + 1, rocprim::plus<uint32_t>(), stream);
    if (e == hipSuccess) e = rocprim::exclusive_scan(tmp, b2, nfcell, fprefix, 0u, (size_t)nown + 1, rocprim::plus<uint32_t>(), stream);
    const hipError_t e2 = hipStreamSynchronize(stream);
    (void)hipFree(tmp);
    return e != hipSuccess ? e : e2;
}

// the non-Dirichlet faces of a cell in ascending compressed order: comp[s], local index lf[s], s < n
struct CellFaces {
    int32_t comp[4];
    int lf[4];
    int n;
};
__device__ __forceinline__ CellFaces cell_faces_sorted(const uint32_t *cell_faces, const int32_t *face_compress, uint32_t c)
{
    const uint4 f = *reinterpret_cast<const uint4 *>(cell_faces + 4 * (size_t)c);
    const int32_t cc[4] = {face_compress[f.x], face_compress[f.y], face_compress[f.z], face_compress[f.w]};
    CellFaces r;
    r.n = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r.comp[q] = 0x7fffffff; r.lf[q] = 0; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (cc[q] >= 0) {
            // insertion into the sorted prefix (at most 4 entries)
            int p = r.n;
#pragma unroll
            for (int s = 2; s >= 0; --s)
                if (s < r.n && r.comp[s] > cc[q]) { r.comp[s + 1] = r.comp[s]; r.lf[s + 1] = r.lf[s]; p = s; }
            r.comp[p] = cc[q]; r.lf[p] = q;
            ++r.n;
        }
    return r;
}

struct AsmDims {
    int cbs, fbs, ms;
    uint32_t ncells, nown;
    uint64_t cell_nnz;        // entries of all cell rows = cbs (ncells cbs + cprefix[ncells] fbs)
};

__device__ __forceinline__ uint64_t cell_block_start(const AsmDims &d, uint32_t c, uint32_t cpre)
{
    return (uint64_t)d.cbs * ((uint64_t)c * d.cbs + (uint64_t)cpre * d.fbs);
}
__device__ __forceinline__ uint64_t face_block_start(const AsmDims &d, uint32_t fcpre, uint32_t colpre)
{
    return d.cell_nnz + (uint64_t)d.fbs * ((uint64_t)fcpre * d.cbs + (uint64_t)colpre * d.fbs);
}

// ---- pattern: row pointers and column indices ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void asm_pattern_cells_kernel(AsmDims d, const uint32_t *cell_faces, const int32_t *face_compress,
                                                                const uint32_t *cprefix, int64_t *rowptr, int32_t *colind)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = t / (uint32_t)d.cbs;
    const int i = (int)(t % (uint32_t)d.cbs);
    if (c >= d.ncells) return;
    const CellFaces cf = cell_faces_sorted(cell_faces, face_compress, c);
    const int R = d.cbs + cf.n * d.fbs;
    const uint64_t start = cell_block_start(d, c, cprefix[c]) + (uint64_t)i * R;
    rowptr[(size_t)c * d.cbs + i] = (int64_t)start;
    if (colind == nullptr) return;
    for (int j = 0; j < d.cbs; ++j) colind[start + j] = (int32_t)((uint64_t)c * d.cbs + j);
    for (int s = 0; s < cf.n; ++s)
        for (int kp = 0; kp < d.fbs; ++kp)
            colind[start + d.cbs + s * d.fbs + kp] = (int32_t)((uint64_t)d.cbs * d.ncells + (uint64_t)cf.comp[s] * d.fbs + kp);
}

__global__ __launch_bounds__(256) void asm_pattern_faces_kernel(AsmDims d, const CondFace *faces, const uint32_t *colprefix,
                                                                const uint32_t *fprefix, int64_t *rowptr, int32_t *colind)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = t / (uint32_t)d.fbs;
    const int k = (int)(t % (uint32_t)d.fbs);
    int64_t *rp = rowptr + (size_t)d.cbs * d.ncells;
    if (q > d.nown || (q == d.nown && k > 0)) return;
    if (q == d.nown) { rp[(size_t)d.nown * d.fbs] = (int64_t)face_block_start(d, fprefix[d.nown], colprefix[d.nown]); return; }
    const CondFace &r = faces[q];
    const int ncell = (r.cA >= 0) + (r.cB >= 0);
    const int R = ncell * d.cbs + r.ncol * d.fbs;
    const uint64_t start = face_block_start(d, fprefix[q], colprefix[q]) + (uint64_t)k * R;
    rp[(size_t)q * d.fbs + k] = (int64_t)start;
    if (colind == nullptr) return;
    int o = 0;
    if (r.cA >= 0) { for (int j = 0; j < d.cbs; ++j) colind[start + o + j] = (int32_t)((uint64_t)r.cA * d.cbs + j); o += d.cbs; }
    if (r.cB >= 0) { for (int j = 0; j < d.cbs; ++j) colind[start + o + j] = (int32_t)((uint64_t)r.cB * d.cbs + j); o += d.cbs; }
    for (int s = 0; s < r.ncol; ++s)
        for (int kp = 0; kp < d.fbs; ++kp)
            colind[start + o + s * d.fbs + kp] = (int32_t)((uint64_t)d.cbs * d.ncells + (uint64_t)r.colcomp[s] * d.fbs + kp);
}

hipError_t asm_pattern(hipStream_t stream, const CondMesh &m, int cbs, int fbs, uint32_t ncells, uint32_t nown, uint64_t cell_nnz,
                       const CondFace *faces, const uint32_t *colprefix, const uint32_t *cprefix, const uint32_t *fprefix,
                       int64_t *rowptr, int32_t *colind)
{
    const AsmDims d = {cbs, fbs, cbs + 4 * fbs, ncells, nown, cell_nnz};
    hipLaunchKernelGGL(asm_pattern_cells_kernel, dim3(blocks_for((size_t)ncells * cbs)), dim3(256), 0, stream, d, m.cell_faces,
                       m.face_compress, cprefix, rowptr, colind);
    hipLaunchKernelGGL(asm_pattern_faces_kernel, dim3(blocks_for(((size_t)nown + 1) * fbs)), dim3(256), 0, stream, d, faces, colprefix,
                       fprefix, rowptr, colind);
    return hipGetLastError();
}

// ---- numeric phase ---------------------------------------------------------------------------------------------------
// Cell rows: one wavefront per U cells at a time; lane e of a pass holds entry e of the cell's block of rows (row i = e / R,
// position e % R): consecutive lanes write consecutive entries.  The reads run along a row of the column-major lc (stride
// msize); the lines they touch are shared by the rows of the cell and stay in the vector L1 / L2.
template <int CBS, int FBS, int U>
__global__ __launch_bounds__(256) void asm_fill_cells_kernel(AsmDims d, uint32_t c_begin, uint32_t c_end, const uint32_t *__restrict__ cell_faces,
                                                             const int32_t *__restrict__ face_compress,
                                                             const uint32_t *__restrict__ cprefix, const double *__restrict__ lc,
                                                             const double *__restrict__ rhs, const double *__restrict__ g,
                                                             double *__restrict__ values, double *__restrict__ RHS)
{
    constexpr int MS = CBS + 4 * FBS, RMAX = CBS + 4 * FBS, EMAX = CBS * RMAX;
    constexpr int PASSES = (EMAX + 63) / 64;
    const uint32_t lane = threadIdx.x % 64u, wave = blockIdx.x * (256u / 64u) + threadIdx.x / 64u;
    CellFaces cf[U];
    uint32_t cpre[U];
    bool on[U];
    uint32_t cell[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t c = c_begin + wave * U + u;
        on[u] = c < c_end;
        cell[u] = on[u] ? c : 0u;
        cf[u] = cell_faces_sorted(cell_faces, face_compress, cell[u]);
        cpre[u] = cprefix[cell[u]];
    }
    const double *src[U][PASSES];
    size_t dst[U][PASSES];
    bool ok[U][PASSES];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t R = (uint32_t)(CBS + cf[u].n * FBS);
        const uint64_t start = cell_block_start(d, cell[u], cpre[u]);
        const double *A = lc + (size_t)cell[u] * (MS * MS);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t e = lane + 64u * p;
            ok[u][p] = on[u] && e < (uint32_t)CBS * R;
            const uint32_t ee = ok[u][p] ? e : 0u;
            const uint32_t i = ee / R, jj = ee - i * R;
            uint32_t j = jj;
            if (jj >= (uint32_t)CBS) {
                const uint32_t s = (jj - CBS) / (uint32_t)FBS, kp = (jj - CBS) % (uint32_t)FBS;
                const int lf = s == 0 ? cf[u].lf[0] : s == 1 ? cf[u].lf[1] : s == 2 ? cf[u].lf[2] : cf[u].lf[3];
                j = (uint32_t)(CBS + lf * FBS) + kp;
            }
            src[u][p] = A + (size_t)j * MS + i;
            dst[u][p] = (size_t)start + e;
        }
    }
    double v[U][PASSES];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < PASSES; ++p) v[u][p] = *src[u][p];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
            if (ok[u][p]) values[dst[u][p]] = v[u][p];
    // right-hand side of the cell rows: rhs_c(i) minus the Dirichlet columns times the boundary data, in local column order
    // (hho.hpp:401, 405; the sums pa_triplets_batch returns per local row)
    if (RHS != nullptr) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (on[u] && lane < (uint32_t)CBS) {
                const double *A = lc + (size_t)cell[u] * (MS * MS);
                double s = rhs != nullptr ? rhs[(size_t)cell[u] * CBS + lane] : 0.0;
                if (cf[u].n < 4) {
                    const uint4 f = *reinterpret_cast<const uint4 *>(cell_faces + 4 * (size_t)cell[u]);
                    const uint32_t fl[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                    for (int lf = 0; lf < 4; ++lf)
                        if (face_compress[fl[lf]] < 0)
                            for (int kp = 0; kp < FBS; ++kp) {
                                const double dd = g != nullptr ? g[(size_t)fl[lf] * FBS + kp] : 0.0;
                                s -= A[(size_t)(CBS + lf * FBS + kp) * MS + lane] * dd;
                            }
                }
                RHS[(size_t)cell[u] * CBS + lane] = s;
            }
        }
    }
}

// contribution of local cell c to the right-hand side of its local row `row` (a face row): minus the Dirichlet columns
// times the boundary data, accumulated from zero in local column order (hho.hpp:401)
template <int CBS, int FBS>
__device__ __forceinline__ double asm_face_rhs_contrib(const uint32_t *cell_faces, const int32_t *face_compress, const double *lc,
                                                       const double *g, int32_t c, int row)
{
    constexpr int MS = CBS + 4 * FBS;
    const double *A = lc + (size_t)c * (MS * MS);
    const uint4 f = *reinterpret_cast<const uint4 *>(cell_faces + 4 * (size_t)c);
    const uint32_t fl[4] = {f.x, f.y, f.z, f.w};
    double s = 0.0;
#pragma unroll
    for (int lf = 0; lf < 4; ++lf)
        if (face_compress[fl[lf]] < 0)
            for (int kp = 0; kp < FBS; ++kp) {
                const double dd = g != nullptr ? g[(size_t)fl[lf] * FBS + kp] : 0.0;
                s -= A[(size_t)(CBS + lf * FBS + kp) * MS + row] * dd;
            }
    return s;
}

// Face rows: one wavefront per U faces at a time, lane e of a pass = entry e of the face's block of fbs rows.
template <int CBS, int FBS, int U>
__global__ __launch_bounds__(256) void asm_fill_faces_kernel(AsmDims d, uint32_t q_begin, uint32_t q_end, const uint32_t *__restrict__ cell_faces,
                                                             const int32_t *__restrict__ face_compress,
                                                             const CondFaceLean *__restrict__ lean, const uint32_t *__restrict__ colprefix,
                                                             const uint32_t *__restrict__ fprefix, const double *__restrict__ lc,
                                                             const double *__restrict__ g, double *__restrict__ values,
                                                             double *__restrict__ RHS)
{
    constexpr int MS = CBS + 4 * FBS, RMAX = 2 * CBS + 7 * FBS, EMAX = FBS * RMAX;
    constexpr int PASSES = (EMAX + 63) / 64;
    const uint32_t lane = threadIdx.x % 64u, wave = blockIdx.x * (256u / 64u) + threadIdx.x / 64u;
    CondFaceLean r[U];
    uint32_t cpre[U], fpre[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t q = q_begin + wave * U + u;
        on[u] = q < q_end;
        const uint32_t qq = on[u] ? q : 0u;
        r[u] = lean[qq];
        cpre[u] = colprefix[qq];
        fpre[u] = fprefix[qq];
    }
    const double *pa_[U][PASSES], *pb_[U][PASSES];
    size_t dst[U][PASSES];
    bool ok[U][PASSES], two[U][PASSES];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t ncol = (uint32_t)(r[u].packed >> 46) & 7u, rows = (uint32_t)(r[u].packed >> 42) & 15u;
        const bool hasA_ = r[u].cA >= 0, hasB_ = r[u].cB >= 0;
        const uint32_t ncell = (uint32_t)hasA_ + (uint32_t)hasB_;
        const uint32_t R = ncell * CBS + ncol * FBS;
        const uint64_t start = face_block_start(d, fpre[u], cpre[u]);
        const double *LA = lc + (size_t)(hasA_ ? r[u].cA : 0) * (MS * MS), *LB = lc + (size_t)(hasB_ ? r[u].cB : 0) * (MS * MS);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const uint32_t e = lane + 64u * p;
            ok[u][p] = on[u] && e < (uint32_t)FBS * R;
            const uint32_t ee = ok[u][p] ? e : 0u;
            const uint32_t k = ee / R, jj = ee - k * R;
            const uint32_t rowA = CBS + (rows & 3u) * FBS + k, rowB = CBS + ((rows >> 2) & 3u) * FBS + k;
            const double *pA = LA, *pB = LB;
            bool useA = false, useB = false;
            if (jj < ncell * CBS) {
                const bool second = jj >= (uint32_t)CBS;               // the second cell block (only with two cells)
                const uint32_t j = second ? jj - CBS : jj;
                const bool fromA = hasA_ && !second;
                useA = fromA; useB = !fromA;
                pA = LA + (size_t)j * MS + rowA;
                pB = LB + (size_t)j * MS + rowB;
            } else {
                const uint32_t s = (jj - ncell * CBS) / (uint32_t)FBS, kp = (jj - ncell * CBS) % (uint32_t)FBS;
                const uint32_t code = (uint32_t)(r[u].packed >> (6 * s)) & 63u;
                useA = (code & 4u) != 0; useB = (code & 32u) != 0;
                pA = LA + (size_t)(CBS + (code & 3u) * FBS + kp) * MS + rowA;
                pB = LB + (size_t)(CBS + ((code >> 3) & 3u) * FBS + kp) * MS + rowB;
            }
            useA = useA && ok[u][p]; useB = useB && ok[u][p];
            pa_[u][p] = useA ? pA : (useB ? pB : lc);
            pb_[u][p] = useB ? pB : lc;
            two[u][p] = useA && useB;
            dst[u][p] = (size_t)start + e;
        }
    }
    double va[U][PASSES], vb[U][PASSES];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < PASSES; ++p) { va[u][p] = *pa_[u][p]; vb[u][p] = *pb_[u][p]; }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
            if (ok[u][p]) values[dst[u][p]] = two[u][p] ? va[u][p] + vb[u][p] : va[u][p];
    if (RHS != nullptr) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (on[u] && lane < (uint32_t)FBS) {
                const uint32_t rows = (uint32_t)(r[u].packed >> 42) & 15u;
                const bool dcols = (r[u].packed >> 49) & 1u;
                const int rowA = CBS + (int)(rows & 3u) * FBS + (int)lane, rowB = CBS + (int)((rows >> 2) & 3u) * FBS + (int)lane;
                double b = 0.0;
                bool have = false;
                if (r[u].cA >= 0) { b = dcols ? asm_face_rhs_contrib<CBS, FBS>(cell_faces, face_compress, lc, g, r[u].cA, rowA) : 0.0; have = true; }
                if (r[u].cB >= 0) {
                    const double w = dcols ? asm_face_rhs_contrib<CBS, FBS>(cell_faces, face_compress, lc, g, r[u].cB, rowB) : 0.0;
                    b = have ? b + w : w;
                }
                RHS[(size_t)d.cbs * d.ncells + (size_t)(q_begin + wave * U + u) * FBS + lane] = b;
            }
        }
    }
}

// cells / faces a wavefront has in flight at a time (their descriptors, then their gathers)
#ifndef PA_ASM_UNROLL
#define PA_ASM_UNROLL 2
#endif
// lc of the cells of one piece of the numeric phase (see asm_fill_t).  One piece: the pieces were meant to keep a piece's lc in
// the Infinity Cache between its cell rows and its face rows (lc fetched from HBM once instead of 1.7 times) -- measured SLOWER at
// 1024 x 1024: k = 2 3.35 ms in one piece, 3.68 in pieces of 96 MB, 4.87 in pieces of 32 MB (k = 3: 6.3 / 7.4 / 10.3): the tails of
// the extra launches cost more than the second fetch
#ifndef PA_ASM_PIECE_BYTES
#define PA_ASM_PIECE_BYTES ((size_t)1 << 40)
#endif
template <int CBS, int FBS>
static hipError_t asm_fill_t(hipStream_t stream, const CondMesh &m, const AsmDims &d, const CondFaceLean *lean, const uint32_t *colprefix,
                             const uint32_t *cprefix, const uint32_t *fprefix, const double *lc, const double *rhs, const double *g,
                             double *values, double *RHS)
{
    constexpr int U = PA_ASM_UNROLL;
    // (in pieces of about PA_ASM_PIECE_BYTES of lc -- the cell rows of a piece, then the face rows of the same fraction of the
    // compressed faces; one piece by default, see above)
    constexpr size_t lc_bytes = (size_t)(CBS + 4 * FBS) * (CBS + 4 * FBS) * 8;
    const size_t piece_cells = PA_ASM_PIECE_BYTES / lc_bytes ? PA_ASM_PIECE_BYTES / lc_bytes : 1;
    const uint32_t npieces = (uint32_t)((d.ncells + piece_cells - 1) / piece_cells);
    for (uint32_t p = 0; p < npieces; ++p) {
        const uint32_t c0 = (uint32_t)((uint64_t)d.ncells * p / npieces), c1 = (uint32_t)((uint64_t)d.ncells * (p + 1) / npieces);
        const uint32_t q0 = (uint32_t)((uint64_t)d.nown * p / npieces), q1 = (uint32_t)((uint64_t)d.nown * (p + 1) / npieces);
        if (c1 > c0)
            hipLaunchKernelGGL((asm_fill_cells_kernel<CBS, FBS, U>), dim3(((size_t)(c1 - c0) + 4 * U - 1) / (4 * U)), dim3(256), 0, stream, d, c0, c1,
                               m.cell_faces, m.face_compress, cprefix, lc, rhs, g, values, RHS);
        if (q1 > q0)
            hipLaunchKernelGGL((asm_fill_faces_kernel<CBS, FBS, U>), dim3(((size_t)(q1 - q0) + 4 * U - 1) / (4 * U)), dim3(256), 0, stream, d, q0, q1,
                               m.cell_faces, m.face_compress, lean, colprefix, fprefix, lc, g, values, RHS);
    }
    return hipGetLastError();
}

hipError_t asm_fill(hipStream_t stream, const CondMesh &m, int cbs, int fbs, uint32_t ncells, uint32_t nown, uint64_t cell_nnz,
                    const CondFaceLean *lean, const uint32_t *colprefix, const uint32_t *cprefix, const uint32_t *fprefix,
                    const double *lc, const double *rhs, const double *g, double *values, double *RHS)
{
    if (ncells == 0) return hipSuccess;
    const AsmDims d = {cbs, fbs, cbs + 4 * fbs, ncells, nown, cell_nnz};
#define PA_ASM_CASE(C, F) if (cbs == C && fbs == F) return asm_fill_t<C, F>(stream, m, d, lean, colprefix, cprefix, fprefix, lc, rhs, g, values, RHS)
    PA_ASM_CASE(1, 1); PA_ASM_CASE(3, 1); PA_ASM_CASE(1, 2); PA_ASM_CASE(3, 2); PA_ASM_CASE(6, 2); PA_ASM_CASE(3, 3); PA_ASM_CASE(6, 3);
    PA_ASM_CASE(10, 3); PA_ASM_CASE(6, 4); PA_ASM_CASE(10, 4); PA_ASM_CASE(15, 4);
#undef PA_ASM_CASE
    return hipErrorInvalidValue;
}

}  // namespace pa
