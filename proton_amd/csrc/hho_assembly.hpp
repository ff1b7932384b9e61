// hho_assembly.hpp -- device side of assembler<Mesh> (src/methods/hho_bits/hho.hpp:252-463):
// face tables of the structured generator mesh in closed form (basic_mesh.hpp:266-297 sorts and
// uniques the (lo,hi) pairs: every point of a row owns a horizontal then a vertical face, the last
// point of the row a vertical only, the top row horizontals only), the Dirichlet data of boundary
// faces (hho.hpp:381-386) and the per-cell triplets / right-hand-side updates (hho.hpp:362-405).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hho_aux.hpp"
#include "hho_device.hpp"

namespace pa {

struct StructuredMesh {
    uint32_t Nx, Ny, row0, row1;      // the context owns cell rows [row0, row1)
};

__host__ __device__ inline uint32_t sm_face_row(const StructuredMesh &m) { return 2 * m.Nx + 1; }
// global ids of the faces of the generator mesh
__host__ __device__ inline uint32_t sm_hface(const StructuredMesh &m, uint32_t i, uint32_t j)
{
    return j < m.Ny ? j * sm_face_row(m) + 2 * i : m.Ny * sm_face_row(m) + i;
}
__host__ __device__ inline uint32_t sm_vface(const StructuredMesh &m, uint32_t i, uint32_t j)
{
    return j * sm_face_row(m) + (i < m.Nx ? 2 * i + 1 : 2 * m.Nx);
}
__host__ __device__ inline uint32_t sm_face_base(const StructuredMesh &m) { return m.row0 * sm_face_row(m); }
__host__ __device__ inline uint32_t sm_faces_local(const StructuredMesh &m)
{
    // rows row0..row1-1 in full, plus the horizontals that close the slab on top (they live in
    // row row1's block, or in the top row)
    return (m.row1 - m.row0) * sm_face_row(m) + (m.row1 < m.Ny ? sm_face_row(m) : m.Nx);
}
__host__ __device__ inline uint32_t sm_num_other_faces(const StructuredMesh &m)
{
    return m.Nx * (m.Ny + 1) + m.Ny * (m.Nx + 1) - 2 * (m.Nx + m.Ny);
}

// decode a global face id: endpoints (global point ids, lo < hi), Dirichlet flag (every boundary
// face, basic_mesh.hpp:293-297) and the compress-table value (hho.hpp:313-323) in closed form
__host__ __device__ inline void sm_face_decode(const StructuredMesh &m, uint32_t gid, uint32_t &lo, uint32_t &hi,
                                               bool &dirichlet, int32_t &compress)
{
    const uint32_t row = sm_face_row(m), npr = m.Nx + 1;
    if (gid >= m.Ny * row) {                         // top row: horizontals only, all on the boundary
        const uint32_t i = gid - m.Ny * row;
        lo = m.Ny * npr + i; hi = lo + 1; dirichlet = true; compress = -1;
        return;
    }
    const uint32_t jj = gid / row, pos = gid % row;
    // non-Dirichlet faces in the rows below: row 0 has Nx-1 interior verticals, every other row
    // Nx horizontals and Nx-1 interior verticals
    uint32_t cnt = jj >= 1 ? (m.Nx - 1) + (jj - 1) * (2 * m.Nx - 1) : 0;
    // ... and in this row at positions < pos: horizontals sit at even positions 2i, verticals at
    // 2i+1 (i < Nx) and at 2Nx (i = Nx)
    const uint32_t nh = (pos + 1) / 2 < m.Nx ? (pos + 1) / 2 : m.Nx;
    if (jj > 0) cnt += nh;
    const uint32_t nv_all = pos / 2;                 // verticals i = 0 .. nv_all-1 lie before pos
    uint32_t nv_int = nv_all > 0 ? nv_all - 1 : 0;   // i = 0 is on the boundary
    if (nv_int > m.Nx - 1) nv_int = m.Nx - 1;
    cnt += nv_int;
    const bool horizontal = (pos % 2 == 0) && pos < 2 * m.Nx;
    if (horizontal) {
        const uint32_t i = pos / 2;
        lo = jj * npr + i; hi = lo + 1;
        dirichlet = jj == 0;
    } else {
        const uint32_t i = pos == 2 * m.Nx ? m.Nx : pos / 2;
        lo = jj * npr + i; hi = lo + npr;
        dirichlet = (i == 0) || (i == m.Nx);
    }
    compress = dirichlet ? -1 : (int32_t)cnt;
}

// face tables of the slab: local face index = global id - face_base; point ids local to the slab
__global__ void structured_faces_kernel(StructuredMesh m, uint32_t nfaces_local, uint32_t *face_pts, uint8_t *face_dir,
                                        int32_t *face_compress, uint32_t ncells_local, uint32_t *cell_faces)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t fbase = sm_face_base(m), pbase = m.row0 * (m.Nx + 1);
    if (t < nfaces_local) {
        uint32_t lo, hi; bool d; int32_t comp;
        sm_face_decode(m, fbase + t, lo, hi, d, comp);
        // faces of row row1's block that are not the slab's top horizontals reference points outside the slab
        const uint32_t np_local = (m.Nx + 1) * (m.row1 - m.row0 + 1);
        const bool inside = lo >= pbase && hi - pbase < np_local;
        face_pts[2 * t] = inside ? lo - pbase : 0;
        face_pts[2 * t + 1] = inside ? hi - pbase : 0;
        face_dir[t] = d ? 1 : 0;
        face_compress[t] = comp;
    }
    if (t < ncells_local) {
        const uint32_t ci = t % m.Nx, cj = m.row0 + t / m.Nx;
        cell_faces[4 * t + 0] = sm_hface(m, ci, cj) - fbase;           // bottom, right, top, left
        cell_faces[4 * t + 1] = sm_vface(m, ci + 1, cj) - fbase;       // basic_geom.hpp:194-203
        cell_faces[4 * t + 2] = sm_hface(m, ci, cj + 1) - fbase;
        cell_faces[4 * t + 3] = sm_vface(m, ci, cj) - fbase;
    }
}

// Dirichlet data of every boundary face: mass.llt().solve(rhs) of the boundary function
// (hho.hpp:383-385).  M_F = (|F|/2) M^ and rhs_k = (|F|/2) sum_q w_q t_q^k f(x_q): the length
// cancels, g = M^^-1 sum_q w_q t_q^k f(x_q).  One thread per face; zeros for other faces.
template <int FD>
__global__ __launch_bounds__(256) void dirichlet_data_kernel(const QuadTables *tab, const double *points,
                                                             const uint32_t *face_pts, const uint8_t *face_dir,
                                                             uint32_t nfaces, int fn, const double *fvals, double *g)
{
    constexpr int FBS = FD + 1, NFQ = FD + 1;          // integrate(msh, fc, 2*facdeg): facdeg+1 nodes
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfaces) return;
    double x[FBS];
#pragma unroll
    for (int k = 0; k < FBS; ++k) x[k] = 0.0;
    if (face_dir[t]) {
        const double2 a = *reinterpret_cast<const double2 *>(points + 2 * (size_t)face_pts[2 * t]);
        const double2 b = *reinterpret_cast<const double2 *>(points + 2 * (size_t)face_pts[2 * t + 1]);
        const FaceTables &ft = tab->face[FD];
#pragma unroll
        for (int q = 0; q < NFQ; ++q) {
            const double tq = tab->gauss_x[NFQ][q];
            const double px = 0.5 * (1 - tq) * a.x + 0.5 * (1 + tq) * b.x;    // quadratures.hpp:420-428
            const double py = 0.5 * (1 - tq) * a.y + 0.5 * (1 + tq) * b.y;
            const double fv = fn == FN_SAMPLED ? fvals[(size_t)t * NFQ + q] : builtin_fn(fn, px, py);
#pragma unroll
            for (int k = 0; k < FBS; ++k) x[k] += ft.cw[q][k] * fv;
        }
#pragma unroll
        for (int i = 0; i < FBS; ++i) {                // L^ y = b
            double s = x[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= ft.lf[i][k] * x[k];
            x[i] = s * ft.lf[i][i];
        }
#pragma unroll
        for (int i = FBS - 1; i >= 0; --i) {           // L^^T g = y
            double s = x[i];
#pragma unroll
            for (int k = i + 1; k < FBS; ++k) s -= ft.lf[k][i] * x[k];
            x[i] = s * ft.lf[i][i];
        }
    }
#pragma unroll
    for (int k = 0; k < FBS; ++k) g[(size_t)t * FBS + k] = x[k];
}

// face quadrature points (x, y, w) in the reference's order, for caller-sampled boundary data
__global__ void face_qpoints_kernel(const QuadTables *tab, const double *points, const uint32_t *face_pts,
                                    uint32_t nfaces, int nfq, double *xyw)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfaces) return;
    const double2 a = *reinterpret_cast<const double2 *>(points + 2 * (size_t)face_pts[2 * t]);
    const double2 b = *reinterpret_cast<const double2 *>(points + 2 * (size_t)face_pts[2 * t + 1]);
    const double len = sqrt((b.x - a.x) * (b.x - a.x) + (b.y - a.y) * (b.y - a.y));
    for (int q = 0; q < nfq; ++q) {
        const double tq = tab->gauss_x[nfq][q];
        double *dst = xyw + ((size_t)t * nfq + q) * 3;
        dst[0] = 0.5 * (1 - tq) * a.x + 0.5 * (1 + tq) * b.x;
        dst[1] = 0.5 * (1 - tq) * a.y + 0.5 * (1 + tq) * b.y;
        dst[2] = tab->gauss_w[nfq][q] * len * 0.5;
    }
}

struct TripletArgs {
    const uint32_t *cell_faces;    // n_local x 4, local face indices
    const uint8_t *face_dir;
    const int32_t *face_compress;  // global compress value, -1 for Dirichlet
    const double *g;               // nfaces_local x fbs Dirichlet data (may be null: homogeneous)
    const double *lc;              // n x msize^2 (column-major per cell)
    const double *rhs;             // n x cbs (may be null)
    size_t first, n;
    uint64_t cell_base, ncells_global;
    int cbs, fbs;
    int32_t *rows, *cols;          // n x msize^2, slot i*msize + j (the reference's push order), -1 = not assembled
    double *vals;
    int32_t *rhs_rows;             // n x msize, -1 = Dirichlet row
    double *rhs_vals;
};

// assembler::assemble (hho.hpp:344-406): one block per cell, one thread per (i, j) slot.
__global__ __launch_bounds__(256) void triplets_kernel(TripletArgs a)
{
    extern __shared__ double sh[];                    // dirichlet data (msize), then int32 idx (msize)
    const int msize = a.cbs + 4 * a.fbs;
    double *dd = sh;
    int32_t *idx = reinterpret_cast<int32_t *>(sh + msize);
    for (size_t c = blockIdx.x; c < a.n; c += gridDim.x) {
        const size_t cl = a.first + c;
        for (int i = threadIdx.x; i < msize; i += blockDim.x) {
            int32_t gi; double d = 0.0;
            if (i < a.cbs) {
                gi = (int32_t)((a.cell_base + cl) * a.cbs + i);                           // hho.hpp:362-366
            } else {
                const int f = (i - a.cbs) / a.fbs, k = (i - a.cbs) % a.fbs;
                const uint32_t fl = a.cell_faces[4 * cl + f];
                const int32_t comp = a.face_compress[fl];
                gi = comp < 0 ? -1 : (int32_t)(a.cbs * a.ncells_global + (uint64_t)comp * a.fbs + k);   // :374-379
                if (comp < 0 && a.g != nullptr) d = a.g[(size_t)fl * a.fbs + k];          // :381-386
            }
            idx[i] = gi; dd[i] = d;
        }
        __syncthreads();
        const double *A = a.lc + c * (size_t)(msize * msize);
        for (int e = threadIdx.x; e < msize * msize; e += blockDim.x) {
            const int i = e / msize, j = e % msize;
            const bool keep = idx[i] >= 0 && idx[j] >= 0;                                 // :393,398
            const size_t o = c * (size_t)(msize * msize) + e;
            a.rows[o] = keep ? idx[i] : -1;
            a.cols[o] = keep ? idx[j] : -1;
            a.vals[o] = A[i + j * msize];
        }
        for (int i = threadIdx.x; i < msize; i += blockDim.x) {
            double s = (i < a.cbs && a.rhs != nullptr) ? a.rhs[c * a.cbs + i] : 0.0;      // :405
            if (idx[i] >= 0)
                for (int j = a.cbs; j < msize; ++j)
                    if (idx[j] < 0) s -= A[i + j * msize] * dd[j];                        // :401
            a.rhs_rows[c * msize + i] = idx[i];
            a.rhs_vals[c * msize + i] = idx[i] >= 0 ? s : 0.0;
        }
        __syncthreads();
    }
}

}  // namespace pa
