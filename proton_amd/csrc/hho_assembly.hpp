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
#include "scan.hpp"
#include "structured_mesh.hpp"

namespace pa {

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

// assembler::take_local_data (hho.hpp:408-449): one thread per (cell, local dof)
struct TakeArgs {
    const uint32_t *cell_faces; const int32_t *face_compress; const double *g;
    const double *solution;        // global vector (compressed faces)   -- or the expanded one
    size_t first, n; uint64_t cell_base, ncells_global, face_base;
    int cbs, fbs, expanded;        // expanded: faces at cbs*ncells + face_id*fbs (hho.hpp:753-782)
    double *out;
};

__global__ __launch_bounds__(256) void take_local_data_kernel(TakeArgs a)
{
    const int msize = a.cbs + 4 * a.fbs;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n * msize) return;
    const size_t c = t / msize, cl = a.first + c;
    const int i = (int)(t % msize);
    double v;
    if (i < a.cbs) {
        v = a.solution[(a.cell_base + cl) * a.cbs + i];
    } else {
        const int f = (i - a.cbs) / a.fbs, k = (i - a.cbs) % a.fbs;
        const uint32_t fl = a.cell_faces[4 * cl + f];
        if (a.expanded) {
            v = a.solution[a.cbs * a.ncells_global + (a.face_base + fl) * a.fbs + k];
        } else {
            const int32_t comp = a.face_compress[fl];
            v = comp < 0 ? (a.g ? a.g[(size_t)fl * a.fbs + k] : 0.0)
                         : a.solution[a.cbs * a.ncells_global + (uint64_t)comp * a.fbs + k];
        }
    }
    a.out[t] = v;
}

// Face part of project_function (utils.hpp:216-222): for local face lf of a cell,
// make_mass_matrix(fc, facdeg, di).llt().solve(make_rhs(fc, facdeg, f, di)).  At the Gauss point
// t_q of the face (parameter running from the lower-id endpoint, bases.hpp:255-272) the basis is
// t_q^k, and |F|/2 cancels between the two sides.  One thread per (cell, local face).
template <int FD>
__global__ __launch_bounds__(256) void face_project_kernel(const QuadTables *tab, const double *points,
                                                           const uint32_t *face_pts, const uint32_t *cell_faces,
                                                           size_t first, size_t n, int nfq, int fn, const double *fvals,
                                                           double *out, int out_stride, int out_offset)
{
    constexpr int FBS = FD + 1;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * n) return;
    const size_t c = t / 4; const int lf = (int)(t % 4);
    const uint32_t f = cell_faces[4 * (first + c) + lf];
    const double2 a = *reinterpret_cast<const double2 *>(points + 2 * (size_t)face_pts[2 * f]);
    const double2 b = *reinterpret_cast<const double2 *>(points + 2 * (size_t)face_pts[2 * f + 1]);
    double M[FBS][FBS], x[FBS];
#pragma unroll
    for (int i = 0; i < FBS; ++i) {
        x[i] = 0.0;
#pragma unroll
        for (int j = 0; j < FBS; ++j) M[i][j] = 0.0;
    }
    for (int q = 0; q < nfq; ++q) {
        const double tq = tab->gauss_x[nfq][q], wq = tab->gauss_w[nfq][q];
        const double px = 0.5 * (1 - tq) * a.x + 0.5 * (1 + tq) * b.x;
        const double py = 0.5 * (1 - tq) * a.y + 0.5 * (1 + tq) * b.y;
        const double fv = fn == FN_SAMPLED ? fvals[(size_t)f * nfq + q] : builtin_fn(fn, px, py);
        double phi[FBS];
        phi[0] = 1.0;
#pragma unroll
        for (int k = 1; k < FBS; ++k) phi[k] = phi[k - 1] * tq;
#pragma unroll
        for (int i = 0; i < FBS; ++i) {
            x[i] += wq * phi[i] * fv;
#pragma unroll
            for (int j = 0; j <= i; ++j) M[i][j] += wq * phi[i] * phi[j];
        }
    }
#pragma unroll
    for (int j = 0; j < FBS; ++j) {
        double d = M[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= M[j][k] * M[j][k];
        const double r = 1.0 / sqrt(d);
        M[j][j] = r;
#pragma unroll
        for (int i = j + 1; i < FBS; ++i) {
            double s = M[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= M[i][k] * M[j][k];
            M[i][j] = s * r;
        }
    }
#pragma unroll
    for (int i = 0; i < FBS; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= M[i][k] * x[k];
        x[i] = s * M[i][i];
    }
#pragma unroll
    for (int i = FBS - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < FBS; ++k) s -= M[k][i] * x[k];
        x[i] = s * M[i][i];
    }
#pragma unroll
    for (int k = 0; k < FBS; ++k) out[c * out_stride + out_offset + lf * FBS + k] = x[k];
}

// diff.dot(lc * diff) per cell (the energy error of convergence_test.cpp / obstacle.cpp:202-213):
// one wavefront per cell, lc read once with coalesced loads.
__global__ __launch_bounds__(64) void energy_form_kernel(size_t n, int msize, const double *lc, const double *u,
                                                         const double *v, double *out)
{
    __shared__ double d[64];
    const int l = threadIdx.x;
    for (size_t c = blockIdx.x; c < n; c += gridDim.x) {
        if (l < msize) d[l] = u[c * msize + l] - (v ? v[c * msize + l] : 0.0);
        __syncthreads();
        const double *A = lc + c * (size_t)(msize * msize);
        double s = 0.0;
        for (int e = l; e < msize * msize; e += 64) s += A[e] * d[e % msize] * d[e / msize];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (l == 0) out[c] = s;
        __syncthreads();
    }
}

// ---- obstacle_assembler (hho.hpp:471-751) ----------------------------------------------------
// compress tables A_ct / B_ct (hho.hpp:538-578) = exclusive prefix counts of the cells outside /
// inside the active set: the three-pass scan of scan.hpp.
struct ObstacleArgs {
    TripletArgs t;                 // rows/cols/vals hold msize^2 + 1 slots per cell (the multiplier coupling last)
    const uint8_t *in_A;
    const int32_t *A_ct, *B_ct;
    const double *gamma;           // ncells x cbs... the reference indexes gamma(cell_offset): one value per cell
    uint64_t num_I, num_other;
};

// obstacle_assembler::assemble (hho.hpp:609-695): one block per cell, one thread per (i, j) slot.
__global__ __launch_bounds__(256) void obstacle_triplets_kernel(ObstacleArgs o)
{
    extern __shared__ double sh[];                    // dirichlet data (msize), then int32 row (msize), col (msize)
    const TripletArgs &a = o.t;
    const int msize = a.cbs + 4 * a.fbs, slots = msize * msize + 1;
    double *dd = sh;
    int32_t *row = reinterpret_cast<int32_t *>(sh + msize), *col = row + msize;
    for (size_t c = blockIdx.x; c < a.n; c += gridDim.x) {
        const size_t cl = a.first + c;
        const bool active = o.in_A[cl] != 0;
        for (int i = threadIdx.x; i < msize; i += blockDim.x) {
            int32_t r, q; double d = 0.0;
            if (i < a.cbs) {
                r = (int32_t)(cl + i);                                                    // hho.hpp:631 (no * cbs)
                q = active ? -1 : (int32_t)((uint64_t)o.A_ct[cl] * a.cbs + i);            // :625,632
            } else {
                const int f = (i - a.cbs) / a.fbs, k = (i - a.cbs) % a.fbs;
                const uint32_t fl = a.cell_faces[4 * cl + f];
                const int32_t comp = a.face_compress[fl];
                r = comp < 0 ? -1 : (int32_t)(a.cbs * a.ncells_global + (uint64_t)comp * a.fbs + k);   // :644
                q = comp < 0 ? -1 : (int32_t)(a.cbs * o.num_I + (uint64_t)comp * a.fbs + k);           // :645
                if (comp < 0 && a.g != nullptr) d = a.g[(size_t)fl * a.fbs + k];                       // :655-660
            }
            row[i] = r; col[i] = q; dd[i] = d;
        }
        __syncthreads();
        const double *A = a.lc + c * (size_t)(msize * msize);
        const double gam = o.gamma[cl];
        for (int e = threadIdx.x; e < slots; e += blockDim.x) {
            const size_t dst = c * (size_t)slots + e;
            if (e == slots - 1) {                                                         // :688-693
                a.rows[dst] = active ? (int32_t)(cl * a.cbs) : -1;
                a.cols[dst] = active ? (int32_t)(o.num_I * a.cbs + o.num_other * a.fbs + (uint64_t)o.B_ct[cl]) : -1;
                a.vals[dst] = 1.0;
            } else {
                const int i = e / msize, j = e % msize;
                const bool keep = row[i] >= 0 && col[j] >= 0;                             // :668,673
                a.rows[dst] = keep ? row[i] : -1;
                a.cols[dst] = keep ? col[j] : -1;
                a.vals[dst] = A[i + j * msize];
            }
        }
        for (int i = threadIdx.x; i < msize; i += blockDim.x) {
            double s = 0.0;
            if (row[i] >= 0)
                for (int j = 0; j < msize; ++j)
                    if (col[j] < 0) s -= A[i + j * msize] * (j < a.cbs ? gam : dd[j]);    // :676-679
            if (i < a.cbs && a.rhs != nullptr) s += a.rhs[c * a.cbs + i];                 // :686
            a.rhs_rows[c * msize + i] = row[i];
            a.rhs_vals[c * msize + i] = row[i] >= 0 ? s : 0.0;
        }
        __syncthreads();
    }
}

// obstacle_assembler::expand_solution (hho.hpp:698-744): thread t < ncells*cbs handles a cell dof
// of alpha and beta, the following nfaces*fbs threads the face dofs of alpha
struct ExpandArgs {
    const uint8_t *in_A, *face_dir; const int32_t *A_ct, *B_ct, *face_compress;
    const double *solution, *g, *gamma;
    uint64_t ncells, nfaces, num_I, num_other; int cbs, fbs;
    double *alpha, *beta;
};

__global__ __launch_bounds__(256) void obstacle_expand_kernel(ExpandArgs a)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ncd = a.ncells * a.cbs;
    if (t < ncd) {
        const uint64_t c = t / a.cbs; const int k = (int)(t % a.cbs);
        const bool active = a.in_A[c] != 0;
        a.alpha[t] = active ? a.gamma[t] : a.solution[(uint64_t)a.A_ct[c] * a.cbs + k];               // :709-714
        a.beta[t] = active ? a.solution[a.num_I * a.cbs + a.num_other * a.fbs + (uint64_t)a.B_ct[c] * a.cbs + k] : 0.0;   // :716-721
    } else if (t < ncd + a.nfaces * a.fbs) {
        const uint64_t u = t - ncd, f = u / a.fbs; const int k = (int)(u % a.fbs);
        const int32_t comp = a.face_compress[f];
        a.alpha[t] = comp < 0 ? (a.g ? a.g[u] : 0.0) : a.solution[a.cbs * a.num_I + (uint64_t)comp * a.fbs + k];   // :723-743
    }
}

}  // namespace pa
