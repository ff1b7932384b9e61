// hho_aux.hpp -- the smaller kernels around the local-operator kernel:
//   structured mesh generation   (src/core/core_bits/basic_mesh.hpp:230-298)
//   cell quadrature points       (src/core/core_bits/quadratures.hpp:311-402)
//   cell right-hand sides        (src/core/core_bits/utils.hpp:153-174)
//   static condensation          (not in the reference)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hho_device.hpp"

#ifndef PA_SC_FENCE_MIN
#define PA_SC_FENCE_MIN 11
#endif
#ifndef PA_SC_INPLACE
#define PA_SC_INPLACE 1
#endif
namespace pa {

enum { FN_SAMPLED = 0, FN_SIN_SIN_RHS = 1, FN_SIN_SIN_SOL = 2, FN_OBSTACLE_RHS = 3, FN_OBSTACLE_SOL = 4, FN_ONE = 5 };

// mesh_impl<T,4>(mesh_init_params): points (min + i*hx, min + j*hy), point id j*(Nx+1)+i;
// cells {p, p+1, p+Nx+2, p+Nx+1}, cell id j*Nx+i (basic_mesh.hpp:239-264; the sort at :289 is the
// identity).  The context holds rows [row0, row1): point ids are local to the slab, which keeps
// their relative order (all the face-basis orientation depends on).
__global__ void mesh_generate_kernel(double *points, uint32_t *ptids, size_t Nx, size_t row0, size_t row1,
                                     double min_x, double hx, double min_y, double hy)
{
    const size_t npr = Nx + 1;
    const size_t np = npr * (row1 - row0 + 1), nc = Nx * (row1 - row0);
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < np; t += (size_t)gridDim.x * blockDim.x) {
        const size_t j = t / npr, i = t % npr;
        points[2 * t] = min_x + (double)i * hx;                    // parms.min_x + i*hx  :243
        points[2 * t + 1] = min_y + (double)(row0 + j) * hy;
        if (t < nc) {
            const size_t cj = t / Nx, ci = t % Nx;
            const uint32_t p0 = (uint32_t)(cj * npr + ci);
            ptids[4 * t + 0] = p0;
            ptids[4 * t + 1] = p0 + 1;
            ptids[4 * t + 2] = p0 + (uint32_t)Nx + 2;
            ptids[4 * t + 3] = p0 + (uint32_t)Nx + 1;
        }
    }
}

struct CellGeom {
    double px[4], py[4], barx, bary, hT;
};

__device__ __forceinline__ void load_cell_geom(const double *points, const uint32_t *ptids, size_t cell, CellGeom &c)
{
    const uint4 idv = *reinterpret_cast<const uint4 *>(ptids + 4 * cell);
    const uint32_t ids[4] = {idv.x, idv.y, idv.z, idv.w};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const double2 pt = *reinterpret_cast<const double2 *>(points + 2 * (size_t)ids[v]);
        c.px[v] = pt.x; c.py[v] = pt.y;
    }
    double rx = 0.0, ry = 0.0, den = 0.0;                           // basic_geom.hpp:247-270
#pragma unroll
    for (int i = 2; i < 4; ++i) {
        const double ax = c.px[i - 1] - c.px[0], ay = c.py[i - 1] - c.py[0];
        const double bx = c.px[i] - c.px[0], by = c.py[i] - c.py[0];
        const double d = (ax * by - ay * bx) / 2.0;
        rx += (ax + bx) * d; ry += (ay + by) * d; den += d;
    }
    c.barx = c.px[0] + rx / (den * 3); c.bary = c.py[0] + ry / (den * 3);
    double h = 0.0;                                                 // basic_geom.hpp:288-305
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const double dx = c.px[j] - c.px[i], dy = c.py[j] - c.py[i];
            h = fmax(h, sqrt(dx * dx + dy * dy));
        }
    c.hT = h;
}

// q-th point of integrate(msh, cl, degree): tensor Gauss (quadratures.hpp:311-375) or fan (:377-402)
template <int QUAD>
__device__ __forceinline__ void cell_qp(const QuadTables *tab, const CellGeom &c, int degree, int q,
                                        double &x, double &y, double &w)
{
    if (QUAD == QUAD_TENSOR) {
        const int n = gauss_nodes(degree);
        const int i = q % n, j = q / n;
        const double xi = tab->gauss_x[n][i], eta = tab->gauss_x[n][j];
        x = 0.25 * c.px[0] * (1 - xi) * (1 - eta) + 0.25 * c.px[1] * (1 + xi) * (1 - eta) +
            0.25 * c.px[2] * (1 + xi) * (1 + eta) + 0.25 * c.px[3] * (1 - xi) * (1 + eta);
        y = 0.25 * c.py[0] * (1 - xi) * (1 - eta) + 0.25 * c.py[1] * (1 + xi) * (1 - eta) +
            0.25 * c.py[2] * (1 + xi) * (1 + eta) + 0.25 * c.py[3] * (1 - xi) * (1 + eta);
        const double j11 = 0.25 * ((c.px[1] - c.px[0]) * (1 - eta) + (c.px[2] - c.px[3]) * (1 + eta));
        const double j12 = 0.25 * ((c.py[1] - c.py[0]) * (1 - eta) + (c.py[2] - c.py[3]) * (1 + eta));
        const double j21 = 0.25 * ((c.px[3] - c.px[0]) * (1 - xi) + (c.px[2] - c.px[1]) * (1 + xi));
        const double j22 = 0.25 * ((c.py[3] - c.py[0]) * (1 - xi) + (c.py[2] - c.py[1]) * (1 + xi));
        w = tab->gauss_w[n][i] * tab->gauss_w[n][j] * fabs(j11 * j22 - j12 * j21);
    } else {
        const int R = degree == 0 ? 1 : degree;                     // rules[deg]  quadratures.hpp:242-257
        const int nt = tab->dun_n[R];
        const int t = q / nt, row = q % nt, t1 = (t + 1) & 3;
        const double ax = c.px[t], ay = c.py[t], bx = c.px[t1], by = c.py[t1];
        const double v0x = bx - ax, v0y = by - ay, v1x = c.barx - ax, v1y = c.bary - ay;
        const double tarea = fabs((v0x * v1y - v0y * v1x) / 2.0);
        const double l0 = tab->dun[R][row][0], l1 = tab->dun[R][row][1], l2 = tab->dun[R][row][2];
        x = ax * l0 + bx * l1 + c.barx * l2;
        y = ay * l0 + by * l1 + c.bary * l2;
        w = tarea * tab->dun[R][row][3];
    }
}

__host__ __device__ inline int cell_qp_count(const QuadTables *tab, int quad, int degree)
{
    if (quad == QUAD_TENSOR) { const int n = gauss_nodes(degree); return n * n; }
    return 4 * tab->dun_n[degree == 0 ? 1 : degree];
}

__device__ __forceinline__ double builtin_fn(int fn, double x, double y)
{
    const double pi = 3.14159265358979323846;                       // M_PI
    switch (fn) {
    // sinpi(x): sin(pi x) without the product's rounding and the general argument reduction of sin() -- within 1 ulp of
    // the exact value, 2e-16 absolute from the reference's std::sin(M_PI * x); the two calls are most of make_rhs
    case FN_SIN_SIN_RHS: return 2.0 * pi * pi * sinpi(x) * sinpi(y);         // convergence_test.cpp:100-102
    case FN_SIN_SIN_SOL: return sinpi(x) * sinpi(y);                         // convergence_test.cpp:104-106
    case FN_OBSTACLE_RHS: {                                                  // obstacle.cpp:65-74
        const double r0 = 0.7, r = sqrt(x * x + y * y);
        return r > r0 ? -16 * r * r + 8 * r0 * r0 : -8.0 * (r0 * r0 * (r0 * r0 + 1)) + 8 * r0 * r0 * r * r;
    }
    case FN_OBSTACLE_SOL: {                                                  // obstacle.cpp:76-81
        const double r0 = 0.7, r = sqrt(x * x + y * y);
        const double s = r * r - r0 * r0, t = fmax(s, 0.0);
        return t * t;
    }
    default: return 1.0;
    }
}

template <int QUAD>
__global__ __launch_bounds__(256) void cell_qpoints_kernel(const QuadTables *tab, const double *points,
                                                           const uint32_t *ptids, size_t first, size_t n,
                                                           int degree, int nqp, double *xyw)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    CellGeom c;
    load_cell_geom(points, ptids, first + t, c);
    for (int q = 0; q < nqp; ++q) {
        double x, y, w;
        cell_qp<QUAD>(tab, c, degree, q, x, y, w);
        double *dst = xyw + (t * nqp + q) * 3;
        dst[0] = x; dst[1] = y; dst[2] = w;
    }
}

// make_rhs(msh, cl, degree, f, di): ret += qp.second * phi * f(qp.first)   utils.hpp:163-171.
// One thread per cell; DEG is compile time so the accumulators stay in registers.
template <int DEG, int QUAD>
__global__ __launch_bounds__(256) void cell_rhs_kernel(const QuadTables *tab, const double *points,
                                                       const uint32_t *ptids, size_t first, size_t n,
                                                       int qdegree, int nqp, int fn, const double *fvals,
                                                       double *rhs, const int8_t *cell_loc = nullptr, int where = 0)
{
    constexpr int CBS = P2(DEG);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    // fictitious domain (cut make_rhs, cuthho_square.cpp:628-629): a cell that is not on the `where` side -- outside the
    // domain, or cut: its right-hand side comes from the cut kernel -- gets zeros and costs nothing
    if (cell_loc != nullptr && cell_loc[first + t] != where) {
#pragma unroll
        for (int m = 0; m < CBS; ++m) rhs[t * CBS + m] = 0.0;
        return;
    }
    CellGeom c;
    load_cell_geom(points, ptids, first + t, c);
    const double ihalf = 1.0 / (0.5 * c.hT);
    double acc[CBS];
#pragma unroll
    for (int m = 0; m < CBS; ++m) acc[m] = 0.0;
    auto point = [&](int q, double x, double y, double w) {
        const double fv = (fn == FN_SAMPLED) ? fvals[t * nqp + q] : builtin_fn(fn, x, y);
        const double bx = (x - c.barx) * ihalf, by = (y - c.bary) * ihalf;
        double pwx[DEG + 1], pwy[DEG + 1];
        pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
        for (int e = 1; e <= DEG; ++e) { pwx[e] = pwx[e - 1] * bx; pwy[e] = pwy[e - 1] * by; }
        int m = 0;
#pragma unroll
        for (int kk = 0; kk <= DEG; ++kk)
#pragma unroll
            for (int ii = 0; ii <= kk; ++ii, ++m) acc[m] += (w * (pwx[kk - ii] * pwy[ii])) * fv;
    };
    if (QUAD == QUAD_TENSOR) {
        // outer eta, inner xi (quadratures.hpp:355-357): the point order of cell_qp without its division of q per point; the
        // terms that depend on eta alone are formed once per row
        const int ng = gauss_nodes(qdegree);
        int q = 0;
        for (int j = 0; j < ng; ++j) {
            const double eta = tab->gauss_x[ng][j], wj = tab->gauss_w[ng][j];
            for (int i = 0; i < ng; ++i, ++q) {
                const double xi = tab->gauss_x[ng][i];
                const double x = 0.25 * c.px[0] * (1 - xi) * (1 - eta) + 0.25 * c.px[1] * (1 + xi) * (1 - eta) +
                                 0.25 * c.px[2] * (1 + xi) * (1 + eta) + 0.25 * c.px[3] * (1 - xi) * (1 + eta);
                const double y = 0.25 * c.py[0] * (1 - xi) * (1 - eta) + 0.25 * c.py[1] * (1 + xi) * (1 - eta) +
                                 0.25 * c.py[2] * (1 + xi) * (1 + eta) + 0.25 * c.py[3] * (1 - xi) * (1 + eta);
                const double j11 = 0.25 * ((c.px[1] - c.px[0]) * (1 - eta) + (c.px[2] - c.px[3]) * (1 + eta));
                const double j12 = 0.25 * ((c.py[1] - c.py[0]) * (1 - eta) + (c.py[2] - c.py[3]) * (1 + eta));
                const double j21 = 0.25 * ((c.px[3] - c.px[0]) * (1 - xi) + (c.px[2] - c.px[1]) * (1 + xi));
                const double j22 = 0.25 * ((c.py[3] - c.py[0]) * (1 - xi) + (c.py[2] - c.py[1]) * (1 + xi));
                point(q, x, y, tab->gauss_w[ng][i] * wj * fabs(j11 * j22 - j12 * j21));
            }
        }
    } else {
        // fan triangles (p_t, p_t+1, barycenter), the Dunavant rows of rules[deg] inside (quadratures.hpp:390-396, 242-268): the
        // point order of cell_qp with the triangle's edge vectors and area formed once per triangle, not per point
        const int R = qdegree == 0 ? 1 : qdegree;
        const int nt = tab->dun_n[R];
        int q = 0;
#pragma unroll
        for (int tr = 0; tr < 4; ++tr) {      // (unrolled: the vertices are registers)
            const int t1 = (tr + 1) & 3;
            const double ax = c.px[tr], ay = c.py[tr], bx = c.px[t1], by = c.py[t1];
            const double v0x = bx - ax, v0y = by - ay, v1x = c.barx - ax, v1y = c.bary - ay;
            const double tarea = fabs((v0x * v1y - v0y * v1x) / 2.0);
            for (int row = 0; row < nt; ++row, ++q) {
                const double l0 = tab->dun[R][row][0], l1 = tab->dun[R][row][1], l2 = tab->dun[R][row][2];
                point(q, ax * l0 + bx * l1 + c.barx * l2, ay * l0 + by * l1 + c.bary * l2, tarea * tab->dun[R][row][3]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < CBS; ++m) rhs[t * CBS + m] = acc[m];
}

// Cell part of project_function(msh, cl, hdi, f, di) (utils.hpp:199-214): mass and right-hand
// side at degree 2*(celdeg + di), then mass.llt().solve(rhs).  One thread per cell (a postprocess
// kernel: the packed lower triangle of the mass matrix lives in per-thread memory).
template <int DEG, int QUAD>
__global__ __launch_bounds__(256) void cell_project_kernel(const QuadTables *tab, const double *points,
                                                           const uint32_t *ptids, size_t first, size_t n,
                                                           int qdegree, int nqp, int fn, const double *fvals,
                                                           double *out, int out_stride, int32_t *info)
{
    constexpr int CBS = P2(DEG);
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    CellGeom c;
    load_cell_geom(points, ptids, first + t, c);
    const double ihalf = 1.0 / (0.5 * c.hT);
    double b[CBS], M[CBS * (CBS + 1) / 2];
#pragma unroll
    for (int m = 0; m < CBS; ++m) b[m] = 0.0;
#pragma unroll
    for (int m = 0; m < CBS * (CBS + 1) / 2; ++m) M[m] = 0.0;
    for (int q = 0; q < nqp; ++q) {
        double x, y, w;
        cell_qp<QUAD>(tab, c, qdegree, q, x, y, w);
        const double fv = (fn == FN_SAMPLED) ? fvals[t * nqp + q] : builtin_fn(fn, x, y);
        const double bx = (x - c.barx) * ihalf, by = (y - c.bary) * ihalf;
        double pwx[DEG + 1], pwy[DEG + 1], phi[CBS];
        pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
        for (int e = 1; e <= DEG; ++e) { pwx[e] = pwx[e - 1] * bx; pwy[e] = pwy[e - 1] * by; }
        int m = 0;
#pragma unroll
        for (int kk = 0; kk <= DEG; ++kk)
#pragma unroll
            for (int ii = 0; ii <= kk; ++ii, ++m) phi[m] = pwx[kk - ii] * pwy[ii];
#pragma unroll
        for (int i = 0; i < CBS; ++i) {
            const double wp = w * phi[i];
            b[i] += wp * fv;
#pragma unroll
            for (int j = 0; j <= i; ++j) M[i * (i + 1) / 2 + j] += wp * phi[j];
        }
    }
    int bad = 0;
#pragma unroll
    for (int j = 0; j < CBS; ++j) {                                  // unpivoted lower Cholesky, like Eigen's LLT
        double d = M[j * (j + 1) / 2 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= M[j * (j + 1) / 2 + k] * M[j * (j + 1) / 2 + k];
        if (!(d > 0.0) && bad == 0) bad = j + 1;
        const double r = 1.0 / sqrt(d);
        M[j * (j + 1) / 2 + j] = r;                                  // reciprocal of the diagonal
#pragma unroll
        for (int i = j + 1; i < CBS; ++i) {
            double s = M[i * (i + 1) / 2 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= M[i * (i + 1) / 2 + k] * M[j * (j + 1) / 2 + k];
            M[i * (i + 1) / 2 + j] = s * r;
        }
    }
#pragma unroll
    for (int i = 0; i < CBS; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= M[i * (i + 1) / 2 + k] * b[k];
        b[i] = s * M[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = CBS - 1; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < CBS; ++k) s -= M[k * (k + 1) / 2 + i] * b[k];
        b[i] = s * M[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int m = 0; m < CBS; ++m) out[t * out_stride + m] = b[m];
    if (info != nullptr) info[t] = bad;
}

// Static condensation (not on the reference's path; the face-dof form of the multi-GPU exchange).
// A = lc (MS x MS), T = first CBS dofs:
//   rec = A_TT^-1 [ f_T | -A_TF ],  S = A_FF + A_FT rec[:,1:],  g = -A_FT rec[:,0]
// G lanes per cell (16 for k <= 2, 32 for k = 3), 64 / G cells per wavefront, persistent grid.  The
// cell's matrix is staged in LDS with coalesced 16-byte loads (reading the needed blocks straight from
// HBM, strided, was measured 40 % slower).  Lane c owns column c of rec (c = 0: the right-hand side,
// c >= 1: face dof c-1) in registers from the substitutions to the Schur column; A_FT is read from
// LDS as broadcast rows.
template <int CBS, int NF, int G>
__global__ __launch_bounds__(64) void static_condensation_kernel(size_t n, const double *lc, const double *rhs,
                                                                 double *Sout, double *gout, double *recout,
                                                                 int32_t *info, int packed)
{
    // A_TT is symmetric: with an even msize its block of the staged matrix IS a row-major, 16-byte-aligned image with
    // stride msize, and the factorization runs in place (no copy, 20 % less LDS per cell: 10 instead of 8 blocks per CU
    // at k = 2); A_TF / A_FT / A_FF lie outside the entries it overwrites
    constexpr int MS = CBS + NF, CPW = 64 / G;
    constexpr bool INPLACE = MS % 2 == 0 && PA_SC_INPLACE;
    constexpr int LDC = INPLACE ? MS : ((CBS + 1) & ~1);
    constexpr int oA = 0, oLT = INPLACE ? 0 : ((MS * MS + 1) & ~1);
    constexpr int PER_CELL = INPLACE ? ((MS * MS + 1) & ~1) : ((oLT + CBS * LDC + 1) & ~1);
    static_assert(NF + 1 <= G && CBS <= G, "one lane per column / per row");
    __shared__ __attribute__((aligned(16))) double smem[CPW * PER_CELL];
    const int lane = threadIdx.x, g = lane / G, l = lane % G;
    double *A = smem + g * PER_CELL + oA, *LT = smem + g * PER_CELL + oLT;
    const size_t stride = (size_t)gridDim.x * CPW;
    for (size_t base = (size_t)blockIdx.x * CPW; base < n; base += stride) {
        const bool valid = base + g < n;
        const size_t cell = valid ? base + g : n - 1;
        const double *src = lc + cell * (size_t)(MS * MS);
        if ((MS * MS) % 2 == 0) {                          // every cell block is 16-byte aligned
            for (int e = l; e < MS * MS / 2; e += G)
                *reinterpret_cast<double2 *>(A + 2 * e) = *reinterpret_cast<const double2 *>(src + 2 * e);
        } else {
            for (int e = l; e < MS * MS; e += G) A[e] = src[e];
        }
        __syncthreads();
        if (!INPLACE) {
            for (int e = l; e < CBS * CBS; e += G) LT[(e / CBS) * LDC + (e % CBS)] = A[(e % CBS) + (e / CBS) * MS];     // A_TT (symmetric)
            __syncthreads();
        }
        const int bad = lds_cholesky<CBS, LDC, G>(LT, l);
        double x[CBS];
        const int c = l <= NF ? l : 0;
#pragma unroll
        for (int i = 0; i < CBS; ++i)
            x[i] = (c == 0) ? (rhs != nullptr ? rhs[cell * CBS + i] : 0.0) : -A[i + (CBS + c - 1) * MS];
        // row by row, with a scheduling fence between rows when the system is large: left alone the compiler issues the
        // LDS reads of all rows up front (336 VGPRs at cbs = 15: one wave per SIMD)
#pragma unroll
        for (int i = 0; i < CBS; ++i) {
            if (CBS >= PA_SC_FENCE_MIN) __builtin_amdgcn_sched_barrier(0);
            const double s = i == 0 ? x[0] : lds_dotsub_n(x[i], LT + i * LDC, x, i);
            x[i] = s * LT[i * LDC + i];
        }
#pragma unroll
        for (int i = CBS - 1; i >= 0; --i) {
            if (CBS >= PA_SC_FENCE_MIN) __builtin_amdgcn_sched_barrier(0);
            double s = x[i];
#pragma unroll
            for (int k = i + 1; k < CBS; ++k) s -= LT[k * LDC + i] * x[k];
            x[i] = s * LT[i * LDC + i];
        }
        if (recout != nullptr && valid && l <= NF) {
#pragma unroll
            for (int i = 0; i < CBS; ++i) recout[cell * (size_t)(CBS * (NF + 1)) + (size_t)c * CBS + i] = x[i];
        }
        double v[NF];                                       // v = A_FT x : rows CBS.. of column k of A are contiguous
#pragma unroll
        for (int i = 0; i < NF; ++i) v[i] = 0.0;
#pragma unroll
        for (int k = 0; k < CBS; ++k) {
            const double *col = A + CBS + k * MS;
            if (CBS % 2 == 0 && MS % 2 == 0) {              // 16-byte aligned runs: ds_read_b128
#pragma unroll
                for (int i = 0; i < NF; i += 2) {
                    const double2 p = lds_pair(col + i);
                    v[i] = __builtin_fma(p.x, x[k], v[i]);
                    v[i + 1] = __builtin_fma(p.y, x[k], v[i + 1]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NF; ++i) v[i] = __builtin_fma(col[i], x[k], v[i]);
            }
        }
        if (valid && l <= NF) {
            if (c == 0) {
                if (gout != nullptr) {
#pragma unroll
                    for (int i = 0; i < NF; ++i) gout[cell * NF + i] = -v[i];
                }
            } else if (Sout != nullptr) {
                const int j = c - 1;
                const double *aff = A + CBS + (CBS + j) * MS;                    // A_FF(:, j)
                if (packed) {
                    double *dst = Sout + cell * (size_t)(NF * (NF + 1) / 2) + j * (j + 1) / 2;      // rows 0..j of column j
#pragma unroll
                    for (int i = 0; i < NF; ++i)
                        if (i <= j) dst[i] = aff[i] + v[i];
                } else {
                    double *dst = Sout + cell * (size_t)(NF * NF) + (size_t)j * NF;
#pragma unroll
                    for (int i = 0; i < NF; ++i) dst[i] = aff[i] + v[i];
                }
            }
        }
        if (info != nullptr && valid && l == 0) info[cell] = bad;
        __syncthreads();
    }
}

}  // namespace pa
