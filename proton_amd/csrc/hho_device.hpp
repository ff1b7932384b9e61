// hho_device.hpp -- CDNA4 (gfx950) device code for the per-cell HHO local operators.
//
// WHAT is computed is defined by the reference (paths relative to the reference root):
//   make_hho_laplacian            src/methods/hho_bits/hho.hpp:32-96
//   make_hho_naive_stabilization  hho.hpp:99-148      (h = cell AREA,     hho.hpp:119)
//   make_hho_fancy_stabilization  hho.hpp:155-237     (h = cell DIAMETER, hho.hpp:201)
//   cell_basis / face_basis       src/core/core_bits/bases.hpp:70-195, 241-291
//   integrate (tensor / fan / face), gauss_legendre, triangle_quadrature
//                                 src/core/core_bits/quadratures.hpp:78-158, 238-432
//   barycenter/diameter/measure/normals  src/core/core_bits/basic_geom.hpp:247-372
//
// HOW is not the reference's.  G lanes of a 64-wide wavefront cooperate on one cell:
//   * the cell basis is a set of scaled monomials, so every cell integral of phi_i phi_j or
//     grad phi_i . grad phi_j is a *moment*  sum_q w_q bx_q^p by_q^r  of the same quadrature
//     rule: P2(2 recdeg) moments are accumulated (one lane each) and the stiffness / mass
//     matrices are gathered from them -- term by term the same sums the reference forms;
//   * the face basis evaluated at the face Gauss points is t_q^k exactly, so every face mass
//     matrix is (|F|/2) M^ with one constant factorization shared by all faces of all cells;
//   * data = gr_rhs^T (L L^T)^-1 gr_rhs = Y^T Y with Y = L^-1 gr_rhs;
//   * B_F = M_F^-1 T_F - E_F (the reference's proj2 + proj3, hho.hpp:222-231) is block
//     structured whenever T_F = [trace_F | 0]: always for the naive stabilization, and for the
//     fancy one when celdeg == recdeg (then pi_T^k p_T^k v == p_T^k v, hho.hpp:184-190 cancels);
//   * local-matrix entries are accumulated in registers and streamed to HBM, coalesced.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pa {

enum { QUAD_TENSOR = 0, QUAD_FAN = 1 };
enum { STAB_NONE = 0, STAB_NAIVE = 1, STAB_FANCY = 2 };

// Quadrature tables, filled by the host (quad_tables.hpp) and passed by device pointer:
//   gauss_*[n][i]: the n-node rule of gauss_legendre() in the reference's emission order;
//   dun[r][row] = (l0, l1, l2, w) of dunavant rules[r] (0-based, rules[r] == rule_{r+1}).
struct QuadTables {
    double gauss_x[6][5];
    double gauss_w[6][5];
    double dun[9][16][4];
    int dun_n[9];
};

__host__ __device__ constexpr int P2(int d) { return (d + 2) * (d + 1) / 2; }
// points of triangle_quadrature(deg) including the rules[deg] off-by-one (quadratures.hpp:257-268)
__host__ __device__ constexpr int dunavant_points(int deg)
{
    const int d = deg == 0 ? 1 : deg;
    return d == 1 ? 3 : d == 2 ? 4 : d == 3 ? 6 : d == 4 ? 7 : d == 5 ? 12 : d == 6 ? 13 : d == 7 ? 16 : 0;
}
__host__ __device__ constexpr int gauss_nodes(int deg) { return ((deg | 1) + 1) / 2; }
__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }

template <int CD_, int FD_, int QUAD_, int STAB_, int G_>
struct Cfg {
    static constexpr int CD = CD_, FD = FD_, RD = FD_ + 1, QUAD = QUAD_, STAB = STAB_, G = G_;
    static constexpr int RBS = P2(RD), CBS = P2(CD), FBS = FD + 1;
    static constexpr int NF = 4 * FBS, MS = CBS + NF, NR = RBS - 1;
    static constexpr int QDEG = 2 * RD;                       // hho.hpp:55,174
    static constexpr int NG = gauss_nodes(QDEG);
    static constexpr int NT = dunavant_points(QDEG);          // per fan triangle
    static constexpr int NQ = QUAD == QUAD_TENSOR ? NG * NG : 4 * NT;
    static constexpr int NFQ = gauss_nodes(2 * FD);           // hho.hpp:74,132,208
    static constexpr int NFP = 4 * NFQ;
    static constexpr int NP = NQ + NFP;
    static constexpr int NPW = 2 * RD + 1;                    // powers 0..2 recdeg
    static constexpr int NMOM = P2(2 * RD);
    static constexpr int CPW = 64 / G;                        // cells per wavefront
    static constexpr int PPL = cdiv(NP, G);                   // evaluation points per lane
    static constexpr int MPL = cdiv(NMOM, G);                 // moments per lane
    static constexpr int SPL = cdiv(RBS * RBS, G);            // stiffness entries per lane
    static constexpr int EPL = cdiv(MS * MS, G);              // local-matrix entries per lane
    static constexpr bool FANCY = STAB == STAB_FANCY, NAIVE = STAB == STAB_NAIVE;
    // T_F = [trace_F | 0]: block-structured stabilization
    static constexpr bool BLOCK_STAB = NAIVE || (FANCY && CD == RD);
    static constexpr bool GENERAL_FANCY = FANCY && CD != RD;
    static constexpr int TC = GENERAL_FANCY ? RBS : CBS;      // trace columns kept

    // ---- LDS map (doubles, per cell) ------------------------------------------------
    // region A: quadrature-point tables; dead after S3b
    static constexpr int oWPX = 0;                            // NQ x NPW   w * bx^e
    static constexpr int oPY = oWPX + NQ * NPW;               // NQ x NPW   by^e
    static constexpr int oPHF = oPY + NQ * NPW;               // NFP x RBS  phi at face points
    static constexpr int oDN = oPHF + NFP * RBS;              // NFP x NR   w * (grad phi . n)
    static constexpr int endA = oDN + NFP * NR;
    // region C (aliases A): stabilization temporaries
    static constexpr int oPT = 0;                             // NF x CBS   blockdiag(M_F)^-1 trace   (block path)
    static constexpr int oLM = 0;                             // CBS x CBS  chol(M1)                   (general fancy)
    static constexpr int oPR1 = oLM + CBS * CBS;              // CBS x MS
    static constexpr int oTB = oPR1 + CBS * MS;               // NF x MS    B
    static constexpr int oMB = oTB + NF * MS;                 // NF x MS    M_F B
    static constexpr int endC = GENERAL_FANCY ? oMB + NF * MS : (BLOCK_STAB ? NF * CBS : 0);
    static constexpr int oB = imax(endA, endC);
    // region B: lives for the whole cell
    static constexpr int oMOM = oB;                           // NMOM moments
    static constexpr int oST = oMOM + NMOM;                   // RBS x RBS  stiffness
    static constexpr int oMA = oST + RBS * RBS;               // RBS x RBS  mass (general fancy)
    static constexpr int oGR = oMA + (GENERAL_FANCY ? RBS * RBS : 0);   // NR x MS gr_rhs
    static constexpr int oY = oGR + NR * MS;                  // NR x MS    Y = L^-1 gr_rhs
    static constexpr int oOP = oY + NR * MS;                  // NR x MS    oper (general fancy only)
    static constexpr int oLG = oOP + (GENERAL_FANCY ? NR * MS : 0);     // NR x NR chol(gr_lhs)
    static constexpr int oFT = oLG + NR * NR;                 // NF x TC    face traces
    static constexpr int LDS_PER_CELL = oFT + NF * TC;
    // kernel-invariant face tables shared by the cells of a block
    static constexpr int oFB = CPW * LDS_PER_CELL;            // NFQ x FBS: t_q^k
    static constexpr int oLF = oFB + NFQ * FBS;               // FBS x FBS: chol of M^ (reciprocal diagonal)
    static constexpr int oMF = oLF + FBS * FBS;               // FBS x FBS: M^ = sum_q w t^(i+j)
    static constexpr int LDS_DOUBLES = oMF + FBS * FBS;
};

struct LocalOpsArgs {
    const QuadTables *tab;     // device copy of the quadrature tables
    const double *points;      // np x 2
    const uint32_t *ptids;     // nc x 4
    size_t first, n;
    double *oper, *data, *stab, *lc;
    int32_t *info;
};

// index of the monomial bx^p by^r in the graded ordering (total degree, then r)  bases.hpp:114-128
__device__ __forceinline__ int mono_index(int p, int r) { const int k = p + r; return k * (k + 1) / 2 + r; }
// exponents (p, r) of monomial m
__device__ __forceinline__ void mono_exps(int m, int &p, int &r)
{
    int k = 0;
    while ((k + 1) * (k + 2) / 2 <= m) ++k;
    r = m - k * (k + 1) / 2;
    p = k - r;
}

__device__ __forceinline__ double rcp_sqrt(double d) { return 1.0 / sqrt(d); }

// -------------------------------------------------------------------------------------
// Cholesky of an N x N SPD matrix in LDS (column-major, leading dim LD, lower part used),
// one lane per row.  L overwrites the lower triangle; the diagonal receives 1/L[j][j].
// Returns 0 or 1+index of the first non-positive pivot (uniform over the G-lane group).
// -------------------------------------------------------------------------------------
template <int N, int LD, int G>
__device__ __forceinline__ int lds_cholesky(double *A, int l)
{
    double row[N];
    const bool act = l < N;
    const int i = act ? l : 0;
#pragma unroll
    for (int k = 0; k < N; ++k) row[k] = A[i + k * LD];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = row[j];                 // A[i][j] - sum_{k<j} L[i][k] L[j][k]; row j's prefix is in LDS
#pragma unroll
        for (int k = 0; k < j; ++k) s -= row[k] * A[j + k * LD];
        const double d = __shfl(s, j, G);
        if (!(d > 0.0) && !bad) bad = j + 1;
        const double r = rcp_sqrt(d);
        row[j] = s * r;
        if (act && l >= j) A[i + j * LD] = (l == j) ? r : row[j];
        __syncthreads();
    }
    return bad;
}

// x <- L^-1 x (forward) and x <- L^-T x (backward); L as left by lds_cholesky.
template <int N, int LD>
__device__ __forceinline__ void lds_forward(const double *L, double (&x)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i + k * LD] * x[k];
        x[i] = s * L[i + i * LD];
    }
}
template <int N, int LD>
__device__ __forceinline__ void lds_backward(const double *L, double (&x)[N])
{
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[k + i * LD] * x[k];
        x[i] = s * L[i + i * LD];
    }
}

// lane-dependent choice among four register values (kept as scalars: an array indexed this way
// is demoted to scratch memory)
__device__ __forceinline__ double sel4(double v0, double v1, double v2, double v3, int i)
{
    double r = v0;
    r = (i == 1) ? v1 : r;
    r = (i == 2) ? v2 : r;
    r = (i == 3) ? v3 : r;
    return r;
}
__device__ __forceinline__ uint32_t sel4u(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, int i)
{
    uint32_t r = v0;
    r = (i == 1) ? v1 : r;
    r = (i == 2) ? v2 : r;
    r = (i == 3) ? v3 : r;
    return r;
}

// -------------------------------------------------------------------------------------
// The kernel.  One wavefront per block; G lanes per cell; persistent over cells.
// -------------------------------------------------------------------------------------
template <class C>
__global__ __launch_bounds__(64) void hho_local_ops_kernel(LocalOpsArgs a)
{
    constexpr int G = C::G, RBS = C::RBS, CBS = C::CBS, FBS = C::FBS, MS = C::MS, NR = C::NR, NF = C::NF;
    constexpr int NQ = C::NQ, NFQ = C::NFQ, NFP = C::NFP, NP = C::NP, RD = C::RD, NPW = C::NPW, TC = C::TC;
    static_assert(MS <= G, "one lane per local-matrix column");
    static_assert(RBS <= G && NF <= G, "one lane per row in the factorizations");
    static_assert(C::NQ > 0, "empty quadrature rule (the rules[8] hole)");

    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int g = lane / G, l0 = lane % G;
    const int l = l0;
    double *S = smem + g * C::LDS_PER_CELL;
    double *FB = smem + C::oFB, *LF = smem + C::oLF, *MF = smem + C::oMF;
    const QuadTables *__restrict__ tab = a.tab;

    // ---- kernel-invariant tables.  ep = 4 (base . (x - bar_F)) / h_F^2 (bases.hpp:269-272) equals
    // the Gauss abscissa t on the segment, so phi_F(x_q) = t_q^k and M_F = (|F|/2) sum_q w_q t_q^(i+j).
    if (lane < NFQ * FBS) {
        const int q = lane / FBS, k = lane % FBS;
        const double t = tab->gauss_x[NFQ][q];
        double v = 1.0;
        for (int e = 0; e < k; ++e) v *= t;
        FB[q * FBS + k] = v;
    }
    __syncthreads();
    if (lane < FBS * FBS) {
        const int i = lane % FBS, j = lane / FBS;
        double s = 0.0;
        for (int q = 0; q < NFQ; ++q) s += (tab->gauss_w[NFQ][q] * FB[q * FBS + i]) * FB[q * FBS + j];
        MF[i + j * FBS] = s;
        LF[i + j * FBS] = s;
    }
    __syncthreads();
    lds_cholesky<FBS, FBS, 64>(LF, lane);

    // ---- per-lane, cell-invariant bookkeeping -------------------------------------------
    // reference coordinates of the evaluation points this lane owns
    double r0[C::PPL], r1[C::PPL], r2[C::PPL], rw[C::PPL];
#pragma unroll
    for (int r = 0; r < C::PPL; ++r) {
        const int p = l + r * G;
        r0[r] = r1[r] = r2[r] = rw[r] = 0.0;
        if (p < NQ) {
            if (C::QUAD == QUAD_TENSOR) {
                const int i = p % C::NG, j = p / C::NG;            // outer eta, inner xi  quadratures.hpp:355-357
                r0[r] = tab->gauss_x[C::NG][i];
                r1[r] = tab->gauss_x[C::NG][j];
                rw[r] = tab->gauss_w[C::NG][i] * tab->gauss_w[C::NG][j];
            } else {
                const int row = p % C::NT;
                constexpr int R = C::QDEG == 0 ? 1 : C::QDEG;      // rules[deg]  quadratures.hpp:257
                r0[r] = tab->dun[R][row][0];
                r1[r] = tab->dun[R][row][1];
                r2[r] = tab->dun[R][row][2];
                rw[r] = tab->dun[R][row][3];
            }
        } else if (p < NP) {
            const int q = (p - NQ) % NFQ;
            r0[r] = tab->gauss_x[NFQ][q];
            rw[r] = tab->gauss_w[NFQ][q];
        }
    }
    // moments owned by this lane: exponents (p, r)
    int mom_pr[C::MPL];
#pragma unroll
    for (int t = 0; t < C::MPL; ++t) {
        const int mu = l + t * G;
        int p = 0, r = 0;
        if (mu < C::NMOM) mono_exps(mu, p, r);
        mom_pr[t] = p | (r << 8);
    }
    // stiffness entries owned by this lane: stiff(i,j) = ih^2 (a a' MOM(a+a'-2, b+b') + b b' MOM(a+a', b+b'-2))
    // packed as idx1 | idx2<<8 | c1<<16 | c2<<24   (bases.hpp:170-176 with hho.hpp:57-61)
    uint32_t st_code[C::SPL];
#pragma unroll
    for (int t = 0; t < C::SPL; ++t) {
        const int e = l + t * G;
        uint32_t code = 0;
        if (e < RBS * RBS) {
            int ai, bi, aj, bj;
            mono_exps(e % RBS, ai, bi);
            mono_exps(e / RBS, aj, bj);
            const int c1 = ai * aj, c2 = bi * bj;
            const int i1 = c1 ? mono_index(ai + aj - 2, bi + bj) : 0;
            const int i2 = c2 ? mono_index(ai + aj, bi + bj - 2) : 0;
            code = (uint32_t)i1 | ((uint32_t)i2 << 8) | ((uint32_t)c1 << 16) | ((uint32_t)c2 << 24);
        }
        st_code[t] = code;
    }

    const size_t stride = (size_t)gridDim.x * C::CPW;
    for (size_t base = (size_t)blockIdx.x * C::CPW; base < a.n; base += stride) {
        // Re-derive the lane index opaquely per cell: otherwise LICM hoists every per-entry index
        // computation of every stage out of the cell loop and the kernel spills.
        int l = l0;
        asm volatile("" : "+v"(l));
        const bool valid = base + g < a.n;
        const size_t cell = a.first + (valid ? base + g : a.n - 1);

        // ================= S0: geometry (every lane of the group, registers) ==========
        const uint4 idv = *reinterpret_cast<const uint4 *>(a.ptids + 4 * cell);
        const double2 q0 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.x);
        const double2 q1 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.y);
        const double2 q2 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.z);
        const double2 q3 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.w);
        const double px0 = q0.x, py0 = q0.y, px1 = q1.x, py1 = q1.y, px2 = q2.x, py2 = q2.y, px3 = q3.x, py3 = q3.y;
        double barx, bary;                          // barycenter  basic_geom.hpp:247-270
        {
            const double ax = px1 - px0, ay = py1 - py0, bx = px2 - px0, by = py2 - py0;
            const double cx = px3 - px0, cy = py3 - py0;
            const double d1 = (ax * by - ay * bx) / 2.0, d2 = (bx * cy - by * cx) / 2.0;
            const double rx = (ax + bx) * d1 + (bx + cx) * d2, ry = (ay + by) * d1 + (by + cy) * d2;
            const double den = d1 + d2;
            barx = px0 + rx / (den * 3); bary = py0 + ry / (den * 3);
        }
        // edge vectors in cell (CCW) order and their lengths: faces, normals, diameter
        const double e0x = px1 - px0, e0y = py1 - py0, e1x = px2 - px1, e1y = py2 - py1;
        const double e2x = px3 - px2, e2y = py3 - py2, e3x = px0 - px3, e3y = py0 - py3;
        const double len0 = sqrt(e0x * e0x + e0y * e0y), len1 = sqrt(e1x * e1x + e1y * e1y);
        const double len2 = sqrt(e2x * e2x + e2y * e2y), len3 = sqrt(e3x * e3x + e3y * e3y);
        double hT;                                  // diameter  basic_geom.hpp:288-305
        {
            const double d02x = px2 - px0, d02y = py2 - py0, d13x = px3 - px1, d13y = py3 - py1;
            hT = fmax(fmax(len0, len1), fmax(len2, len3));
            hT = fmax(hT, fmax(sqrt(d02x * d02x + d02y * d02y), sqrt(d13x * d13x + d13y * d13y)));
        }
        const double ihalf = 1.0 / (0.5 * hT);      // bx = (x - bar)/(h/2)  bases.hpp:98-99
        const double ih = 2.0 / hT;                 // bases.hpp:142
        double area;                                // measure  basic_geom.hpp:317-334
        {
            const double ux = px1 - px0, uy = py1 - py0, vx = px2 - px0, vy = py2 - py0;
            const double wx = px3 - px0, wy = py3 - py0;
            area = fabs(ux * vy - uy * vx) * 0.5 + fabs(vx * wy - vy * wx) * 0.5;
        }
        const double hs0 = 0.5 * len0, hs1 = 0.5 * len1, hs2 = 0.5 * len2, hs3 = 0.5 * len3;   // |F|/2: M_F = hs * M^

        // ================= S1: evaluation points ======================================
#pragma unroll
        for (int r = 0; r < C::PPL; ++r) {
            const int p = l + r * G;
            if (p < NP) {
                double x, y, w, nx = 0.0, ny = 0.0;
                const bool is_cell = p < NQ;
                if (is_cell) {
                    if (C::QUAD == QUAD_TENSOR) {
                        const double xi = r0[r], eta = r1[r];      // quadratures.hpp:331-352
                        x = 0.25 * px0 * (1 - xi) * (1 - eta) + 0.25 * px1 * (1 + xi) * (1 - eta) +
                            0.25 * px2 * (1 + xi) * (1 + eta) + 0.25 * px3 * (1 - xi) * (1 + eta);
                        y = 0.25 * py0 * (1 - xi) * (1 - eta) + 0.25 * py1 * (1 + xi) * (1 - eta) +
                            0.25 * py2 * (1 + xi) * (1 + eta) + 0.25 * py3 * (1 - xi) * (1 + eta);
                        const double j11 = 0.25 * ((px1 - px0) * (1 - eta) + (px2 - px3) * (1 + eta));
                        const double j12 = 0.25 * ((py1 - py0) * (1 - eta) + (py2 - py3) * (1 + eta));
                        const double j21 = 0.25 * ((px3 - px0) * (1 - xi) + (px2 - px1) * (1 + xi));
                        const double j22 = 0.25 * ((py3 - py0) * (1 - xi) + (py2 - py1) * (1 + xi));
                        w = rw[r] * fabs(j11 * j22 - j12 * j21);
                    } else {
                        const int t = p / C::NT;                   // fan triangle (p_t, p_{t+1}, bar)  quadratures.hpp:390-396
                        const double ax = sel4(px0, px1, px2, px3, t), ay = sel4(py0, py1, py2, py3, t);
                        const double bx = sel4(px1, px2, px3, px0, t), by = sel4(py1, py2, py3, py0, t);
                        const double v0x = bx - ax, v0y = by - ay, v1x = barx - ax, v1y = bary - ay;
                        const double tarea = fabs((v0x * v1y - v0y * v1x) / 2.0);      // quadratures.hpp:248-251
                        x = ax * r0[r] + bx * r1[r] + barx * r2[r];
                        y = ay * r0[r] + by * r1[r] + bary * r2[r];
                        w = tarea * rw[r];
                    }
                } else {
                    const int f = (p - NQ) / NFQ;
                    const double ax = sel4(px0, px1, px2, px3, f), ay = sel4(py0, py1, py2, py3, f);
                    const double bx = sel4(px1, px2, px3, px0, f), by = sel4(py1, py2, py3, py0, f);
                    const uint32_t ia = sel4u(idv.x, idv.y, idv.z, idv.w, f), ib = sel4u(idv.y, idv.z, idv.w, idv.x, f);
                    const double ex = sel4(e0x, e1x, e2x, e3x, f), ey = sel4(e0y, e1y, e2y, e3y, f);   // edge in cell order
                    const double len = sel4(len0, len1, len2, len3, f);
                    nx = ey / len; ny = -ex / len;                // outward normal  basic_geom.hpp:361-369
                    // the face runs from its LOWER-id endpoint (basic_geom.hpp:202-203, bases.hpp:260-261):
                    // its q-th point sits at -t_q in cell order when the ids are descending
                    const double t = (ia > ib) ? -r0[r] : r0[r];
                    x = 0.5 * (1 - t) * ax + 0.5 * (1 + t) * bx;   // quadratures.hpp:420-428
                    y = 0.5 * (1 - t) * ay + 0.5 * (1 + t) * by;
                    w = rw[r] * len * 0.5;
                }
                const double bx_ = (x - barx) * ihalf, by_ = (y - bary) * ihalf;
                if (is_cell) {
                    // w * bx^e and by^e, e = 0..2 recdeg: the factors of every cell moment
                    double vx = w, vy = 1.0;
#pragma unroll
                    for (int e = 0; e < NPW; ++e) {
                        S[C::oWPX + p * NPW + e] = vx;
                        S[C::oPY + p * NPW + e] = vy;
                        vx *= bx_; vy *= by_;
                    }
                } else {
                    // scaled monomials and normal derivatives at a face point  bases.hpp:93-184, hho.hpp:77-83
                    const int pf = p - NQ;
                    double pwx[RD + 1], pwy[RD + 1];
                    pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
                    for (int e = 1; e <= RD; ++e) { pwx[e] = pwx[e - 1] * bx_; pwy[e] = pwy[e - 1] * by_; }
                    int m = 0;
#pragma unroll
                    for (int kk = 0; kk <= RD; ++kk) {
#pragma unroll
                        for (int ii = 0; ii <= kk; ++ii, ++m) {
                            const int ex_ = kk - ii, ey_ = ii;      // (px,py) = (k-i, i)  bases.hpp:119-120
                            S[C::oPHF + pf * RBS + m] = pwx[ex_] * pwy[ey_];
                            if (m > 0) {
                                const double gx = ex_ == 0 ? 0.0 : (ex_ * ih) * pwx[ex_ > 0 ? ex_ - 1 : 0] * pwy[ey_];
                                const double gy = ey_ == 0 ? 0.0 : (ey_ * ih) * pwx[ex_] * pwy[ey_ > 0 ? ey_ - 1 : 0];
                                S[C::oDN + pf * NR + (m - 1)] = w * (gx * nx + gy * ny);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();

        // ================= S2: cell moments, face traces ==============================
#pragma unroll
        for (int t = 0; t < C::MPL; ++t) {
            const int mu = l + t * G;
            if (mu < C::NMOM) {
                const int p = mom_pr[t] & 0xff, r = mom_pr[t] >> 8;
                double s = 0.0;
#pragma unroll 4
                for (int q = 0; q < NQ; ++q) s += S[C::oWPX + q * NPW + p] * S[C::oPY + q * NPW + r];
                S[C::oMOM + mu] = s;
            }
        }
        // FT[fk][m] = sum_q (w_q |F|/2 t_q^k) phi_m(x_fq)   hho.hpp:209-216 / 133-140.
        // Point q of face f IS the reference's q-th face point (S1 mirrors t for faces whose lower-id
        // endpoint comes second), so its face-basis value is t_q^k for either orientation.
        if (C::STAB != STAB_NONE) {
            constexpr int NE = NF * TC;
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += G) {
                const int e = e0 + l;
                if (e < NE) {
                    const int m = e / NF, fk = e % NF, f = fk / FBS, k = fk % FBS;
                    const double hs = sel4(hs0, hs1, hs2, hs3, f);
                    double s = 0.0;
#pragma unroll
                    for (int q = 0; q < NFQ; ++q)
                        s += (tab->gauss_w[NFQ][q] * hs * FB[q * FBS + k]) * S[C::oPHF + (f * NFQ + q) * RBS + m];
                    S[C::oFT + fk + m * NF] = s;
                }
            }
        }
        __syncthreads();

        // ================= S3: stiffness (+mass) from moments ========================
        {
            const double ih2 = ih * ih;
#pragma unroll
            for (int t = 0; t < C::SPL; ++t) {
                const int e = l + t * G;
                if (e < RBS * RBS) {
                    const uint32_t code = st_code[t];
                    const double c1 = (double)((code >> 16) & 0xff), c2 = (double)(code >> 24);
                    const double v = c1 * S[C::oMOM + (code & 0xff)] + c2 * S[C::oMOM + ((code >> 8) & 0xff)];
                    S[C::oST + e] = ih2 * v;
                }
            }
            if (C::GENERAL_FANCY) {
#pragma unroll
                for (int e0 = 0; e0 < RBS * RBS; e0 += G) {
                    const int e = e0 + l;
                    if (e < RBS * RBS) {
                        int ai, bi, aj, bj;
                        mono_exps(e % RBS, ai, bi);
                        mono_exps(e / RBS, aj, bj);
                        S[C::oMA + e] = S[C::oMOM + mono_index(ai + aj, bi + bj)];
                    }
                }
            }
        }
        __syncthreads();

        // ================= S3b: gr_rhs, gr_lhs  hho.hpp:63-85 =========================
        {
            constexpr int NE = NR * MS;
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += G) {
                const int e = e0 + l;
                if (e < NE) {
                    const int i = e % NR, j = e / NR;
                    double s;
                    if (j < CBS) {
                        s = S[C::oST + (i + 1) + j * RBS];
#pragma unroll 4
                        for (int pf = 0; pf < NFP; ++pf) s -= S[C::oDN + pf * NR + i] * S[C::oPHF + pf * RBS + j];
                    } else {
                        const int f = (j - CBS) / FBS, k = (j - CBS) % FBS;
                        s = 0.0;
#pragma unroll
                        for (int q = 0; q < NFQ; ++q) s += S[C::oDN + (f * NFQ + q) * NR + i] * FB[q * FBS + k];
                    }
                    S[C::oGR + i + j * NR] = s;
                }
            }
#pragma unroll
            for (int e0 = 0; e0 < NR * NR; e0 += G) {
                const int e = e0 + l;
                if (e < NR * NR) S[C::oLG + e] = S[C::oST + (e % NR + 1) + (e / NR + 1) * RBS];
            }
        }
        __syncthreads();

        // ================= S4/S5: L L^T = gr_lhs ; Y = L^-1 gr_rhs ; oper = L^-T Y  hho.hpp:92
        int bad = lds_cholesky<NR, NR, G>(S + C::oLG, l);
        {
            double xcol[NR];
            const int c = l < MS ? l : 0;
#pragma unroll
            for (int k = 0; k < NR; ++k) xcol[k] = S[C::oGR + k + c * NR];
            lds_forward<NR, NR>(S + C::oLG, xcol);
            if (l < MS) {
#pragma unroll
                for (int k = 0; k < NR; ++k) S[C::oY + k + c * NR] = xcol[k];
            }
            if (C::GENERAL_FANCY || a.oper != nullptr) {
                lds_backward<NR, NR>(S + C::oLG, xcol);
                if (C::GENERAL_FANCY && l < MS) {
#pragma unroll
                    for (int k = 0; k < NR; ++k) S[C::oOP + k + c * NR] = xcol[k];
                }
                if (a.oper != nullptr && valid && l < MS) {
                    double *dst = a.oper + (cell - a.first) * (size_t)(NR * MS) + (size_t)c * NR;
#pragma unroll
                    for (int k = 0; k < NR; ++k) dst[k] = xcol[k];
                }
            }
        }
        __syncthreads();

        // ================= S6: data = gr_rhs^T oper = Y^T Y  hho.hpp:93 ===============
        double acc_d[C::EPL], acc_s[C::EPL];
#pragma unroll
        for (int t = 0; t < C::EPL; ++t) {
            const int e = l + t * G;
            const int i = e < MS * MS ? e % MS : 0, j = e < MS * MS ? e / MS : 0;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NR; ++k) s += S[C::oY + k + i * NR] * S[C::oY + k + j * NR];
            asm volatile("" : "+v"(s));          // pin: keeps the FMAs next to their LDS reads (no sinking to S8)
            acc_d[t] = s;
            acc_s[t] = 0.0;
        }

        // ================= S7: stabilization =========================================
        if (C::BLOCK_STAB) {
            // B_F = [ M_F^-1 trace_F | -E_F ]  =>  (1/h) sum_F B_F^T M_F B_F =
            //   (1/h) [ sum_F tr_F^T M_F^-1 tr_F   -tr^T ;  -tr   blockdiag(M_F) ]
            {
                const int c = l < CBS ? l : 0;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    double xf[FBS];
#pragma unroll
                    for (int k = 0; k < FBS; ++k) xf[k] = S[C::oFT + (f * FBS + k) + c * NF];
                    lds_forward<FBS, FBS>(LF, xf);
                    lds_backward<FBS, FBS>(LF, xf);
                    const double ihs = 1.0 / (f == 0 ? hs0 : f == 1 ? hs1 : f == 2 ? hs2 : hs3);
                    if (l < CBS) {
#pragma unroll
                        for (int k = 0; k < FBS; ++k) S[C::oPT + (f * FBS + k) + c * NF] = xf[k] * ihs;
                    }
                }
            }
            __syncthreads();
            const double hinv = 1.0 / (C::FANCY ? hT : area);      // hho.hpp:201 / hho.hpp:119
#pragma unroll
            for (int t = 0; t < C::EPL; ++t) {
                const int e = l + t * G;
                const int i = e < MS * MS ? e % MS : 0, j = e < MS * MS ? e / MS : 0;
                double s = 0.0;
                if (i < CBS && j < CBS) {
#pragma unroll
                    for (int r = 0; r < NF; ++r) s += S[C::oFT + r + i * NF] * S[C::oPT + r + j * NF];
                } else if (i < CBS) {
                    s = -S[C::oFT + (j - CBS) + i * NF];
                } else if (j < CBS) {
                    s = -S[C::oFT + (i - CBS) + j * NF];
                } else {
                    const int fi = (i - CBS) / FBS, fj = (j - CBS) / FBS;
                    if (fi == fj) s = sel4(hs0, hs1, hs2, hs3, fi) * MF[(i - CBS) % FBS + ((j - CBS) % FBS) * FBS];
                }
                s *= hinv;
                asm volatile("" : "+v"(s));
                acc_s[t] = s;
            }
        } else if (C::GENERAL_FANCY) {
            // proj1 = [I 0] - M1^{-1} (M2 R)   hho.hpp:184-190
#pragma unroll
            for (int e0 = 0; e0 < CBS * CBS; e0 += G) {
                const int e = e0 + l;
                if (e < CBS * CBS) S[C::oLM + e] = S[C::oMA + (e % CBS) + (e / CBS) * RBS];
            }
            __syncthreads();
            const int badm = lds_cholesky<CBS, CBS, G>(S + C::oLM, l);
            if (badm && !bad) bad = 100 + badm;
            const int c = l < MS ? l : 0;
            {
                double xcol[CBS];
#pragma unroll
                for (int i = 0; i < CBS; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < NR; ++k) s += S[C::oMA + i + (1 + k) * RBS] * S[C::oOP + k + c * NR];
                    xcol[i] = s;
                }
                lds_forward<CBS, CBS>(S + C::oLM, xcol);
                lds_backward<CBS, CBS>(S + C::oLM, xcol);
                if (l < MS) {
#pragma unroll
                    for (int i = 0; i < CBS; ++i) S[C::oPR1 + i + c * CBS] = (i == c ? 1.0 : 0.0) - xcol[i];
                }
            }
            __syncthreads();
            // column c of T_F = MR1 R + MR2 proj1 (hho.hpp:222-230; piKF.solve is linear), B = M_F^-1 T - E
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                double xf[FBS];
#pragma unroll
                for (int k = 0; k < FBS; ++k) {
                    const int r = f * FBS + k;
                    double s = 0.0;
#pragma unroll
                    for (int kk = 0; kk < NR; ++kk) s += S[C::oFT + r + (1 + kk) * NF] * S[C::oOP + kk + c * NR];
#pragma unroll
                    for (int kk = 0; kk < CBS; ++kk) s += S[C::oFT + r + kk * NF] * S[C::oPR1 + kk + c * CBS];
                    xf[k] = s;
                }
                lds_forward<FBS, FBS>(LF, xf);
                lds_backward<FBS, FBS>(LF, xf);
                const double hs = f == 0 ? hs0 : f == 1 ? hs1 : f == 2 ? hs2 : hs3, ihs = 1.0 / hs;
#pragma unroll
                for (int k = 0; k < FBS; ++k) {
                    double b = xf[k] * ihs;
                    if (c == CBS + f * FBS + k) b -= 1.0;             // - I_F  hho.hpp:226
                    xf[k] = b;
                }
                if (l < MS) {
#pragma unroll
                    for (int k = 0; k < FBS; ++k) {
                        double mb = 0.0;
#pragma unroll
                        for (int k2 = 0; k2 < FBS; ++k2) mb += MF[k + k2 * FBS] * xf[k2];
                        S[C::oTB + (f * FBS + k) + c * NF] = xf[k];
                        S[C::oMB + (f * FBS + k) + c * NF] = hs * mb;
                    }
                }
            }
            __syncthreads();
            const double hinv = 1.0 / hT;                              // hho.hpp:201,233
#pragma unroll
            for (int t = 0; t < C::EPL; ++t) {
                const int e = l + t * G;
                const int i = e < MS * MS ? e % MS : 0, j = e < MS * MS ? e / MS : 0;
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < NF; ++r) s += S[C::oTB + r + i * NF] * S[C::oMB + r + j * NF];
                s *= hinv;
                asm volatile("" : "+v"(s));
                acc_s[t] = s;
            }
        }

        // ================= S8: stream the local matrices to HBM =======================
        if (valid) {
            const size_t off = (cell - a.first) * (size_t)(MS * MS);
#pragma unroll
            for (int t = 0; t < C::EPL; ++t) {
                const int e = l + t * G;
                if (e < MS * MS) {
                    if (a.lc != nullptr) a.lc[off + e] = acc_d[t] + acc_s[t];
                    if (a.data != nullptr) a.data[off + e] = acc_d[t];
                    if (a.stab != nullptr) a.stab[off + e] = acc_s[t];
                }
            }
            if (a.info != nullptr && l == 0) a.info[cell - a.first] = bad;
        }
        __syncthreads();      // region A is rewritten by the next cell
    }
}

}  // namespace pa
