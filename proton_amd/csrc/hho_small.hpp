// hho_small.hpp -- the local operators of the SMALL degree pairs (msize <= 9: (0,0), (1,0) and the obstacle pair (0,1) of
// apps/obstacle/obstacle.cpp:51), ONE THREAD PER CELL, one kernel, no record.
//
// For msize = 9 the cooperative kernel of hho_device.hpp has nine of sixteen lanes busy in its column-parallel stages and its
// pre-pass record (mass rows + chol(M1) for the dense fancy form) is as large as the output: 1.97 x the algorithmic bytes in
// HBM, 0.18 of the roofline (round 2).  The whole operator of such a cell is a few thousand FMAs and 648 B of output: here a
// lane keeps everything of its cell in registers -- the thread-per-cell form of hho_pre.hpp carried through to lc --
//   geometry, cell quadrature, moments, stiffness, Cholesky of gr_lhs                      hho.hpp:55-63, 92
//   gr_rhs (cell columns from the stiffness, face terms at the face Gauss points)          hho.hpp:64-85
//   Y = L^-1 gr_rhs, data = Y^T Y                                                          hho.hpp:92-93
//   stabilization face by face, U_F = sqrt(|F|/2h) (L^^-1 T_F - L^^T E_F), stab += U_F^T U_F  hho.hpp:99-148, 155-237
//     (T_F = [trace_F | 0] for the naive form and for the fancy one when celdeg == recdeg; otherwise
//      T_F = MR1 R + MR2 proj1 with R = L^-T Y never formed: v^T R = (L^-1 v)^T Y for the rows v of MR1 and of M2)
// -- and the wavefront's 64 local matrices, contiguous in the cell-major output, leave through an LDS transpose in pieces of
// a third of a matrix, so that consecutive lanes store consecutive doubles.  HBM: 80 B read, 8 msize^2 written per cell.
// lc (and info) only: callers that want oper / data / stab, or the condensed mode, take the cooperative kernel.
#pragma once
#include "hho_device.hpp"

namespace pa {

struct SmallOpsArgs {
    const QuadTables *tab;
    const double *points;
    const uint32_t *ptids;
    size_t first, n;
    double *lc;
    int32_t *info;
};

// (register budget: the dense fancy form holds Y, the packed lc, L, proj1 and a face's T_F at once -- 320 registers without a
// spill, one wavefront per SIMD; bounded to two it spills 126: measured slower)
#ifndef PA_SMALL_WAVES_GF
#define PA_SMALL_WAVES_GF 1
#endif
template <class C>
__global__ __launch_bounds__(64, C::GENERAL_FANCY ? PA_SMALL_WAVES_GF : 2) void hho_small_ops_kernel(SmallOpsArgs a)
{
    constexpr int RD = C::RD, RBS = C::RBS, CBS = C::CBS, FBS = C::FBS, MS = C::MS, NR = C::NR, NFQ = C::NFQ;
    constexpr int NPW = C::NPW, NMOM = C::NMOM;
    constexpr bool GF = C::GENERAL_FANCY;
    constexpr int NSYM = MS * (MS + 1) / 2;
    // the output leaves in NCH pieces of CH doubles per cell (CH odd: the lanes' LDS columns fall on different banks)
    constexpr int NCH = 3, CH = (MS * MS + NCH - 1) / NCH | 1;
    static_assert(NCH * CH >= MS * MS, "the pieces cover the matrix");
    __shared__ double stage[64 * CH];
    const int lane = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * 64;
    const size_t i = i0 + lane;
    const bool valid = i < a.n;
    const size_t cell = a.first + (valid ? i : a.n - 1);
    const QuadTables *__restrict__ tab = a.tab;

    // ---- geometry (the expressions of hho_pre.hpp)
    const uint4 idv = *reinterpret_cast<const uint4 *>(a.ptids + 4 * cell);
    const double2 q0 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.x);
    const double2 q1 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.y);
    const double2 q2 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.z);
    const double2 q3 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.w);
    const double px[5] = {q0.x, q1.x, q2.x, q3.x, q0.x}, py[5] = {q0.y, q1.y, q2.y, q3.y, q0.y};
    double barx, bary;                          // barycenter  basic_geom.hpp:247-270
    {
        const double ax = px[1] - px[0], ay = py[1] - py[0], bx = px[2] - px[0], by = py[2] - py[0];
        const double cx = px[3] - px[0], cy = py[3] - py[0];
        const double d1 = (ax * by - ay * bx) * 0.5, d2 = (bx * cy - by * cx) * 0.5;
        const double rx = (ax + bx) * d1 + (bx + cx) * d2, ry = (ay + by) * d1 + (by + cy) * d2;
        const double iden = fast_rcp((d1 + d2) * 3);
        barx = px[0] + rx * iden; bary = py[0] + ry * iden;
    }
    double ex[4], ey[4], s2[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { ex[f] = px[f + 1] - px[f]; ey[f] = py[f + 1] - py[f]; s2[f] = ex[f] * ex[f] + ey[f] * ey[f]; }
    double h2;                                  // diameter^2  basic_geom.hpp:288-305
    {
        const double d02x = px[2] - px[0], d02y = py[2] - py[0], d13x = px[3] - px[1], d13y = py[3] - py[1];
        h2 = fmax(fmax(s2[0], s2[1]), fmax(s2[2], s2[3]));
        h2 = fmax(h2, fmax(d02x * d02x + d02y * d02y, d13x * d13x + d13y * d13y));
    }
    const double rh = fast_rsqrt(h2);           // 1 / h_T
    const double ih = 2.0 * rh;                 // bases.hpp:98-99,142
    double hinv = rh;                           // fancy: h = cell diameter  hho.hpp:201
    if (C::NAIVE) {                             // naive: h = cell area      hho.hpp:119
        const double ux = px[1] - px[0], uy = py[1] - py[0], vx = px[2] - px[0], vy = py[2] - py[0];
        const double wx = px[3] - px[0], wy = py[3] - py[0];
        hinv = fast_rcp(fabs(ux * vy - uy * vx) * 0.5 + fabs(vx * wy - vy * wx) * 0.5);
    }

    // ---- moments  sum_q w_q bx_q^p by_q^r,  p + r <= 2 recdeg
    double mom[NMOM];
#pragma unroll
    for (int m = 0; m < NMOM; ++m) mom[m] = 0.0;
    auto add_point = [&](double x, double y, double w) {
        const double bx_ = (x - barx) * ih, by_ = (y - bary) * ih;
        double wbx[NPW], pby[NPW];
        wbx[0] = w; pby[0] = 1.0;
#pragma unroll
        for (int e = 1; e < NPW; ++e) { wbx[e] = wbx[e - 1] * bx_; pby[e] = pby[e - 1] * by_; }
#pragma unroll
        for (int k = 0; k < NPW; ++k)
#pragma unroll
            for (int r = 0; r <= k; ++r) mom[k * (k + 1) / 2 + r] += wbx[k - r] * pby[r];
    };
    if (C::QUAD == QUAD_TENSOR) {
#pragma unroll
        for (int j = 0; j < C::NG; ++j) {             // outer eta, inner xi  quadratures.hpp:355-357
            const double eta = tab->gauss_x[C::NG][j];
            const double am = 0.25 * (1 - eta), ap = 0.25 * (1 + eta);
#pragma unroll
            for (int ii = 0; ii < C::NG; ++ii) {
                const double xi = tab->gauss_x[C::NG][ii];
                const double rw = tab->gauss_w[C::NG][ii] * tab->gauss_w[C::NG][j];
                const double bm = 0.25 * (1 - xi), bp = 0.25 * (1 + xi);
                const double n0 = (1 - xi) * am, n1 = (1 + xi) * am, n2 = (1 + xi) * ap, n3 = (1 - xi) * ap;
                const double x = n0 * px[0] + n1 * px[1] + n2 * px[2] + n3 * px[3];
                const double y = n0 * py[0] + n1 * py[1] + n2 * py[2] + n3 * py[3];
                const double j11 = ex[0] * am - ex[2] * ap, j12 = ey[0] * am - ey[2] * ap;
                const double j21 = ex[1] * bp - ex[3] * bm, j22 = ey[1] * bp - ey[3] * bm;
                add_point(x, y, rw * fabs(j11 * j22 - j12 * j21));      // quadratures.hpp:331-352
            }
        }
    } else {
        constexpr int R = C::QDEG == 0 ? 1 : C::QDEG;                    // rules[deg]  quadratures.hpp:257
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                    // fan triangle (p_t, p_t+1, bar)  quadratures.hpp:390-396
            const double ax = px[t], ay = py[t], bx = px[t + 1], by = py[t + 1];
            const double v0x = bx - ax, v0y = by - ay, v1x = barx - ax, v1y = bary - ay;
            const double tarea = fabs((v0x * v1y - v0y * v1x) * 0.5);    // quadratures.hpp:248-251
#pragma unroll
            for (int row = 0; row < C::NT; ++row) {
                const double l0 = tab->dun[R][row][0], l1 = tab->dun[R][row][1], l2 = tab->dun[R][row][2];
                add_point(ax * l0 + bx * l1 + barx * l2, ay * l0 + by * l1 + bary * l2, tarea * tab->dun[R][row][3]);
            }
        }
    }

    // ---- stiffness rows 1.. from the moments: stiff(i,j) = ih^2 (a a' MOM(a+a'-2, b+b') + b b' MOM(a+a', b+b'-2))
    //      bases.hpp:170-176, hho.hpp:57-61.  gr_lhs = stiff[1:,1:] -> packed L; gr_rhs[:, c < cbs] = stiff[1:, c]  (hho.hpp:63-64)
    const double ih2 = ih * ih;
    auto stiff = [&](int mi, int mj) -> double {          // monomial indices (compile-time after unrolling)
        int ki = 0, kj = 0;
        while ((ki + 1) * (ki + 2) / 2 <= mi) ++ki;
        while ((kj + 1) * (kj + 2) / 2 <= mj) ++kj;
        const int bi = mi - ki * (ki + 1) / 2, ai = ki - bi, bj = mj - kj * (kj + 1) / 2, aj = kj - bj;
        const int c1 = ai * aj, c2 = bi * bj;
        const int k1 = ai + aj - 2 + bi + bj, k2 = ai + aj + bi + bj - 2;
        const double m1 = c1 ? mom[k1 * (k1 + 1) / 2 + (bi + bj)] : 0.0;
        const double m2 = c2 ? mom[k2 * (k2 + 1) / 2 + (bi + bj - 2 < 0 ? 0 : bi + bj - 2)] : 0.0;
        return ih2 * ((double)c1 * m1 + (double)c2 * m2);
    };
    double L[NR * (NR + 1) / 2], rd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) L[r * (r + 1) / 2 + c] = stiff(r + 1, c + 1);
    double G[NR][MS];                                     // gr_rhs, then Y = L^-1 gr_rhs in place
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < MS; ++c) G[r][c] = c < CBS ? stiff(r + 1, c) : 0.0;
    // rows i < CBS of the cell mass matrix (M2 of hho.hpp:185 is its columns 1..), dense fancy form only
    double massr[GF ? CBS : 1][GF ? RBS : 1];
    if (GF) {
#pragma unroll
        for (int i2 = 0; i2 < CBS; ++i2)
#pragma unroll
            for (int j = 0; j < RBS; ++j) {
                int ki = 0, kj = 0;
                while ((ki + 1) * (ki + 2) / 2 <= i2) ++ki;
                while ((kj + 1) * (kj + 2) / 2 <= j) ++kj;
                const int ri = i2 - ki * (ki + 1) / 2, rj = j - kj * (kj + 1) / 2, kk = ki + kj, pb = ri + rj;
                massr[GF ? i2 : 0][GF ? j : 0] = mom[kk * (kk + 1) / 2 + pb];
            }
    }
    int bad = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {                        // Eigen's LLT: unpivoted, lower, row by row
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            double s = L[r * (r + 1) / 2 + c];
#pragma unroll
            for (int k = 0; k < c; ++k) s = __builtin_fma(-L[r * (r + 1) / 2 + k], L[c * (c + 1) / 2 + k], s);
            if (c < r) {
                L[r * (r + 1) / 2 + c] = s * rd[c];
            } else {
                if (!(s > 0.0) && !bad) bad = r + 1;
                rd[r] = fast_rsqrt<2>(s);
                L[r * (r + 1) / 2 + r] = s * rd[r];
            }
        }
    }
    auto forward = [&](double (&v)[NR]) {                 // v <- L^-1 v
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double s = v[r];
#pragma unroll
            for (int k = 0; k < r; ++k) s = __builtin_fma(-L[r * (r + 1) / 2 + k], v[k], s);
            v[r] = s * rd[r];
        }
    };

    // ---- the evaluation points of a face: the reference's q-th point lies at gauss_x[q] from the LOWER-id endpoint
    //      (basic_geom.hpp:202-203, bases.hpp:260-261), its face-basis values are t_q^k
    const uint32_t ids[5] = {idv.x, idv.y, idv.z, idv.w, idv.x};
    auto face_point = [&](int f, int q, double (&phi)[RBS], double &bx_, double &by_) {
        const bool descending = ids[f] > ids[f + 1];
        const double t0 = tab->gauss_x[NFQ][q];
        const double t = descending ? -t0 : t0;
        const double x = 0.5 * (1 - t) * px[f] + 0.5 * (1 + t) * px[f + 1];      // quadratures.hpp:420-428
        const double y = 0.5 * (1 - t) * py[f] + 0.5 * (1 + t) * py[f + 1];
        bx_ = (x - barx) * ih; by_ = (y - bary) * ih;
        double pwx[RD + 1], pwy[RD + 1];
        pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
        for (int e = 1; e <= RD; ++e) { pwx[e] = pwx[e - 1] * bx_; pwy[e] = pwy[e - 1] * by_; }
        int m = 0;
#pragma unroll
        for (int kk = 0; kk <= RD; ++kk)
#pragma unroll
            for (int ii = 0; ii <= kk; ++ii, ++m) phi[m] = pwx[kk - ii] * pwy[ii];   // (px,py) = (k-i, i)  bases.hpp:119-120
    };
    // face terms of gr_rhs  hho.hpp:68-85:  (w_q |F|/2) n = (w_q / 2) (e_y, -e_x): the edge length cancels
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int q = 0; q < NFQ; ++q) {
            double phi[RBS], bx_, by_;
            face_point(f, q, phi, bx_, by_);
            const double hw = 0.5 * tab->gauss_w[NFQ][q];
            const double gnx = ih * hw * ey[f], gny = -ih * hw * ex[f];
            double pwx[RD + 1], pwy[RD + 1];
            pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
            for (int e = 1; e <= RD; ++e) { pwx[e] = pwx[e - 1] * bx_; pwy[e] = pwy[e - 1] * by_; }
            int m = 0;
#pragma unroll
            for (int kk = 0; kk <= RD; ++kk)
#pragma unroll
                for (int ii = 0; ii <= kk; ++ii, ++m) {
                    if (m == 0) continue;
                    const int ex_ = kk - ii, ey_ = ii;
                    const double gx = ex_ == 0 ? 0.0 : (ex_ * gnx) * pwx[ex_ > 0 ? ex_ - 1 : 0] * pwy[ey_];
                    const double gy = ey_ == 0 ? 0.0 : (ey_ * gny) * pwx[ex_] * pwy[ey_ > 0 ? ey_ - 1 : 0];
                    const double wdn = gx + gy;
#pragma unroll
                    for (int k = 0; k < FBS; ++k) G[m - 1][CBS + f * FBS + k] = __builtin_fma(wdn, tab->face[C::FD].fb[q][k], G[m - 1][CBS + f * FBS + k]);
#pragma unroll
                    for (int j = 0; j < CBS; ++j) G[m - 1][j] = __builtin_fma(-wdn, phi[j], G[m - 1][j]);
                }
        }
    }
    // ---- Y = L^-1 gr_rhs, column by column
#pragma unroll
    for (int c = 0; c < MS; ++c) {
        double v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = G[r][c];
        forward(v);
#pragma unroll
        for (int r = 0; r < NR; ++r) G[r][c] = v[r];
    }
    // ---- data = Y^T Y  (hho.hpp:93), packed upper triangle acc[j (j+1)/2 + i], i <= j
    double acc[NSYM];
#pragma unroll
    for (int j = 0; j < MS; ++j)
#pragma unroll
        for (int ii = 0; ii <= j; ++ii) {
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) s = __builtin_fma(G[r][ii], G[r][j], s);
            acc[j * (j + 1) / 2 + ii] = s;
        }

    // ---- stabilization
    int badm = 0;
    if (C::HAS_STAB) {
        // dense fancy form: proj1 = [I 0] - M1^-1 (M2 R)   hho.hpp:184-190, with v^T R = (L^-1 v)^T Y
        double proj1[GF ? CBS : 1][GF ? MS : 1];
        if (GF) {
            double MC[CBS * (CBS + 1) / 2], mrd[CBS];
#pragma unroll
            for (int r = 0; r < CBS; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) MC[r * (r + 1) / 2 + c] = massr[GF ? r : 0][GF ? c : 0];
#pragma unroll
            for (int r = 0; r < CBS; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    double s = MC[r * (r + 1) / 2 + c];
#pragma unroll
                    for (int k = 0; k < c; ++k) s = __builtin_fma(-MC[r * (r + 1) / 2 + k], MC[c * (c + 1) / 2 + k], s);
                    if (c < r) MC[r * (r + 1) / 2 + c] = s * mrd[c];
                    else {
                        if (!(s > 0.0) && !badm) badm = r + 1;
                        mrd[r] = fast_rsqrt<2>(s);
                        MC[r * (r + 1) / 2 + r] = s * mrd[r];
                    }
                }
#pragma unroll
            for (int i2 = 0; i2 < CBS; ++i2) {
                double v[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) v[r] = massr[GF ? i2 : 0][GF ? r + 1 : 0];
                forward(v);
#pragma unroll
                for (int c = 0; c < MS; ++c) {
                    double s = 0.0;
#pragma unroll
                    for (int r = 0; r < NR; ++r) s = __builtin_fma(v[r], G[r][c], s);
                    proj1[GF ? i2 : 0][GF ? c : 0] = s;
                }
            }
            // M1^-1 (.) column by column: forward then backward with chol(M1)
#pragma unroll
            for (int c = 0; c < MS; ++c) {
#pragma unroll
                for (int r = 0; r < CBS; ++r) {
                    double s = proj1[GF ? r : 0][GF ? c : 0];
#pragma unroll
                    for (int k = 0; k < r; ++k) s = __builtin_fma(-MC[r * (r + 1) / 2 + k], proj1[GF ? k : 0][GF ? c : 0], s);
                    proj1[GF ? r : 0][GF ? c : 0] = s * mrd[r];
                }
#pragma unroll
                for (int r = CBS - 1; r >= 0; --r) {
                    double s = proj1[GF ? r : 0][GF ? c : 0];
#pragma unroll
                    for (int k = r + 1; k < CBS; ++k) s = __builtin_fma(-MC[k * (k + 1) / 2 + r], proj1[GF ? k : 0][GF ? c : 0], s);
                    proj1[GF ? r : 0][GF ? c : 0] = s * mrd[r];
                }
#pragma unroll
                for (int r = 0; r < CBS; ++r) proj1[GF ? r : 0][GF ? c : 0] = (r == c ? 1.0 : 0.0) - proj1[GF ? r : 0][GF ? c : 0];
            }
        }
        const FaceTables &ft = tab->face[C::FD];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            // trace / (|F|/2): tr[k][m] = sum_q w_q t_q^k phi_m(x_fq)   hho.hpp:133-140 / 209-216
            constexpr int TC = GF ? RBS : CBS;
            double tr[FBS][TC];
#pragma unroll
            for (int k = 0; k < FBS; ++k)
#pragma unroll
                for (int m = 0; m < TC; ++m) tr[k][m] = 0.0;
#pragma unroll
            for (int q = 0; q < NFQ; ++q) {
                double phi[RBS], bx_, by_;
                face_point(f, q, phi, bx_, by_);
#pragma unroll
                for (int k = 0; k < FBS; ++k)
#pragma unroll
                    for (int m = 0; m < TC; ++m) tr[k][m] = __builtin_fma(ft.cw[q][k], phi[m], tr[k][m]);
            }
            // T_F / (|F|/2), FBS x MS
            double T[FBS][MS];
#pragma unroll
            for (int k = 0; k < FBS; ++k) {
                if (GF) {
                    double v[NR];                        // MR1 R + MR2 proj1   hho.hpp:222-230
#pragma unroll
                    for (int r = 0; r < NR; ++r) v[r] = tr[k][GF ? r + 1 : 0];
                    forward(v);
#pragma unroll
                    for (int c = 0; c < MS; ++c) {
                        double s = 0.0;
#pragma unroll
                        for (int r = 0; r < NR; ++r) s = __builtin_fma(v[r], G[r][c], s);
#pragma unroll
                        for (int m = 0; m < CBS; ++m) s = __builtin_fma(tr[k][m], proj1[GF ? m : 0][GF ? c : 0], s);
                        T[k][c] = s;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < MS; ++c) T[k][c] = c < CBS ? tr[k][c < TC ? c : 0] : 0.0;      // T_F = [trace_F | 0]
                }
            }
            // U_F = sqrt(|F| / 2h) (L^^-1 T_F/(|F|/2) - L^^T E_F), stab += U_F^T U_F
            const double len = s2[f] * fast_rsqrt(s2[f]);
            const double su = fast_sqrt(0.5 * len * hinv);
#pragma unroll
            for (int c = 0; c < MS; ++c) {
#pragma unroll
                for (int k = 0; k < FBS; ++k) {          // forward substitution with L^ (diagonal of lf holds 1 / L^[i][i])
                    double s = T[k][c];
#pragma unroll
                    for (int k2 = 0; k2 < k; ++k2) s = __builtin_fma(-ft.lf[k][k2], T[k2][c], s);
                    T[k][c] = s * ft.lf[k][k];
                }
#pragma unroll
                for (int k = 0; k < FBS; ++k) {
                    const bool own = c >= CBS + f * FBS && c < CBS + (f + 1) * FBS;
                    const double e = own ? ft.lft[k][own ? c - CBS - f * FBS : 0] : 0.0;      // (L^^T E_F)[k][c]
                    T[k][c] = su * (T[k][c] - e);
                }
            }
#pragma unroll
            for (int j = 0; j < MS; ++j)
#pragma unroll
                for (int ii = 0; ii <= j; ++ii) {
                    double s = acc[j * (j + 1) / 2 + ii];
#pragma unroll
                    for (int k = 0; k < FBS; ++k) s = __builtin_fma(T[k][ii], T[k][j], s);
                    acc[j * (j + 1) / 2 + ii] = s;
                }
        }
    }
    if (a.info != nullptr && valid) a.info[i] = bad ? bad : (badm ? 100 + badm : 0);

    // ---- out: the wavefront's matrices are one contiguous run of 64 MS^2 doubles; piece p of a matrix = its doubles [p CH, (p+1) CH)
    double *out = a.lc + i0 * (size_t)(MS * MS);
    const size_t nvalid = a.n - i0 < 64 ? a.n - i0 : 64;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int t = p * CH + e;
            if (t < MS * MS) {
                const int col = t / MS, row = t % MS;                                     // column-major (Eigen's default)
                const int lo = row < col ? row : col, hi = row < col ? col : row;
                stage[lane * CH + e] = acc[hi * (hi + 1) / 2 + lo];
            }
        }
        __syncthreads();
        const int len = p * CH + CH <= MS * MS ? CH : MS * MS - p * CH;                   // the last piece may be shorter
        for (int t = lane; t < 64 * len; t += 64) {
            const int c = t / len, e = t - c * len;
            if ((size_t)c < nvalid) out[(size_t)c * (MS * MS) + p * CH + e] = stage[c * CH + e];
        }
    }
}

template <class C>
hipError_t launch_small_ops(const SmallOpsArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL((hho_small_ops_kernel<C>), dim3((unsigned)((a.n + 63) / 64)), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace pa
