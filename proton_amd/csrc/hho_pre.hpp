// hho_pre.hpp -- the per-cell head of the local-operator path, ONE THREAD PER CELL.
//
// Everything of make_hho_laplacian (hho.hpp:32-96) that depends on the cell alone and is a short serial
// chain -- geometry (basic_geom.hpp:247-334), the cell quadrature (quadratures.hpp:311-402), the
// moments of the scaled monomials, the stiffness matrix (hho.hpp:55-61) and the Cholesky factor of
// gr_lhs = stiff[1:,1:] (hho.hpp:63,92) -- runs here with the whole state of a cell in the registers of
// one lane: no LDS, no barriers, no cross-lane traffic, 64 cells per wavefront.  The cooperative kernel
// (hho_device.hpp, G lanes per cell) executed the same chain with 9 of 32 lanes useful and three LDS round
// trips per pivot.  The result travels through a small per-cell record in HBM (Cfg::Pre): the packed
// factor L with its true diagonal, the reciprocal diagonal, sqrt(|F|/2h) of the four faces, barycenter,
// 2/h_T, the pivot status, the four vertices and the orientation bits of the faces -- everything the
// cooperative kernel needs of the cell, so that the record is its only input.  The stiffness itself does not travel: gr_rhs[:, c] = stiff[1:, c] - F_c for
// a cell column c, and L^-1 stiff[1:, c] = L^T e_(c-1) exactly, so the consumer adds row c-1 of L to
// L^-1 (-F_c).
#pragma once
#include "hho_device.hpp"

// occupancy the pre-pass is compiled for (2nd __launch_bounds__ argument; 1 = whatever its registers allow)
#ifndef PA_PRE_WAVES
#define PA_PRE_WAVES 1
#endif
namespace pa {


// The head of ONE cell, all of it in the registers of the calling lane; `out` = where pair 0 of the cell's record goes, the
// other 16-byte pairs follow at a stride of 16 doubles (the tiles of 8 records described below).  Called by the pre-pass
// kernel (one thread per cell of a piece) and, in the instances built with PA_SELF_PRE, by the cooperative kernel itself: there a
// wavefront stops every 64 / CPW passes and forms the heads of the 64 cells it will visit next, one per lane, into a ring of 64
// records of its own (hho_device.hpp).
template <class C>
__device__ __forceinline__ void cell_pre_record(const PreArgs &a, size_t cell, double *out)
{
    constexpr int RD = C::RD, NR = C::NR, NPW = C::NPW, NMOM = C::NMOM;
    typedef typename C::Pre PL;
    const QuadTables *__restrict__ tab = a.tab;

    // ---- geometry (the same expressions as S0 of the cooperative kernel)
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
        const double d1 = (ax * by - ay * bx) * 0.5, d2 = (bx * cy - by * cx) * 0.5;
        const double rx = (ax + bx) * d1 + (bx + cx) * d2, ry = (ay + by) * d1 + (by + cy) * d2;
        const double iden = fast_rcp((d1 + d2) * 3);
        barx = px0 + rx * iden; bary = py0 + ry * iden;
    }
    const double e0x = px1 - px0, e0y = py1 - py0, e1x = px2 - px1, e1y = py2 - py1;
    const double e2x = px3 - px2, e2y = py3 - py2, e3x = px0 - px3, e3y = py0 - py3;
    const double s0 = e0x * e0x + e0y * e0y, s1 = e1x * e1x + e1y * e1y;
    const double s2 = e2x * e2x + e2y * e2y, s3 = e3x * e3x + e3y * e3y;
    double h2;                                  // diameter^2  basic_geom.hpp:288-305
    {
        const double d02x = px2 - px0, d02y = py2 - py0, d13x = px3 - px1, d13y = py3 - py1;
        h2 = fmax(fmax(s0, s1), fmax(s2, s3));
        h2 = fmax(h2, fmax(d02x * d02x + d02y * d02y, d13x * d13x + d13y * d13y));
    }
    const double rh = fast_rsqrt(h2);           // 1 / h_T
    const double ih = 2.0 * rh;                 // bases.hpp:98-99,142
    double hinv = rh;                           // fancy: h = cell diameter  hho.hpp:201
    if (C::NAIVE) {                             // naive: h = cell area      hho.hpp:119
        const double ux = px1 - px0, uy = py1 - py0, vx = px2 - px0, vy = py2 - py0;
        const double wx = px3 - px0, wy = py3 - py0;
        hinv = fast_rcp(fabs(ux * vy - uy * vx) * 0.5 + fabs(vx * wy - vy * wx) * 0.5);
    }
    // The records are stored in tiles of 8 cells, pair by pair: [tile][16-byte pair][cell % 8].  The 8 lanes of a
    // tile write one full 128-byte line per store instruction (one record per lane, 496-byte stride, made every
    // 16-byte piece its own request at the L2: 0.23 ms of the k = 2 pre-pass was that); the consumer reads a record
    // as pairs 128 bytes apart, one cell ahead of its use.
    auto put = [&](int e, double2 v) { *reinterpret_cast<double2 *>(out + (e / 2) * 16) = v; };
    {
        double su[4];
        const double sq[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const double len = sq[f] * fast_rsqrt(sq[f]);
            su[f] = C::HAS_STAB ? fast_sqrt(0.5 * len * hinv) : 0.0;
        }
        put(PL::oSCAL, double2{su[0], su[1]});
        put(PL::oSCAL + 2, double2{su[2], su[3]});
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
        for (int k = 0; k < NPW; ++k)                          // graded ordering: total degree k, then r  bases.hpp:114-128
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
                const double x = n0 * px0 + n1 * px1 + n2 * px2 + n3 * px3;
                const double y = n0 * py0 + n1 * py1 + n2 * py2 + n3 * py3;
                const double j11 = e0x * am - e2x * ap, j12 = e0y * am - e2y * ap;
                const double j21 = e1x * bp - e3x * bm, j22 = e1y * bp - e3y * bm;
                add_point(x, y, rw * fabs(j11 * j22 - j12 * j21));      // quadratures.hpp:331-352
            }
        }
    } else {
        constexpr int R = C::QDEG == 0 ? 1 : C::QDEG;                    // rules[deg]  quadratures.hpp:257
        const double vx[5] = {px0, px1, px2, px3, px0}, vy[5] = {py0, py1, py2, py3, py0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                    // fan triangle (p_t, p_t+1, bar)  quadratures.hpp:390-396
            const double ax = vx[t], ay = vy[t], bx = vx[t + 1], by = vy[t + 1];
            const double v0x = bx - ax, v0y = by - ay, v1x = barx - ax, v1y = bary - ay;
            const double tarea = fabs((v0x * v1y - v0y * v1x) * 0.5);    // quadratures.hpp:248-251
#pragma unroll
            for (int row = 0; row < C::NT; ++row) {
                const double l0 = tab->dun[R][row][0], l1 = tab->dun[R][row][1], l2 = tab->dun[R][row][2];
                add_point(ax * l0 + bx * l1 + barx * l2, ay * l0 + by * l1 + bary * l2, tarea * tab->dun[R][row][3]);
            }
        }
    }

    // the moments of degree <= recdeg - 2 + celdeg travel: the consumer forms the cell columns of gr_rhs from them (Cfg::LAPG)
#pragma unroll
    for (int e = 0; e < C::NMG; e += 2) put(PL::oSCAL + 18 + e, double2{mom[e], e + 1 < C::NMG ? mom[e + 1 < NMOM ? e + 1 : 0] : 0.0});

    // ---- gr_lhs = stiff[1:,1:] from the moments, Cholesky row by row (Eigen's LLT: unpivoted, lower)
    //   stiff(i,j) = ih^2 (a a' MOM(a+a'-2, b+b') + b b' MOM(a+a', b+b'-2))   bases.hpp:170-176, hho.hpp:57-61
    const double ih2 = ih * ih;
    double L[NR * (NR + 1) / 2], rd[NR];
    // monomial mi = (ki, ri) has exponents (ki - ri, ri); row r = mi - 1 of gr_lhs (the constant is dropped)
#pragma unroll
    for (int ki = 1; ki <= RD; ++ki)
#pragma unroll
        for (int ri = 0; ri <= ki; ++ri)
#pragma unroll
            for (int kj = 1; kj <= ki; ++kj)
#pragma unroll
                for (int rj = 0; rj <= kj; ++rj) {
                    const int mi = ki * (ki + 1) / 2 + ri, mj = kj * (kj + 1) / 2 + rj;
                    if (mj <= mi) {
                        const int ai = ki - ri, bi = ri, aj = kj - rj, bj = rj;
                        const int c1 = ai * aj, c2 = bi * bj;
                        const int k1 = ai + aj - 2 + bi + bj, k2 = ai + aj + bi + bj - 2;      // total degrees of the two moments
                        const double m1 = c1 ? mom[k1 * (k1 + 1) / 2 + (bi + bj)] : 0.0;
                        const double m2 = c2 ? mom[k2 * (k2 + 1) / 2 + (bi + bj - 2)] : 0.0;
                        L[(mi - 1) * mi / 2 + (mj - 1)] = ih2 * ((double)c1 * m1 + (double)c2 * m2);
                    }
                }
    int bad = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
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
                L[r * (r + 1) / 2 + r] = s * rd[r];              // true diagonal
            }
        }
    }

    // ---- the record
    constexpr int NL = PL::NL;
#pragma unroll
    for (int e = 0; e + 1 < NL; e += 2) put(e, double2{L[e], L[e + 1]});
    {
        // the odd tail of L shares its 16 bytes with the first reciprocal
        double lin[PL::oSCAL - (NL & ~1)];
#pragma unroll
        for (int e = 0; e < PL::oSCAL - (NL & ~1); ++e) {
            const int src = (NL & ~1) + e;
            lin[e] = src < NL ? L[src < NL ? src : 0] : (src - NL < NR ? rd[src - NL < NR ? src - NL : 0] : 0.0);
        }
#pragma unroll
        for (int e = 0; e + 1 < PL::oSCAL - (NL & ~1); e += 2)
            put((NL & ~1) + e, double2{lin[e], lin[e + 1]});
    }
    put(PL::oSCAL + 4, double2{barx, bary});
    put(PL::oSCAL + 6, double2{ih, (double)bad});
    put(PL::oSCAL + 8, q0);
    put(PL::oSCAL + 10, q1);
    put(PL::oSCAL + 12, q2);
    put(PL::oSCAL + 14, q3);
    // a face runs from its LOWER-id endpoint (basic_geom.hpp:202-203, bases.hpp:260-261): bit f = local face f is
    // traversed against that direction by the cell's CCW vertex order
    const int flags = (idv.x > idv.y ? 1 : 0) | (idv.y > idv.z ? 2 : 0) | (idv.z > idv.w ? 4 : 0) | (idv.w > idv.x ? 8 : 0);
    // ---- dense fancy form: rows i < CBS of the cell mass matrix (M2 of hho.hpp:185 is its columns 1..) and the Cholesky
    // factor of M1 = mass[:cbs, :cbs] (hho.hpp:184,188), both from the same moments
    int badm = 0;
    if (C::GENERAL_FANCY) {
        constexpr int CBS = C::CBS, CD = C::CD, NT = C::GENERAL_FANCY ? PL::NPRE - PL::oMR : 2;      // (sizes > 0 in the instances without it)
        double tail[NT];
#pragma unroll
        for (int e = 0; e < NT; ++e) tail[e] = 0.0;
        double MC[CBS * (CBS + 1) / 2], mrd[CBS];
#pragma unroll
        for (int kj = 0; kj <= RD; ++kj)
#pragma unroll
            for (int rj = 0; rj <= kj; ++rj)
#pragma unroll
                for (int ki = 0; ki <= CD; ++ki)
#pragma unroll
                    for (int ri = 0; ri <= ki; ++ri) {
                        const int mi = ki * (ki + 1) / 2 + ri, mj = kj * (kj + 1) / 2 + rj;
                        const int pb = ri + rj, kk = ki + kj;                   // phi_i phi_j = bx^(kk - pb) by^pb
                        const double v = mom[kk * (kk + 1) / 2 + pb];
                        tail[mj * CBS + mi] = v;
                        if (mj <= mi) MC[mi * (mi + 1) / 2 + (mj < CBS ? mj : 0)] = v;
                    }
#pragma unroll
        for (int r = 0; r < CBS; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double s = MC[r * (r + 1) / 2 + c];
#pragma unroll
                for (int k = 0; k < c; ++k) s = __builtin_fma(-MC[r * (r + 1) / 2 + k], MC[c * (c + 1) / 2 + k], s);
                if (c < r) {
                    MC[r * (r + 1) / 2 + c] = s * mrd[c];
                } else {
                    if (!(s > 0.0) && !badm) badm = r + 1;
                    mrd[r] = fast_rsqrt<2>(s);
                    MC[r * (r + 1) / 2 + r] = s * mrd[r];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < PL::NMC; ++e) tail[PL::NMR + e] = MC[e];
#pragma unroll
        for (int e = 0; e < CBS; ++e) tail[PL::NMR + PL::NMC + e] = mrd[e];
#pragma unroll
        for (int e = 0; e + 1 < NT; e += 2) put(PL::oMR + e, double2{tail[e], tail[e + 1]});
    }
    put(PL::oSCAL + 16, double2{(double)flags, (double)badm});
}

template <class C>
__global__ __launch_bounds__(64, PA_PRE_WAVES) void hho_cell_pre_kernel(PreArgs a)
{
    typedef typename C::Pre PL;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    cell_pre_record<C>(a, a.first + i, a.pre + (((i >> 3) * (size_t)PL::NP2) * 8 + (i & 7)) * 2);
}

}  // namespace pa
