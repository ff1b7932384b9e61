// quad_tables.hpp -- host-side construction of the quadrature tables the kernels read
// from __constant__ memory.
//   Gauss-Legendre: closed forms and emission order of gauss_legendre()
//                   (src/core/core_bits/quadratures.hpp:78-158); the rules with six to eight nodes
//                   are golub_welsch's (:32-75): ascending nodes.
//   Dunavant:       the rules triangle_quadrature() indexes (quadratures.hpp:238-271,
//                   quadratures_dunavant.hpp:27-130), kept with the reference's 15 printed
//                   digits, expanded from their symmetry orbits in the reference's row order.
#pragma once

#include <cmath>
#include <cstring>

#include "hho_device.hpp"

namespace pa {

inline void fill_gauss(QuadTables &t)
{
    std::memset(t.gauss_x, 0, sizeof(t.gauss_x));
    std::memset(t.gauss_w, 0, sizeof(t.gauss_w));
    // n = 1
    t.gauss_x[1][0] = 0.0; t.gauss_w[1][0] = 2.0;
    // n = 2
    {
        const double q = 1.0 / std::sqrt(3.0);
        t.gauss_x[2][0] = -q; t.gauss_x[2][1] = q;
        t.gauss_w[2][0] = 1.0; t.gauss_w[2][1] = 1.0;
    }
    // n = 3: -q, +q, 0
    {
        const double q = std::sqrt(3.0 / 5.0);
        t.gauss_x[3][0] = -q; t.gauss_x[3][1] = q; t.gauss_x[3][2] = 0.0;
        t.gauss_w[3][0] = 5.0 / 9.0; t.gauss_w[3][1] = 5.0 / 9.0; t.gauss_w[3][2] = 8.0 / 9.0;
    }
    // n = 4: inner pair then outer pair
    {
        const double a1 = 3.0 / 7.0, a2 = 2.0 * std::sqrt(6.0 / 5.0) / 7.0;
        const double qi = std::sqrt(a1 - a2), wi = (18.0 + std::sqrt(30.0)) / 36.0;
        const double qo = std::sqrt(a1 + a2), wo = (18.0 - std::sqrt(30.0)) / 36.0;
        t.gauss_x[4][0] = -qi; t.gauss_x[4][1] = qi; t.gauss_x[4][2] = -qo; t.gauss_x[4][3] = qo;
        t.gauss_w[4][0] = wi; t.gauss_w[4][1] = wi; t.gauss_w[4][2] = wo; t.gauss_w[4][3] = wo;
    }
    // n = 5: 0, inner pair, outer pair
    {
        const double a1 = 5.0, a2 = 2.0 * std::sqrt(10.0 / 7.0);
        const double qi = std::sqrt(a1 - a2) / 3.0, wi = (322 + 13.0 * std::sqrt(70.0)) / 900.0;
        const double qo = std::sqrt(a1 + a2) / 3.0, wo = (322 - 13.0 * std::sqrt(70.0)) / 900.0;
        t.gauss_x[5][0] = 0.0; t.gauss_w[5][0] = 128.0 / 225.0;
        t.gauss_x[5][1] = -qi; t.gauss_x[5][2] = qi; t.gauss_w[5][1] = wi; t.gauss_w[5][2] = wi;
        t.gauss_x[5][3] = -qo; t.gauss_x[5][4] = qo; t.gauss_w[5][3] = wo; t.gauss_w[5][4] = wo;
    }
    // n = 6 .. 8: what golub_welsch (quadratures.hpp:32-75) returns -- the Gauss-Legendre rule with its nodes in ASCENDING
    // order (eigenvalues of the Jacobi matrix as Eigen's SelfAdjointEigenSolver lists them).  Reached for quadrature degree
    // >= 10, i.e. make_rhs / project_function at cell degree 4 with a degree increase; never by the local operators of k <= 3.
    // Nodes by Newton's iteration on P_n from the Chebyshev guess, weights 2 / ((1 - x^2) P_n'(x)^2): the same numbers as the
    // reference's eigen-solve to rounding.
    for (int n = 6; n <= 8; ++n) {
        for (int i = 0; i < n; ++i) {
            double x = -std::cos(M_PI * (i + 0.75) / (n + 0.5)), dp = 1.0;
            for (int it = 0; it < 100; ++it) {
                double p0 = 1.0, p1 = x;
                for (int k = 2; k <= n; ++k) { const double p2 = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k; p0 = p1; p1 = p2; }
                dp = n * (x * p1 - p0) / (x * x - 1.0);
                const double dx = p1 / dp;
                x -= dx;
                if (std::fabs(dx) < 1e-16) break;
            }
            t.gauss_x[n][i] = x;
            t.gauss_w[n][i] = 2.0 / ((1.0 - x * x) * dp * dp);
        }
    }
}

// Face tables (hho_device.hpp FaceTables): for face degree fd the face basis at the n = fd + 1 Gauss
// points is t_q^k (bases.hpp:269-272 on the segment), the face mass matrix is (|F|/2) M^ with
// M^[i][j] = sum_q w_q t_q^(i+j) (hho.hpp:138,214); L^ is its Cholesky factor.
inline void fill_face_tables(QuadTables &t)
{
    std::memset(t.face, 0, sizeof(t.face));
    for (int fd = 0; fd < 4; ++fd) {
        FaceTables &f = t.face[fd];
        const int n = fd + 1, fbs = fd + 1;
        for (int q = 0; q < n; ++q) {
            double v = 1.0;
            for (int k = 0; k < fbs; ++k) {
                f.fb[q][k] = v;
                f.cw[q][k] = t.gauss_w[n][q] * v;
                v *= t.gauss_x[n][q];
            }
        }
        double M[4][4] = {{0}}, L[4][4] = {{0}};
        for (int i = 0; i < fbs; ++i)
            for (int j = 0; j < fbs; ++j)
                for (int q = 0; q < n; ++q) M[i][j] += f.cw[q][i] * f.fb[q][j];
        for (int j = 0; j < fbs; ++j) {
            double d = M[j][j];
            for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            L[j][j] = std::sqrt(d);
            for (int i = j + 1; i < fbs; ++i) {
                double s = M[i][j];
                for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
                L[i][j] = s / L[j][j];
            }
        }
        for (int i = 0; i < fbs; ++i)
            for (int k = 0; k <= i; ++k) {
                f.lf[i][k] = (i == k) ? 1.0 / L[i][i] : L[i][k];
                f.lft[k][i] = L[i][k];
            }
    }
}

struct DunavantOrbit { int mult; double a, b, c, w; };   // mult 1: (a,a,a); 3: one a two b; 6: all distinct

inline void fill_dunavant(QuadTables &t)
{
    static const DunavantOrbit r1[] = {{1, 0.333333333333333, 0, 0, 1.000000000000000}};
    static const DunavantOrbit r2[] = {{3, 0.666666666666667, 0.166666666666667, 0, 0.333333333333333}};
    static const DunavantOrbit r3[] = {{1, 0.333333333333333, 0, 0, -0.562500000000000},
                                       {3, 0.600000000000000, 0.200000000000000, 0, 0.520833333333333}};
    static const DunavantOrbit r4[] = {{3, 0.108103018168070, 0.445948490915965, 0, 0.223381589678011},
                                       {3, 0.816847572980459, 0.091576213509771, 0, 0.109951743655322}};
    static const DunavantOrbit r5[] = {{1, 0.333333333333333, 0, 0, 0.225000000000000},
                                       {3, 0.059715871789770, 0.470142064105115, 0, 0.132394152788506},
                                       {3, 0.797426985353087, 0.101286507323456, 0, 0.125939180544827}};
    static const DunavantOrbit r6[] = {{3, 0.501426509658179, 0.249286745170910, 0, 0.116786275726379},
                                       {3, 0.873821971016996, 0.063089014491502, 0, 0.050844906370207},
                                       {6, 0.053145049844817, 0.310352451033784, 0.636502499121399, 0.082851075618374}};
    static const DunavantOrbit r7[] = {{1, 0.333333333333333, 0, 0, -0.149570044467682},
                                       {3, 0.479308067841920, 0.260345966079040, 0, 0.175615257433208},
                                       {3, 0.869739794195568, 0.065130102902216, 0, 0.053347235608838},
                                       {6, 0.048690315425316, 0.312865496004874, 0.638444188569810, 0.077113760890257}};
    static const DunavantOrbit r8[] = {{1, 0.333333333333333, 0, 0, 0.144315607677787},
                                       {3, 0.081414823414554, 0.459292588292723, 0, 0.095091634267285},
                                       {3, 0.658861384496480, 0.170569307751760, 0, 0.103217370534718},
                                       {3, 0.898905543365938, 0.050547228317031, 0, 0.032458497623198},
                                       {6, 0.008394777409958, 0.263112829634638, 0.728492392955404, 0.027230314174435}};
    static const struct { int n; const DunavantOrbit *o; } rules[9] = {
        {1, r1}, {1, r2}, {2, r3}, {2, r4}, {3, r5}, {3, r6}, {4, r7}, {5, r8}, {0, nullptr}};   // [8]: the empty sentinel

    std::memset(t.dun, 0, sizeof(t.dun));
    for (int r = 0; r < 9; ++r) {
        int n = 0;
        auto put = [&](double l0, double l1, double l2, double w) {
            t.dun[r][n][0] = l0; t.dun[r][n][1] = l1; t.dun[r][n][2] = l2; t.dun[r][n][3] = w; ++n;
        };
        for (int k = 0; k < rules[r].n; ++k) {
            const DunavantOrbit &q = rules[r].o[k];
            if (q.mult == 1) put(q.a, q.a, q.a, q.w);
            else if (q.mult == 3) { put(q.a, q.b, q.b, q.w); put(q.b, q.a, q.b, q.w); put(q.b, q.b, q.a, q.w); }
            else {
                put(q.a, q.b, q.c, q.w); put(q.a, q.c, q.b, q.w); put(q.b, q.a, q.c, q.w);
                put(q.b, q.c, q.a, q.w); put(q.c, q.a, q.b, q.w); put(q.c, q.b, q.a, q.w);
            }
        }
        t.dun_n[r] = n;
    }
}

}  // namespace pa
