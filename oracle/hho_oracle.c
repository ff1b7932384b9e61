/*
 * hho_oracle.c -- CPU restatement of ProtoN's per-cell HHO local operators.
 * TEST INFRASTRUCTURE ONLY (see hho_oracle.h for the pinning status).
 *
 * Compile like the reference Release build (CMakeLists.txt:16): -O3 -mavx
 * (no FMA contraction on that target, so plain mul/add rounding as in the
 * reference's Eigen loops).
 */
#include "hho_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX(i, j, ld) ((size_t)(i) + (size_t)(j) * (size_t)(ld))

/* ------------------------------------------------------------------ */
/* utils.hpp:62-111                                                    */
/* ------------------------------------------------------------------ */
hho_degrees hho_degree_info1(int degree)
{
    hho_degrees d = { degree, degree, degree + 1 };
    return d;
}

hho_degrees hho_degree_info2(int cd, int fd, int *fell_back)
{
    hho_degrees d;
    int c1 = fd > 0 && (cd == fd - 1 || cd == fd || cd == fd + 1);
    int c2 = fd == 0 && (cd == fd || cd == fd + 1);
    if (fell_back) *fell_back = !(c1 || c2);
    if (c1 || c2) { d.cell_deg = cd; d.face_deg = fd; d.rec_deg = fd + 1; }
    else          { d.cell_deg = fd; d.face_deg = fd; d.rec_deg = fd + 1; } /* utils.hpp:88-91 */
    return d;
}

/* ------------------------------------------------------------------ */
/* bases.hpp:27-50  square-and-multiply, same association order        */
/* ------------------------------------------------------------------ */
double hho_iexp_pow(double x, size_t n)
{
    if (n == 0) return 1;
    double y = 1;
    while (n > 1) {
        if (n % 2 == 0) { x = x * x; n = n / 2; }
        else            { y = x * y; x = x * x; n = (n - 1) / 2; }
    }
    return x * y;
}

/* ------------------------------------------------------------------ */
/* golub_welsch, quadratures.hpp:32-75: nodes = eigenvalues (ascending, the order of Eigen's SelfAdjointEigenSolver) of  */
/* the Jacobi matrix with off-diagonal sqrt(1 / (4 - 1/i^2)), weights = 2 (first component of the unit eigenvector)^2.  */
/* The eigen-solve is a cyclic Jacobi iteration on the dense symmetric matrix (n <= 8): the reference's solver is Eigen's */
/* tridiagonal QR; both deliver eigenpairs to working precision.                                                        */
/* ------------------------------------------------------------------ */
int hho_golub_welsch(int n, double *nd, double *wt)
{
    double A[HHO_MAX_GAUSS][HHO_MAX_GAUSS], V[HHO_MAX_GAUSS][HHO_MAX_GAUSS];
    if (n < 1 || n > HHO_MAX_GAUSS) return -HHO_ERR_DEGREE;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { A[i][j] = 0.0; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int i = 1; i < n; i++) { double p = 4.0 - 1.0 / ((double)i * i); A[i][i - 1] = A[i - 1][i] = sqrt(1.0 / p); }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                if (A[p][q] == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    int order[HHO_MAX_GAUSS];
    for (int i = 0; i < n; i++) order[i] = i;
    for (int i = 1; i < n; i++) {                        /* ascending eigenvalues */
        int o = order[i], j = i - 1;
        while (j >= 0 && A[order[j]][order[j]] > A[o][o]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = o;
    }
    for (int i = 0; i < n; i++) { nd[i] = A[order[i]][order[i]]; wt[i] = 2.0 * V[0][order[i]] * V[0][order[i]]; }
    return n;
}

/* ------------------------------------------------------------------ */
/* quadratures.hpp:78-158                                              */
/* ------------------------------------------------------------------ */
int hho_gauss_legendre(int degree, double *nd, double *wt)
{
    int comp = degree;
    if (degree % 2 == 0) comp = degree + 1;
    int n = (comp + 1) / 2;
    double qp, qw, a1, a2;
    switch (n) {
    case 1:
        nd[0] = 0.0; wt[0] = 2.0; return 1;
    case 2:
        qp = 1.0 / sqrt(3.0); qw = 1.0;
        nd[0] = -qp; wt[0] = qw; nd[1] = qp; wt[1] = qw; return 2;
    case 3:
        qp = sqrt(3.0 / 5.0); qw = 5.0 / 9.0;
        nd[0] = -qp; wt[0] = qw; nd[1] = qp; wt[1] = qw;
        nd[2] = 0.0; wt[2] = 8.0 / 9.0; return 3;
    case 4:
        a1 = 3.0 / 7.0; a2 = 2.0 * sqrt(6.0 / 5.0) / 7.0;
        qp = sqrt(a1 - a2); qw = (18.0 + sqrt(30.0)) / 36.0;
        nd[0] = -qp; wt[0] = qw; nd[1] = qp; wt[1] = qw;
        qp = sqrt(a1 + a2); qw = (18.0 - sqrt(30.0)) / 36.0;
        nd[2] = -qp; wt[2] = qw; nd[3] = qp; wt[3] = qw; return 4;
    case 5:
        nd[0] = 0.0; wt[0] = 128.0 / 225.0;
        a1 = 5.0; a2 = 2.0 * sqrt(10.0 / 7.0);
        qp = sqrt(a1 - a2) / 3.0; qw = (322 + 13.0 * sqrt(70.0)) / 900.0;
        nd[1] = -qp; wt[1] = qw; nd[2] = qp; wt[2] = qw;
        qp = sqrt(a1 + a2) / 3.0; qw = (322 - 13.0 * sqrt(70.0)) / 900.0;
        nd[3] = -qp; wt[3] = qw; nd[4] = qp; wt[4] = qw; return 5;
    default:
        if (n > HHO_MAX_GAUSS) return -HHO_ERR_DEGREE;
        return hho_golub_welsch(n, nd, wt);
    }
}

/* ------------------------------------------------------------------ */
/* Dunavant rules, quadratures_dunavant.hpp:27-130.                    */
/* Stored by symmetry orbit; expanded in the reference's row order.    */
/*   kind 1: centroid (a,a,a)                                          */
/*   kind 3: (a,b,b),(b,a,b),(b,b,a)                                   */
/*   kind 6: (a,b,c),(a,c,b),(b,a,c),(b,c,a),(c,a,b),(c,b,a)           */
/* Constants keep the reference's 15 printed digits.                   */
/* ------------------------------------------------------------------ */
typedef struct { int kind; double a, b, c, w; } dun_orbit;

static const dun_orbit dun_r1[] = { {1, 0.333333333333333, 0, 0, 1.000000000000000} };
static const dun_orbit dun_r2[] = { {3, 0.666666666666667, 0.166666666666667, 0, 0.333333333333333} };
static const dun_orbit dun_r3[] = { {1, 0.333333333333333, 0, 0, -0.562500000000000},
                                    {3, 0.600000000000000, 0.200000000000000, 0, 0.520833333333333} };
static const dun_orbit dun_r4[] = { {3, 0.108103018168070, 0.445948490915965, 0, 0.223381589678011},
                                    {3, 0.816847572980459, 0.091576213509771, 0, 0.109951743655322} };
static const dun_orbit dun_r5[] = { {1, 0.333333333333333, 0, 0, 0.225000000000000},
                                    {3, 0.059715871789770, 0.470142064105115, 0, 0.132394152788506},
                                    {3, 0.797426985353087, 0.101286507323456, 0, 0.125939180544827} };
static const dun_orbit dun_r6[] = { {3, 0.501426509658179, 0.249286745170910, 0, 0.116786275726379},
                                    {3, 0.873821971016996, 0.063089014491502, 0, 0.050844906370207},
                                    {6, 0.053145049844817, 0.310352451033784, 0.636502499121399, 0.082851075618374} };
static const dun_orbit dun_r7[] = { {1, 0.333333333333333, 0, 0, -0.149570044467682},
                                    {3, 0.479308067841920, 0.260345966079040, 0, 0.175615257433208},
                                    {3, 0.869739794195568, 0.065130102902216, 0, 0.053347235608838},
                                    {6, 0.048690315425316, 0.312865496004874, 0.638444188569810, 0.077113760890257} };
static const dun_orbit dun_r8[] = { {1, 0.333333333333333, 0, 0, 0.144315607677787},
                                    {3, 0.081414823414554, 0.459292588292723, 0, 0.095091634267285},
                                    {3, 0.658861384496480, 0.170569307751760, 0, 0.103217370534718},
                                    {3, 0.898905543365938, 0.050547228317031, 0, 0.032458497623198},
                                    {6, 0.008394777409958, 0.263112829634638, 0.728492392955404, 0.027230314174435} };

typedef struct { int n_orbits; const dun_orbit *orbits; } dun_rule;
/* rules[] is 0-based in the reference: rules[i] == rule_{i+1}; rules[8] is the {0,NULL} sentinel
 * (quadratures_dunavant.hpp:120-130).                                                            */
static const dun_rule dun_rules[9] = {
    {1, dun_r1}, {1, dun_r2}, {2, dun_r3}, {2, dun_r4}, {3, dun_r5},
    {3, dun_r6}, {4, dun_r7}, {5, dun_r8}, {0, NULL}
};

static int dun_expand(int idx, double (*rows)[4])
{
    const dun_rule *r = &dun_rules[idx];
    int n = 0;
    for (int o = 0; o < r->n_orbits; o++) {
        const dun_orbit *q = &r->orbits[o];
        double a = q->a, b = q->b, c = q->c, w = q->w;
        if (q->kind == 1) {
            rows[n][0] = a; rows[n][1] = a; rows[n][2] = a; rows[n][3] = w; n++;
        } else if (q->kind == 3) {
            double t[3][3] = { {a, b, b}, {b, a, b}, {b, b, a} };
            for (int k = 0; k < 3; k++) { rows[n][0] = t[k][0]; rows[n][1] = t[k][1]; rows[n][2] = t[k][2]; rows[n][3] = w; n++; }
        } else {
            double t[6][3] = { {a, b, c}, {a, c, b}, {b, a, c}, {b, c, a}, {c, a, b}, {c, b, a} };
            for (int k = 0; k < 6; k++) { rows[n][0] = t[k][0]; rows[n][1] = t[k][1]; rows[n][2] = t[k][2]; rows[n][3] = w; n++; }
        }
    }
    return n;
}

/* quadratures.hpp:238-271 */
int hho_triangle_quadrature(const double p0[2], const double p1[2], const double p2[2],
                            int deg, double *qx, double *qy, double *qw)
{
    if (deg == 0) deg = 1;
    if (deg > 8) return -HHO_ERR_QUADRATURE;
    double v0x = p1[0] - p0[0], v0y = p1[1] - p0[1];
    double v1x = p2[0] - p0[0], v1y = p2[1] - p0[1];
    double area = fabs((v0x * v1y - v0y * v1x) / 2.0);
    double rows[16][4];
    int n = dun_expand(deg, rows);              /* rules[deg] == rule_{deg+1}: the off-by-one */
    for (int i = 0; i < n; i++) {
        qx[i] = p0[0] * rows[i][0] + p1[0] * rows[i][1] + p2[0] * rows[i][2];
        qy[i] = p0[1] * rows[i][0] + p1[1] * rows[i][1] + p2[1] * rows[i][2];
        qw[i] = area * rows[i][3];
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* basic_geom.hpp                                                      */
/* ------------------------------------------------------------------ */
void hho_cell_barycenter(const double pts[8], double bar[2])   /* :247-278 */
{
    double rx = 0.0, ry = 0.0, den = 0.0;
    double p0x = pts[0], p0y = pts[1];
    for (int i = 2; i < 4; i++) {
        double ax = pts[2 * (i - 1)] - p0x, ay = pts[2 * (i - 1) + 1] - p0y;
        double bx = pts[2 * i] - p0x,       by = pts[2 * i + 1] - p0y;
        double d = (ax * by - ay * bx) / 2.0;
        rx = rx + (ax + bx) * d;
        ry = ry + (ay + by) * d;
        den += d;
    }
    bar[0] = p0x + rx / (den * 3);
    bar[1] = p0y + ry / (den * 3);
}

double hho_cell_diameter(const double pts[8])                  /* :288-305 */
{
    double diam = 0.0;
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++) {
            double dx = pts[2 * j] - pts[2 * i], dy = pts[2 * j + 1] - pts[2 * i + 1];
            double d = sqrt(dx * dx + dy * dy);
            diam = d > diam ? d : diam;
        }
    return diam;
}

double hho_cell_measure(const double pts[8])                   /* :317-334 */
{
    double acc = 0.0;
    for (int i = 1; i < 3; i++) {
        double ux = pts[2 * i] - pts[0],       uy = pts[2 * i + 1] - pts[1];
        double vx = pts[2 * (i + 1)] - pts[0], vy = pts[2 * (i + 1) + 1] - pts[1];
        acc += fabs(ux * vy - uy * vx) * 0.5;
    }
    return acc;
}

void hho_cell_normals(const double pts[8], double n[8])        /* :349-372 */
{
    for (int i = 0; i < 4; i++) {
        int k = (i + 1) % 4;
        double vx = pts[2 * k] - pts[2 * i], vy = pts[2 * k + 1] - pts[2 * i + 1];
        double nx = vy, ny = -vx;
        double nrm = sqrt(nx * nx + ny * ny);
        n[2 * i] = nx / nrm; n[2 * i + 1] = ny / nrm;
    }
}

void hho_cell_face_points(const double pts[8], const uint64_t ids[4], int f, double a[2], double b[2])
{
    int i0 = f, i1 = (f + 1) % 4;
    if (ids[i0] > ids[i1]) { int t = i0; i0 = i1; i1 = t; }   /* basic_geom.hpp:202-203 */
    a[0] = pts[2 * i0]; a[1] = pts[2 * i0 + 1];
    b[0] = pts[2 * i1]; b[1] = pts[2 * i1 + 1];
}

/* ------------------------------------------------------------------ */
/* quadratures.hpp:311-432                                             */
/* ------------------------------------------------------------------ */
static int tensor_cell_quadrature(const double p[8], int degree, double *qx, double *qy, double *qw)
{
    double nd[HHO_MAX_GAUSS], wt[HHO_MAX_GAUSS];
    int n = hho_gauss_legendre(degree, nd, wt);
    if (n < 0) return n;
    int k = 0;
    for (int j = 0; j < n; j++) {           /* outer: eta (:355) */
        for (int i = 0; i < n; i++) {       /* inner: xi  (:357) */
            double xi = nd[i], eta = nd[j];
            double px = 0.25 * p[0] * (1 - xi) * (1 - eta) + 0.25 * p[2] * (1 + xi) * (1 - eta)
                      + 0.25 * p[4] * (1 + xi) * (1 + eta) + 0.25 * p[6] * (1 - xi) * (1 + eta);
            double py = 0.25 * p[1] * (1 - xi) * (1 - eta) + 0.25 * p[3] * (1 + xi) * (1 - eta)
                      + 0.25 * p[5] * (1 + xi) * (1 + eta) + 0.25 * p[7] * (1 - xi) * (1 + eta);
            double j11 = 0.25 * ((p[2] - p[0]) * (1 - eta) + (p[4] - p[6]) * (1 + eta));
            double j12 = 0.25 * ((p[3] - p[1]) * (1 - eta) + (p[5] - p[7]) * (1 + eta));
            double j21 = 0.25 * ((p[6] - p[0]) * (1 - xi) + (p[4] - p[2]) * (1 + xi));
            double j22 = 0.25 * ((p[7] - p[1]) * (1 - xi) + (p[5] - p[3]) * (1 + xi));
            double J = fabs(j11 * j22 - j12 * j21);
            qx[k] = px; qy[k] = py; qw[k] = wt[i] * wt[j] * J;
            k++;
        }
    }
    return k;
}

static int fan_cell_quadrature(const double p[8], int degree, double *qx, double *qy, double *qw)
{
    double bar[2];
    hho_cell_barycenter(p, bar);
    int k = 0;
    for (int i = 0; i < 4; i++) {           /* :390-399 */
        const double *p0 = &p[2 * i], *p1 = &p[2 * ((i + 1) % 4)];
        int n = hho_triangle_quadrature(p0, p1, bar, degree, qx + k, qy + k, qw + k);
        if (n < 0) return n;
        k += n;
    }
    return k;
}

int hho_cell_quadrature(const double pts[8], int quad_kind, int degree, double *qx, double *qy, double *qw)
{
    if (quad_kind == HHO_QUAD_TENSOR) return tensor_cell_quadrature(pts, degree, qx, qy, qw);
    if (quad_kind == HHO_QUAD_FAN)    return fan_cell_quadrature(pts, degree, qx, qy, qw);
    return -HHO_ERR_ARG;
}

int hho_face_quadrature(const double p0[2], const double p1[2], int degree, double *qx, double *qy, double *qw)
{
    double nd[HHO_MAX_GAUSS], wt[HHO_MAX_GAUSS];
    int n = hho_gauss_legendre(degree, nd, wt);
    if (n < 0) return n;
    double sx = p1[0] - p0[0], sy = p1[1] - p0[1];
    double meas = sqrt(sx * sx + sy * sy);
    for (int i = 0; i < n; i++) {
        double t = nd[i];
        double a = 0.5 * (1 - t), b = 0.5 * (1 + t);
        qx[i] = a * p0[0] + b * p1[0];
        qy[i] = a * p0[1] + b * p1[1];
        qw[i] = wt[i] * meas * 0.5;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* bases.hpp                                                           */
/* ------------------------------------------------------------------ */
void hho_cell_basis_eval(const double bar[2], double h, int degree, double x, double y, double *phi)
{
    double bx = (x - bar[0]) / (0.5 * h);
    double by = (y - bar[1]) / (0.5 * h);
    int pos = 0;
    for (int k = 0; k <= degree; k++)
        for (int i = 0; i <= k; i++)
            phi[pos++] = hho_iexp_pow(bx, (size_t)(k - i)) * hho_iexp_pow(by, (size_t)i);
}

void hho_cell_basis_grad(const double bar[2], double h, int degree, double x, double y,
                         double *gx, double *gy)
{
    double bx = (x - bar[0]) / (0.5 * h);
    double by = (y - bar[1]) / (0.5 * h);
    double ih = 2.0 / h;
    int pos = 0;
    for (int k = 0; k <= degree; k++)
        for (int i = 0; i <= k; i++) {
            size_t pow_x = (size_t)(k - i), pow_y = (size_t)i;
            double px = hho_iexp_pow(bx, pow_x);
            double py = hho_iexp_pow(by, pow_y);
            double dx = (pow_x == 0) ? 0 : pow_x * ih * hho_iexp_pow(bx, pow_x - 1);
            double dy = (pow_y == 0) ? 0 : pow_y * ih * hho_iexp_pow(by, pow_y - 1);
            gx[pos] = dx * py;
            gy[pos] = px * dy;
            pos++;
        }
}

void hho_face_basis_eval(const double p0[2], const double p1[2], int degree, double x, double y, double *phi)
{
    double barx = (p0[0] + p1[0]) / 2.0, bary = (p0[1] + p1[1]) / 2.0;   /* basic_geom.hpp:280-286 */
    double dx = p1[0] - p0[0], dy = p1[1] - p0[1];
    double fh = sqrt(dx * dx + dy * dy);                                   /* basic_geom.hpp:307-315 */
    double basex = barx - p0[0], basey = bary - p0[1];                     /* bases.hpp:260-261 */
    double tx = x - barx, ty = y - bary;
    double dot = basex * tx + basey * ty;
    double ep = 4.0 * dot / (fh * fh);
    for (int i = 0; i <= degree; i++) phi[i] = hho_iexp_pow(ep, (size_t)i);
}

/* ------------------------------------------------------------------ */
/* Eigen LLT semantics: unpivoted lower Cholesky, forward + backward   */
/* ------------------------------------------------------------------ */
int hho_llt_factor(double *A, int n)
{
    int bad = 0;
    for (int j = 0; j < n; j++) {
        double d = A[IDX(j, j, n)];
        for (int k = 0; k < j; k++) d -= A[IDX(j, k, n)] * A[IDX(j, k, n)];
        if (!(d > 0.0) && !bad) bad = j + 1;
        d = sqrt(d);
        A[IDX(j, j, n)] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[IDX(i, j, n)];
            for (int k = 0; k < j; k++) s -= A[IDX(i, k, n)] * A[IDX(j, k, n)];
            A[IDX(i, j, n)] = s / d;
        }
    }
    for (int j = 1; j < n; j++)
        for (int i = 0; i < j; i++) A[IDX(i, j, n)] = 0.0;
    return bad;
}

void hho_llt_solve_inplace(const double *L, int n, double *B, int nrhs)
{
    for (int c = 0; c < nrhs; c++) {
        double *b = B + (size_t)c * n;
        for (int i = 0; i < n; i++) {
            double s = b[i];
            for (int k = 0; k < i; k++) s -= L[IDX(i, k, n)] * b[k];
            b[i] = s / L[IDX(i, i, n)];
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = b[i];
            for (int k = i + 1; k < n; k++) s -= L[IDX(k, i, n)] * b[k];
            b[i] = s / L[IDX(i, i, n)];
        }
    }
}

/* ------------------------------------------------------------------ */
/* hho.hpp:32-96                                                       */
/* ------------------------------------------------------------------ */
int hho_make_laplacian(const double pts[8], const uint64_t ids[4], hho_degrees di, int quad_kind,
                       double *oper, double *data)
{
    int recdeg = di.rec_deg, celdeg = di.cell_deg, facdeg = di.face_deg;
    if (recdeg > HHO_MAX_RECDEG || recdeg < 1 || celdeg < 0 || facdeg < 0) return HHO_ERR_DEGREE;
    int rbs = hho_cell_basis_size(recdeg), cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs, nr = rbs - 1;

    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);

    double stiff[HHO_MAX_RBS * HHO_MAX_RBS];
    memset(stiff, 0, sizeof(double) * (size_t)rbs * rbs);
    double gr_lhs[HHO_MAX_RBS * HHO_MAX_RBS];
    double gr_rhs[HHO_MAX_RBS * HHO_MAX_MSIZE];
    memset(gr_rhs, 0, sizeof(double) * (size_t)nr * msize);

    double qx[HHO_MAX_CELL_QPS], qy[HHO_MAX_CELL_QPS], qw[HHO_MAX_CELL_QPS];
    int nq = hho_cell_quadrature(pts, quad_kind, 2 * recdeg, qx, qy, qw);
    if (nq < 0) return -nq;

    double gx[HHO_MAX_RBS], gy[HHO_MAX_RBS], phi[HHO_MAX_RBS], fphi[HHO_MAX_FBS];
    for (int q = 0; q < nq; q++) {                       /* :57-61 */
        hho_cell_basis_grad(bar, h, recdeg, qx[q], qy[q], gx, gy);
        for (int j = 0; j < rbs; j++)
            for (int i = 0; i < rbs; i++)
                stiff[IDX(i, j, rbs)] += (qw[q] * gx[i]) * gx[j] + (qw[q] * gy[i]) * gy[j];
    }
    for (int j = 0; j < nr; j++)                         /* :63 */
        for (int i = 0; i < nr; i++) gr_lhs[IDX(i, j, nr)] = stiff[IDX(i + 1, j + 1, rbs)];
    for (int j = 0; j < cbs; j++)                        /* :64 */
        for (int i = 0; i < nr; i++) gr_rhs[IDX(i, j, nr)] = stiff[IDX(i + 1, j, rbs)];

    double nrm[8]; hho_cell_normals(pts, nrm);
    for (int f = 0; f < 4; f++) {                        /* :68-85 */
        double fp0[2], fp1[2];
        hho_cell_face_points(pts, ids, f, fp0, fp1);
        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
        int nfq = hho_face_quadrature(fp0, fp1, 2 * facdeg, fx, fy, fw);
        if (nfq < 0) return -nfq;
        for (int q = 0; q < nfq; q++) {
            hho_cell_basis_eval(bar, h, recdeg, fx[q], fy[q], phi);
            hho_cell_basis_grad(bar, h, recdeg, fx[q], fy[q], gx, gy);
            hho_face_basis_eval(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            for (int i = 0; i < nr; i++) {
                double dn = gx[i + 1] * nrm[2 * f] + gy[i + 1] * nrm[2 * f + 1];
                double wdn = fw[q] * dn;
                for (int j = 0; j < fbs; j++) gr_rhs[IDX(i, cbs + f * fbs + j, nr)] += wdn * fphi[j];
                for (int j = 0; j < cbs; j++) gr_rhs[IDX(i, j, nr)] -= wdn * phi[j];
            }
        }
    }

    int bad = hho_llt_factor(gr_lhs, nr);                /* :92 */
    memcpy(oper, gr_rhs, sizeof(double) * (size_t)nr * msize);
    hho_llt_solve_inplace(gr_lhs, nr, oper, msize);
    for (int j = 0; j < msize; j++)                      /* :93 data = gr_rhs^T oper */
        for (int i = 0; i < msize; i++) {
            double s = 0.0;
            for (int k = 0; k < nr; k++) s += gr_rhs[IDX(k, i, nr)] * oper[IDX(k, j, nr)];
            data[IDX(i, j, msize)] = s;
        }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

/* ------------------------------------------------------------------ */
/* hho.hpp:99-148                                                      */
/* ------------------------------------------------------------------ */
int hho_make_naive_stabilization(const double pts[8], const uint64_t ids[4], hho_degrees di, double *data)
{
    int celdeg = di.cell_deg, facdeg = di.face_deg;
    if (di.rec_deg > HHO_MAX_RECDEG || celdeg > HHO_MAX_RECDEG) return HHO_ERR_DEGREE;
    int cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs;
    memset(data, 0, sizeof(double) * (size_t)msize * msize);

    double bar[2]; hho_cell_barycenter(pts, bar);
    double hT = hho_cell_diameter(pts);
    double h = hho_cell_measure(pts);                    /* :119 -- the cell AREA */
    int bad = 0;

    for (int f = 0; f < 4; f++) {
        double fp0[2], fp1[2];
        hho_cell_face_points(pts, ids, f, fp0, fp1);
        double oper[HHO_MAX_FBS * HHO_MAX_MSIZE];
        double mass[HHO_MAX_FBS * HHO_MAX_FBS], L[HHO_MAX_FBS * HHO_MAX_FBS];
        double trace[HHO_MAX_FBS * HHO_MAX_RBS];
        memset(oper, 0, sizeof(double) * (size_t)fbs * msize);
        memset(mass, 0, sizeof(double) * (size_t)fbs * fbs);
        memset(trace, 0, sizeof(double) * (size_t)fbs * cbs);
        for (int i = 0; i < fbs; i++) oper[IDX(i, cbs + f * fbs + i, fbs)] = -1.0;   /* :130 */

        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
        int nfq = hho_face_quadrature(fp0, fp1, 2 * facdeg, fx, fy, fw);
        if (nfq < 0) return -nfq;
        double cphi[HHO_MAX_RBS], fphi[HHO_MAX_FBS];
        for (int q = 0; q < nfq; q++) {                  /* :133-140 */
            hho_cell_basis_eval(bar, hT, celdeg, fx[q], fy[q], cphi);
            hho_face_basis_eval(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            for (int j = 0; j < fbs; j++)
                for (int i = 0; i < fbs; i++) mass[IDX(i, j, fbs)] += (fw[q] * fphi[i]) * fphi[j];
            for (int j = 0; j < cbs; j++)
                for (int i = 0; i < fbs; i++) trace[IDX(i, j, fbs)] += (fw[q] * fphi[i]) * cphi[j];
        }
        memcpy(L, mass, sizeof(double) * (size_t)fbs * fbs);
        if (hho_llt_factor(L, fbs)) bad = 1;
        hho_llt_solve_inplace(L, fbs, trace, cbs);       /* :142 */
        memcpy(oper, trace, sizeof(double) * (size_t)fbs * cbs);

        /* :144  data += ((oper^T * mass) * oper) * (1/h), left to right as Eigen evaluates it */
        double otm[HHO_MAX_MSIZE * HHO_MAX_FBS];
        for (int k = 0; k < fbs; k++)
            for (int i = 0; i < msize; i++) {
                double s = 0.0;
                for (int l = 0; l < fbs; l++) s += oper[IDX(l, i, fbs)] * mass[IDX(l, k, fbs)];
                otm[IDX(i, k, msize)] = s;
            }
        for (int j = 0; j < msize; j++)
            for (int i = 0; i < msize; i++) {
                double s = 0.0;
                for (int k = 0; k < fbs; k++) s += otm[IDX(i, k, msize)] * oper[IDX(k, j, fbs)];
                data[IDX(i, j, msize)] += s * (1. / h);
            }
    }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

/* ------------------------------------------------------------------ */
/* hho.hpp:155-237                                                     */
/* ------------------------------------------------------------------ */
int hho_make_fancy_stabilization(const double pts[8], const uint64_t ids[4], hho_degrees di,
                                 int quad_kind, const double *R, double *data)
{
    int recdeg = di.rec_deg, celdeg = di.cell_deg, facdeg = di.face_deg;
    if (recdeg > HHO_MAX_RECDEG || recdeg < 1) return HHO_ERR_DEGREE;
    int rbs = hho_cell_basis_size(recdeg), cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs, nr = rbs - 1;
    int bad = 0;

    double bar[2]; hho_cell_barycenter(pts, bar);
    double hT = hho_cell_diameter(pts);

    double mass[HHO_MAX_RBS * HHO_MAX_RBS];
    memset(mass, 0, sizeof(double) * (size_t)rbs * rbs);
    double qx[HHO_MAX_CELL_QPS], qy[HHO_MAX_CELL_QPS], qw[HHO_MAX_CELL_QPS];
    int nq = hho_cell_quadrature(pts, quad_kind, 2 * recdeg, qx, qy, qw);
    if (nq < 0) return -nq;
    double phi[HHO_MAX_RBS], fphi[HHO_MAX_FBS];
    for (int q = 0; q < nq; q++) {                       /* :175-179 */
        hho_cell_basis_eval(bar, hT, recdeg, qx[q], qy[q], phi);
        for (int j = 0; j < rbs; j++)
            for (int i = 0; i < rbs; i++) mass[IDX(i, j, rbs)] += (qw[q] * phi[i]) * phi[j];
    }

    /* :184-190  proj1 = -M1^{-1} (M2 R) ; proj1[:, :cbs] += I */
    double M1[HHO_MAX_RBS * HHO_MAX_RBS];
    for (int j = 0; j < cbs; j++)
        for (int i = 0; i < cbs; i++) M1[IDX(i, j, cbs)] = mass[IDX(i, j, rbs)];
    double proj1[HHO_MAX_RBS * HHO_MAX_MSIZE];
    for (int j = 0; j < msize; j++)
        for (int i = 0; i < cbs; i++) {
            double s = 0.0;
            for (int k = 0; k < nr; k++) s += mass[IDX(i, 1 + k, rbs)] * R[IDX(k, j, nr)];
            proj1[IDX(i, j, cbs)] = s;
        }
    if (hho_llt_factor(M1, cbs)) bad = 1;
    hho_llt_solve_inplace(M1, cbs, proj1, msize);
    for (size_t t = 0; t < (size_t)cbs * msize; t++) proj1[t] = -proj1[t];
    for (int i = 0; i < cbs; i++) proj1[IDX(i, i, cbs)] += 1.0;

    memset(data, 0, sizeof(double) * (size_t)msize * msize);

    for (int f = 0; f < 4; f++) {                        /* :199-234 */
        double h = hT;                                   /* :201 -- the cell DIAMETER */
        double fp0[2], fp1[2];
        hho_cell_face_points(pts, ids, f, fp0, fp1);
        double fmass[HHO_MAX_FBS * HHO_MAX_FBS], L[HHO_MAX_FBS * HHO_MAX_FBS];
        double ftrace[HHO_MAX_FBS * HHO_MAX_RBS];
        memset(fmass, 0, sizeof(double) * (size_t)fbs * fbs);
        memset(ftrace, 0, sizeof(double) * (size_t)fbs * rbs);
        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
        int nfq = hho_face_quadrature(fp0, fp1, 2 * facdeg, fx, fy, fw);
        if (nfq < 0) return -nfq;
        for (int q = 0; q < nfq; q++) {                  /* :209-216 */
            hho_face_basis_eval(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            hho_cell_basis_eval(bar, hT, recdeg, fx[q], fy[q], phi);
            for (int j = 0; j < fbs; j++)
                for (int i = 0; i < fbs; i++) fmass[IDX(i, j, fbs)] += (fw[q] * fphi[i]) * fphi[j];
            for (int j = 0; j < rbs; j++)
                for (int i = 0; i < fbs; i++) ftrace[IDX(i, j, fbs)] += (fw[q] * fphi[i]) * phi[j];
        }
        memcpy(L, fmass, sizeof(double) * (size_t)fbs * fbs);
        if (hho_llt_factor(L, fbs)) bad = 1;

        /* :222-226  proj2 = M_F^{-1} (MR1 R) ; proj2[:, face block] -= I */
        double proj2[HHO_MAX_FBS * HHO_MAX_MSIZE], proj3[HHO_MAX_FBS * HHO_MAX_MSIZE];
        for (int j = 0; j < msize; j++)
            for (int i = 0; i < fbs; i++) {
                double s = 0.0;
                for (int k = 0; k < nr; k++) s += ftrace[IDX(i, 1 + k, fbs)] * R[IDX(k, j, nr)];
                proj2[IDX(i, j, fbs)] = s;
            }
        hho_llt_solve_inplace(L, fbs, proj2, msize);
        for (int i = 0; i < fbs; i++) proj2[IDX(i, cbs + f * fbs + i, fbs)] -= 1.0;
        /* :229-230  proj3 = M_F^{-1} (MR2 proj1) */
        for (int j = 0; j < msize; j++)
            for (int i = 0; i < fbs; i++) {
                double s = 0.0;
                for (int k = 0; k < cbs; k++) s += ftrace[IDX(i, k, fbs)] * proj1[IDX(k, j, cbs)];
                proj3[IDX(i, j, fbs)] = s;
            }
        hho_llt_solve_inplace(L, fbs, proj3, msize);
        double BRF[HHO_MAX_FBS * HHO_MAX_MSIZE], BtM[HHO_MAX_MSIZE * HHO_MAX_FBS];
        for (size_t t = 0; t < (size_t)fbs * msize; t++) BRF[t] = proj2[t] + proj3[t];
        /* :233  data += ((BRF^T * M_F) * BRF) / h, left to right as Eigen evaluates it */
        for (int k = 0; k < fbs; k++)
            for (int i = 0; i < msize; i++) {
                double s = 0.0;
                for (int l = 0; l < fbs; l++) s += BRF[IDX(l, i, fbs)] * fmass[IDX(l, k, fbs)];
                BtM[IDX(i, k, msize)] = s;
            }
        for (int j = 0; j < msize; j++)
            for (int i = 0; i < msize; i++) {
                double s = 0.0;
                for (int k = 0; k < fbs; k++) s += BtM[IDX(i, k, msize)] * BRF[IDX(k, j, fbs)];
                data[IDX(i, j, msize)] += s / h;
            }
    }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

/* ------------------------------------------------------------------ */
/* driver lambdas                                                      */
/* ------------------------------------------------------------------ */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
static double fn_sin_sin_rhs(double x, double y, void *u) { (void)u; return 2.0 * M_PI * M_PI * sin(M_PI * x) * sin(M_PI * y); }
static double fn_sin_sin_sol(double x, double y, void *u) { (void)u; return sin(M_PI * x) * sin(M_PI * y); }
static double fn_obstacle_rhs(double x, double y, void *u)
{
    (void)u;
    double r0 = 0.7, r = sqrt(x * x + y * y);
    if (r > r0) return -16 * r * r + 8 * r0 * r0;
    return -8.0 * (r0 * r0 * (r0 * r0 + 1)) + 8 * r0 * r0 * r * r;
}
static double fn_obstacle_sol(double x, double y, void *u)
{
    (void)u;
    double r0 = 0.7, r = sqrt(x * x + y * y);
    double s = r * r - r0 * r0, t = s > 0.0 ? s : 0.0;
    return t * t;
}
static double fn_one(double x, double y, void *u) { (void)x; (void)y; (void)u; return 1.0; }

hho_scalar_fn hho_builtin_fn(int id)
{
    switch (id) {
    case 1: return fn_sin_sin_rhs;
    case 2: return fn_sin_sin_sol;
    case 3: return fn_obstacle_rhs;
    case 4: return fn_obstacle_sol;
    case 5: return fn_one;
    default: return NULL;
    }
}

/* ------------------------------------------------------------------ */
/* utils.hpp:113-227                                                   */
/* ------------------------------------------------------------------ */
int hho_cell_mass_matrix(const double pts[8], int quad_kind, int degree, int di, double *mass)
{
    if (degree > HHO_MAX_RECDEG) return HHO_ERR_DEGREE;
    int cbs = hho_cell_basis_size(degree);
    memset(mass, 0, sizeof(double) * (size_t)cbs * cbs);
    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);
    double qx[HHO_MAX_CELL_QPS], qy[HHO_MAX_CELL_QPS], qw[HHO_MAX_CELL_QPS], phi[HHO_MAX_RBS];
    int nq = hho_cell_quadrature(pts, quad_kind, 2 * (degree + di), qx, qy, qw);
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        hho_cell_basis_eval(bar, h, degree, qx[q], qy[q], phi);
        for (int j = 0; j < cbs; j++)
            for (int i = 0; i < cbs; i++) mass[IDX(i, j, cbs)] += (qw[q] * phi[i]) * phi[j];
    }
    return HHO_OK;
}

int hho_face_mass_matrix(const double p0[2], const double p1[2], int degree, int di, double *mass)
{
    if (degree >= HHO_MAX_FBS) return HHO_ERR_DEGREE;
    int fbs = hho_face_basis_size(degree);
    memset(mass, 0, sizeof(double) * (size_t)fbs * fbs);
    double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS], phi[HHO_MAX_FBS];
    int n = hho_face_quadrature(p0, p1, 2 * (degree + di), fx, fy, fw);
    if (n < 0) return -n;
    for (int q = 0; q < n; q++) {
        hho_face_basis_eval(p0, p1, degree, fx[q], fy[q], phi);
        for (int j = 0; j < fbs; j++)
            for (int i = 0; i < fbs; i++) mass[IDX(i, j, fbs)] += (fw[q] * phi[i]) * phi[j];
    }
    return HHO_OK;
}

int hho_cell_rhs(const double pts[8], int quad_kind, int degree, int di, hho_scalar_fn f, void *user, double *rhs)
{
    if (degree > HHO_MAX_RECDEG) return HHO_ERR_DEGREE;
    int cbs = hho_cell_basis_size(degree);
    memset(rhs, 0, sizeof(double) * (size_t)cbs);
    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);
    double qx[HHO_MAX_CELL_QPS], qy[HHO_MAX_CELL_QPS], qw[HHO_MAX_CELL_QPS], phi[HHO_MAX_RBS];
    int nq = hho_cell_quadrature(pts, quad_kind, 2 * (degree + di), qx, qy, qw);
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        hho_cell_basis_eval(bar, h, degree, qx[q], qy[q], phi);
        double fv = f(qx[q], qy[q], user);
        for (int i = 0; i < cbs; i++) rhs[i] += (qw[q] * phi[i]) * fv;
    }
    return HHO_OK;
}

int hho_face_rhs(const double p0[2], const double p1[2], int degree, int di, hho_scalar_fn f, void *user, double *rhs)
{
    if (degree >= HHO_MAX_FBS) return HHO_ERR_DEGREE;
    int fbs = hho_face_basis_size(degree);
    memset(rhs, 0, sizeof(double) * (size_t)fbs);
    double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS], phi[HHO_MAX_FBS];
    int n = hho_face_quadrature(p0, p1, 2 * (degree + di), fx, fy, fw);
    if (n < 0) return -n;
    for (int q = 0; q < n; q++) {
        hho_face_basis_eval(p0, p1, degree, fx[q], fy[q], phi);
        double fv = f(fx[q], fy[q], user);
        for (int i = 0; i < fbs; i++) rhs[i] += (fw[q] * phi[i]) * fv;
    }
    return HHO_OK;
}

int hho_project_function(const double pts[8], const uint64_t ids[4], hho_degrees hdi, int quad_kind,
                         hho_scalar_fn f, void *user, int di, double *out)
{
    int cbs = hho_cell_basis_size(hdi.cell_deg), fbs = hho_face_basis_size(hdi.face_deg);
    double mm[HHO_MAX_RBS * HHO_MAX_RBS];
    int st = hho_cell_mass_matrix(pts, quad_kind, hdi.cell_deg, di, mm);
    if (st) return st;
    st = hho_cell_rhs(pts, quad_kind, hdi.cell_deg, di, f, user, out);
    if (st) return st;
    int bad = hho_llt_factor(mm, cbs);
    hho_llt_solve_inplace(mm, cbs, out, 1);
    for (int i = 0; i < 4; i++) {
        double p0[2], p1[2], fm[HHO_MAX_FBS * HHO_MAX_FBS];
        hho_cell_face_points(pts, ids, i, p0, p1);
        st = hho_face_mass_matrix(p0, p1, hdi.face_deg, di, fm);
        if (st) return st;
        st = hho_face_rhs(p0, p1, hdi.face_deg, di, f, user, out + cbs + i * fbs);
        if (st) return st;
        if (hho_llt_factor(fm, fbs)) bad = 1;
        hho_llt_solve_inplace(fm, fbs, out + cbs + i * fbs, 1);
    }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

/* ------------------------------------------------------------------ */
/* Static condensation (not in the reference)           */
/* ------------------------------------------------------------------ */
int hho_static_condensation(const double *lc, const double *f, int cbs, int nf,
                            double *S, double *g, double *rec)
{
    int ms = cbs + nf;
    double ATT[HHO_MAX_RBS * HHO_MAX_RBS];
    for (int j = 0; j < cbs; j++)
        for (int i = 0; i < cbs; i++) ATT[IDX(i, j, cbs)] = lc[IDX(i, j, ms)];
    /* rec = A_TT^{-1} [ f_T | -A_TF ] */
    for (int i = 0; i < cbs; i++) rec[IDX(i, 0, cbs)] = f ? f[i] : 0.0;
    for (int j = 0; j < nf; j++)
        for (int i = 0; i < cbs; i++) rec[IDX(i, 1 + j, cbs)] = -lc[IDX(i, cbs + j, ms)];
    int bad = hho_llt_factor(ATT, cbs);
    hho_llt_solve_inplace(ATT, cbs, rec, nf + 1);
    /* S = A_FF + A_FT rec[:,1:] ; g = -A_FT rec[:,0] */
    for (int j = 0; j < nf; j++)
        for (int i = 0; i < nf; i++) {
            double s = lc[IDX(cbs + i, cbs + j, ms)];
            for (int k = 0; k < cbs; k++) s += lc[IDX(cbs + i, k, ms)] * rec[IDX(k, 1 + j, cbs)];
            S[IDX(i, j, nf)] = s;
        }
    for (int i = 0; i < nf; i++) {
        double s = 0.0;
        for (int k = 0; k < cbs; k++) s -= lc[IDX(cbs + i, k, ms)] * rec[IDX(k, 0, cbs)];
        g[i] = s;
    }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

/* ------------------------------------------------------------------ */
/* basic_mesh.hpp:230-298                                              */
/* ------------------------------------------------------------------ */
size_t hho_mesh_num_points(const hho_mesh_params *p) { return (p->Nx + 1) * (p->Ny + 1); }
size_t hho_mesh_num_cells(const hho_mesh_params *p)  { return p->Nx * p->Ny; }
size_t hho_mesh_num_faces(const hho_mesh_params *p)  { return p->Nx * (p->Ny + 1) + p->Ny * (p->Nx + 1); }

void hho_mesh_generate(const hho_mesh_params *p, double *points, uint64_t *cell_ptids)
{
    double hx = (p->max_x - p->min_x) / p->Nx;           /* :190-196 */
    double hy = (p->max_y - p->min_y) / p->Ny;
    size_t k = 0;
    for (size_t j = 0; j < p->Ny + 1; j++)
        for (size_t i = 0; i < p->Nx + 1; i++) {
            points[2 * k]     = p->min_x + i * hx;
            points[2 * k + 1] = p->min_y + j * hy;
            k++;
        }
    /* cells are pushed row-major and then sorted lexicographically by ptids (:289); since
     * ptids[0] is strictly increasing in push order the sort is the identity.              */
    k = 0;
    for (size_t j = 0; j < p->Ny; j++)
        for (size_t i = 0; i < p->Nx; i++) {
            uint64_t p0 = j * (p->Nx + 1) + i;
            cell_ptids[4 * k + 0] = p0;
            cell_ptids[4 * k + 1] = p0 + 1;
            cell_ptids[4 * k + 2] = p0 + p->Nx + 2;
            cell_ptids[4 * k + 3] = p0 + p->Nx + 1;
            k++;
        }
}

typedef struct { uint64_t lo, hi; uint8_t bnd; } face_rec;
static int face_cmp(const void *a, const void *b)
{
    const face_rec *x = (const face_rec *)a, *y = (const face_rec *)b;
    if (x->lo != y->lo) return x->lo < y->lo ? -1 : 1;
    if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
    return 0;
}

void hho_mesh_generate_faces(const hho_mesh_params *p, uint64_t *faces, uint8_t *is_boundary)
{
    size_t nc = p->Nx * p->Ny;
    face_rec *all = (face_rec *)malloc(sizeof(face_rec) * 4 * nc);
    size_t k = 0;
    for (size_t j = 0; j < p->Ny; j++)
        for (size_t i = 0; i < p->Nx; i++) {             /* :253-286 */
            uint64_t p0 = j * (p->Nx + 1) + i, p1 = p0 + 1, p2 = p0 + p->Nx + 2, p3 = p0 + p->Nx + 1;
            face_rec f0 = { p0, p1, (uint8_t)(j == 0) };
            face_rec f1 = { p1, p2, (uint8_t)(i == p->Nx - 1) };
            face_rec f2 = { p3, p2, (uint8_t)(j == p->Ny - 1) };
            face_rec f3 = { p0, p3, (uint8_t)(i == 0) };
            all[k++] = f0; all[k++] = f1; all[k++] = f2; all[k++] = f3;
        }
    qsort(all, k, sizeof(face_rec), face_cmp);           /* :290 */
    size_t n = 0;
    for (size_t i = 0; i < k; i++) {                     /* :291 unique keeps the first of a run */
        if (i > 0 && all[i].lo == all[i - 1].lo && all[i].hi == all[i - 1].hi) continue;
        faces[2 * n] = all[i].lo; faces[2 * n + 1] = all[i].hi; is_boundary[n] = all[i].bnd; n++;
    }
    free(all);
}

size_t hho_mesh_face_id(const hho_mesh_params *p, size_t i, size_t j, int lf)
{
    size_t Nx = p->Nx, Ny = p->Ny;
    /* faces sorted by (lo,hi): every point of row jj<Ny owns a horizontal (lo,lo+1) then a
     * vertical (lo,lo+Nx+1) face, except the last point of the row (vertical only); the top
     * row owns horizontals only.                                                          */
    size_t row = 2 * Nx + 1;
    switch (lf) {
    case 0: /* bottom: horizontal at (i,j) */
        return (j < Ny) ? j * row + 2 * i : Ny * row + i;
    case 1: /* right: vertical at (i+1,j) */
        return j * row + ((i + 1 < Nx) ? 2 * (i + 1) + 1 : 2 * Nx);
    case 2: /* top: horizontal at (i,j+1) */
        return (j + 1 < Ny) ? (j + 1) * row + 2 * i : Ny * row + i;
    default: /* left: vertical at (i,j) */
        return j * row + ((i < Nx) ? 2 * i + 1 : 2 * Nx);
    }
}

int hho_mesh_face_is_boundary(const hho_mesh_params *p, size_t i, size_t j, int lf)
{
    switch (lf) {
    case 0: return j == 0;
    case 1: return i == p->Nx - 1;
    case 2: return j == p->Ny - 1;
    default: return i == 0;
    }
}

/* ------------------------------------------------------------------ */
/* assembler<Mesh>  hho.hpp:252-463                                    */
/* ------------------------------------------------------------------ */
size_t hho_assembler_compress_table(const uint8_t *is_dirichlet, size_t nfaces, int64_t *compress)
{
    size_t co = 0;                                           /* hho.hpp:313-323 */
    for (size_t i = 0; i < nfaces; i++) {
        if (!is_dirichlet[i]) compress[i] = (int64_t)co++;
        else compress[i] = -1;
    }
    return co;
}

size_t hho_assembler_system_size(hho_degrees di, size_t ncells, size_t num_other_faces)
{
    return (size_t)hho_cell_basis_size(di.cell_deg) * ncells + (size_t)hho_face_basis_size(di.face_deg) * num_other_faces;
}

int hho_dirichlet_face_data(const double p0[2], const double p1[2], int facdeg, hho_scalar_fn bf, void *user, double *out)
{
    double mass[HHO_MAX_FBS * HHO_MAX_FBS];
    int fbs = hho_face_basis_size(facdeg);
    int st = hho_face_mass_matrix(p0, p1, facdeg, 0, mass);   /* hho.hpp:383 */
    if (st) return st;
    st = hho_face_rhs(p0, p1, facdeg, 0, bf, user, out);      /* hho.hpp:384 */
    if (st) return st;
    int bad = hho_llt_factor(mass, fbs);
    hho_llt_solve_inplace(mass, fbs, out, 1);                 /* hho.hpp:385 */
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

int hho_assembler_assemble_cell(hho_degrees di, size_t cell_offset, size_t ncells,
                                const uint64_t face_ids[4], const uint8_t face_dirichlet[4],
                                const int64_t *compress, const double *lhs, const double *rhs,
                                const double *dirichlet_data,
                                int32_t *trip_rows, int32_t *trip_cols, double *trip_vals, size_t *ntrip,
                                int64_t *rhs_rows, double *rhs_vals)
{
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg);
    int msize = cbs + 4 * fbs;
    int64_t idx[HHO_MAX_MSIZE];
    int assem[HHO_MAX_MSIZE];
    size_t cell_LHS_offset = cell_offset * (size_t)cbs;               /* :362-363 */
    for (int i = 0; i < cbs; i++) { idx[i] = (int64_t)(cell_LHS_offset + i); assem[i] = 1; }   /* :365-366 */
    for (int f = 0; f < 4; f++) {                                      /* :370-387 */
        int dirichlet = face_dirichlet[f];
        int64_t face_LHS_offset = (int64_t)((size_t)cbs * ncells) + (dirichlet ? 0 : compress[face_ids[f]]) * fbs;   /* :374 */
        for (int i = 0; i < fbs; i++) { idx[cbs + f * fbs + i] = face_LHS_offset + i; assem[cbs + f * fbs + i] = !dirichlet; }
    }
    size_t nt = 0;
    for (int i = 0; i < msize; i++) {
        rhs_rows[i] = assem[i] ? idx[i] : -1;
        rhs_vals[i] = 0.0;
    }
    for (int i = 0; i < msize; i++) {                                  /* :391-403 */
        if (!assem[i]) continue;
        for (int j = 0; j < msize; j++) {
            double v = lhs[IDX(i, j, msize)];
            if (assem[j]) { trip_rows[nt] = (int32_t)idx[i]; trip_cols[nt] = (int32_t)idx[j]; trip_vals[nt] = v; nt++; }
            else rhs_vals[i] -= v * dirichlet_data[j];
        }
    }
    for (int i = 0; i < cbs; i++) rhs_vals[i] += rhs[i];              /* :405 */
    *ntrip = nt;
    return HHO_OK;
}

/* ------------------------------------------------------------------ */
/* obstacle_assembler (hho.hpp:471-751)                                 */
/* ------------------------------------------------------------------ */
void hho_obstacle_tables(const uint8_t *in_A, size_t ncells, int64_t *A_ct, int64_t *B_ct, size_t *num_I, size_t *num_A)
{
    size_t ci = 0, ca = 0;
    for (size_t i = 0; i < ncells; i++) {                              /* :538-578 */
        if (!in_A[i]) { A_ct[i] = (int64_t)ci++; B_ct[i] = -1; }
        else          { B_ct[i] = (int64_t)ca++; A_ct[i] = -1; }
    }
    *num_I = ci; *num_A = ca;
}

int hho_obstacle_assemble_cell(hho_degrees di, size_t cell_offset, size_t ncells, size_t num_I, size_t num_other_faces,
                               const uint64_t face_ids[4], const uint8_t face_dirichlet[4], const int64_t *face_ct,
                               const uint8_t *in_A, const int64_t *A_ct, const int64_t *B_ct,
                               const double *lhs, const double *rhs, const double *gamma, const double *dirichlet_data,
                               int32_t *trip_rows, int32_t *trip_cols, double *trip_vals, size_t *ntrip,
                               int64_t *rhs_rows, double *rhs_vals)
{
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg);
    int msize = cbs + 4 * fbs;
    int64_t row[HHO_MAX_MSIZE], col[HHO_MAX_MSIZE];
    int row_ok[HHO_MAX_MSIZE], col_ok[HHO_MAX_MSIZE];
    int active = in_A[cell_offset] != 0;
    int64_t cell_LHS_offset = active ? 0 : A_ct[cell_offset] * cbs;   /* :625 (the value is unused when active) */
    for (int i = 0; i < cbs; i++) {                                    /* :629-633 */
        row[i] = (int64_t)cell_offset + i; row_ok[i] = 1;
        col[i] = cell_LHS_offset + i;      col_ok[i] = !active;
    }
    for (int f = 0; f < 4; f++) {                                      /* :640-661 */
        int dirichlet = face_dirichlet[f];
        int64_t comp = dirichlet ? 0 : face_ct[face_ids[f]];
        for (int i = 0; i < fbs; i++) {
            row[cbs + f * fbs + i] = (int64_t)((size_t)cbs * ncells) + comp * fbs + i;   /* :644 */
            col[cbs + f * fbs + i] = (int64_t)((size_t)cbs * num_I) + comp * fbs + i;    /* :645 */
            row_ok[cbs + f * fbs + i] = col_ok[cbs + f * fbs + i] = !dirichlet;
        }
    }
    size_t nt = 0;
    for (int i = 0; i < msize; i++) { rhs_rows[i] = row_ok[i] ? row[i] : -1; rhs_vals[i] = 0.0; }
    for (int i = 0; i < msize; i++) {                                  /* :666-682 */
        if (!row_ok[i]) continue;
        for (int j = 0; j < msize; j++) {
            double v = lhs[IDX(i, j, msize)];
            if (col_ok[j]) { trip_rows[nt] = (int32_t)row[i]; trip_cols[nt] = (int32_t)col[j]; trip_vals[nt] = v; nt++; }
            else if (j < cbs) rhs_vals[i] -= v * gamma[cell_offset];   /* :677 */
            else rhs_vals[i] -= v * dirichlet_data[j];                 /* :679 */
        }
    }
    for (int i = 0; i < cbs; i++) rhs_vals[i] += rhs[i];              /* :686 */
    if (active) {                                                      /* :688-693 */
        trip_rows[nt] = (int32_t)(cell_offset * (size_t)cbs);
        trip_cols[nt] = (int32_t)(num_I * (size_t)cbs + num_other_faces * (size_t)fbs + (size_t)B_ct[cell_offset]);
        trip_vals[nt] = 1.0;
        nt++;
    }
    *ntrip = nt;
    return HHO_OK;
}

void hho_obstacle_expand_solution(hho_degrees di, size_t ncells, size_t nfaces, const uint8_t *face_dirichlet,
                                  const int64_t *face_ct, const uint8_t *in_A, const double *solution,
                                  const double *g, const double *gamma, double *alpha, double *beta)
{
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg);
    size_t num_I = 0, num_other = 0;
    for (size_t i = 0; i < ncells; i++) num_I += !in_A[i];
    for (size_t f = 0; f < nfaces; f++) num_other += !face_dirichlet[f];
    for (size_t i = 0; i < ncells * (size_t)cbs; i++) alpha[i] = gamma[i];                    /* :709 */
    for (size_t i = 0, co = 0; i < ncells; i++)                                               /* :710-714 */
        if (!in_A[i]) { for (int k = 0; k < cbs; k++) alpha[i * cbs + k] = solution[co * cbs + k]; co++; }
    for (size_t i = 0; i < ncells * (size_t)cbs; i++) beta[i] = 0.0;   /* :716 sizes beta by the CELL count: enough for cbs = 1 only */
    for (size_t i = 0, co = 0; i < ncells; i++)                                               /* :717-721 */
        if (in_A[i]) {
            for (int k = 0; k < cbs; k++) beta[i * cbs + k] = solution[num_I * cbs + num_other * fbs + co * cbs + k];
            co++;
        }
    for (size_t f = 0; f < nfaces; f++)                                                       /* :723-743 */
        for (int k = 0; k < fbs; k++)
            alpha[ncells * cbs + f * fbs + k] = face_dirichlet[f] ? g[f * fbs + k]
                                                                  : solution[cbs * num_I + (size_t)face_ct[f] * fbs + k];
}

void hho_obstacle_take_local_data(hho_degrees di, size_t cell_offset, size_t ncells, const uint64_t face_ids[4],
                                  const double *expanded, double *out)
{
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg);
    for (int i = 0; i < cbs; i++) out[i] = expanded[cell_offset * cbs + i];                   /* :764-770 */
    for (int f = 0; f < 4; f++)                                                               /* :772-778 */
        for (int k = 0; k < fbs; k++) out[cbs + f * fbs + k] = expanded[(size_t)cbs * ncells + face_ids[f] * fbs + k];
}

/* ------------------------------------------------------------------ */
/* Batched loop == the reference's per-cell "Matrix assembly" span     */
/* (convergence_test.cpp:202-213 without the triplet push).            */
/* ------------------------------------------------------------------ */
int hho_local_ops_batch(const double *points, const uint64_t *cell_ptids, size_t first, size_t n,
                        hho_degrees di, int quad_kind, int stab_kind,
                        hho_scalar_fn f, void *user, int rhs_di,
                        double *out_oper, double *out_data, double *out_stab, double *out_lc,
                        double *out_rhs)
{
    int rbs = hho_cell_basis_size(di.rec_deg), cbs = hho_cell_basis_size(di.cell_deg);
    int fbs = hho_face_basis_size(di.face_deg), msize = cbs + 4 * fbs, nr = rbs - 1;
    size_t mm = (size_t)msize * msize, om = (size_t)nr * msize;
    int worst = HHO_OK;
    double oper[HHO_MAX_RBS * HHO_MAX_MSIZE], data[HHO_MAX_MSIZE * HHO_MAX_MSIZE], stab[HHO_MAX_MSIZE * HHO_MAX_MSIZE];
    for (size_t c = 0; c < n; c++) {
        const uint64_t *ids = cell_ptids + 4 * (first + c);
        double pts[8];
        for (int v = 0; v < 4; v++) { pts[2 * v] = points[2 * ids[v]]; pts[2 * v + 1] = points[2 * ids[v] + 1]; }
        int st = hho_make_laplacian(pts, ids, di, quad_kind, oper, data);
        if (st && st != HHO_ERR_NOT_SPD) return st;
        if (st) worst = st;
        if (stab_kind == HHO_STAB_FANCY)      st = hho_make_fancy_stabilization(pts, ids, di, quad_kind, oper, stab);
        else if (stab_kind == HHO_STAB_NAIVE) st = hho_make_naive_stabilization(pts, ids, di, stab);
        else { memset(stab, 0, sizeof(double) * mm); st = HHO_OK; }
        if (st && st != HHO_ERR_NOT_SPD) return st;
        if (st) worst = st;
        if (out_oper) memcpy(out_oper + c * om, oper, sizeof(double) * om);
        if (out_data) memcpy(out_data + c * mm, data, sizeof(double) * mm);
        if (out_stab) memcpy(out_stab + c * mm, stab, sizeof(double) * mm);
        if (out_lc) for (size_t t = 0; t < mm; t++) out_lc[c * mm + t] = data[t] + stab[t];
        if (out_rhs && f) {
            st = hho_cell_rhs(pts, quad_kind, di.cell_deg, rhs_di, f, user, out_rhs + c * (size_t)cbs);
            if (st) return st;
        }
    }
    return worst;
}

/* ------------------------------------------------------------------ */
/* CPU baseline of bench.py        */
/* ------------------------------------------------------------------ */
#ifndef HHO_FLOPCOUNT      /* (not part of the counted restatement: oracle/flopcount.cpp) */
static double wall_seconds(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int hho_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* The per-cell loop of the reference's drivers over cells [first, first + n), `nthreads` threads over cells
 * (1 = the reference: it is single-threaded).  Same arithmetic per cell as hho_local_ops_batch. */
int hho_local_ops_batch_mt(const double *points, const uint64_t *cell_ptids, size_t first, size_t n,
                           hho_degrees di, int quad_kind, int stab_kind, hho_scalar_fn f, void *user, int rhs_di,
                           double *out_lc, double *out_rhs, int nthreads)
{
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg), msize = cbs + 4 * fbs;
    size_t mm = (size_t)msize * msize;
    int worst = HHO_OK;
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(max : worst)
#endif
    for (long long c = 0; c < (long long)n; c++) {
        int st = hho_local_ops_batch(points, cell_ptids, first + (size_t)c, 1, di, quad_kind, stab_kind, f, user, rhs_di,
                                     NULL, NULL, NULL, out_lc ? out_lc + (size_t)c * mm : NULL,
                                     out_rhs ? out_rhs + (size_t)c * cbs : NULL);
        if (st > worst) worst = st;
    }
    return worst;
}

typedef struct { int32_t col; double val; } csr_ent;
static int csr_ent_cmp(const void *a, const void *b)
{
    int32_t x = ((const csr_ent *)a)->col, y = ((const csr_ent *)b)->col;
    return x < y ? -1 : x > y;
}

/* SparseMatrix::setFromTriplets (hho.hpp:451-455): rows bucketed by a counting pass in push order, every row sorted
 * by column (stable for equal columns: qsort on a key that includes the push position) and duplicates summed.
 * Returns nnz; rowptr nrows + 1, colind / values with room for ntrip entries. */
size_t hho_set_from_triplets(size_t ntrip, const int32_t *rows, const int32_t *cols, const double *vals, size_t nrows,
                             int64_t *rowptr, int32_t *colind, double *values, int nthreads)
{
    int64_t *count = (int64_t *)calloc(nrows + 1, sizeof(int64_t));
    for (size_t t = 0; t < ntrip; t++) if (rows[t] >= 0) count[rows[t] + 1]++;
    for (size_t r = 0; r < nrows; r++) count[r + 1] += count[r];
    size_t total = (size_t)count[nrows];
    csr_ent *ent = (csr_ent *)malloc(sizeof(csr_ent) * (total ? total : 1));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (nrows + 1));
    memcpy(fill, count, sizeof(int64_t) * (nrows + 1));
    for (size_t t = 0; t < ntrip; t++) {
        if (rows[t] < 0) continue;
        int64_t pos = fill[rows[t]]++;
        ent[pos].col = cols[t]; ent[pos].val = vals[t];
    }
    int64_t *rowlen = (int64_t *)malloc(sizeof(int64_t) * (nrows + 1));
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (long long r = 0; r < (long long)nrows; r++) {
        csr_ent *e = ent + count[r];
        int64_t m = count[r + 1] - count[r], k = 0;
        /* insertion sort: stable, rows are short (<= 4 msize entries) and arrive almost sorted */
        for (int64_t i = 1; i < m; i++) {
            csr_ent x = e[i]; int64_t j = i - 1;
            while (j >= 0 && e[j].col > x.col) { e[j + 1] = e[j]; j--; }
            e[j + 1] = x;
        }
        for (int64_t i = 0; i < m; i++) {
            if (k > 0 && e[k - 1].col == e[i].col) e[k - 1].val += e[i].val;
            else e[k++] = e[i];
        }
        rowlen[r] = k;
    }
    (void)csr_ent_cmp;
    size_t nnz = 0;
    for (size_t r = 0; r < nrows; r++) {
        rowptr[r] = (int64_t)nnz;
        const csr_ent *e = ent + count[r];
        for (int64_t i = 0; i < rowlen[r]; i++) { colind[nnz] = e[i].col; values[nnz] = e[i].val; nnz++; }
    }
    rowptr[nrows] = (int64_t)nnz;
    free(count); free(ent); free(fill); free(rowlen);
    return nnz;
}

/* The span the reference labels "Matrix assembly" (apps/cuthho/cuthho_square.cpp:881-905, apps/obstacle/obstacle.cpp:145-161,
 * apps/convergence_test/convergence_test.cpp:201-217) on the generator mesh, cell rows [row_begin, row_end) of it:
 * make_assembler, then per cell make_hho_laplacian + stabilization + make_rhs + assembler.assemble, then finalize.
 * seconds[0]: the per-cell operators + rhs alone; seconds[1]: the whole span (operators recomputed).  The system has
 * the full mesh's numbering; only the sampled rows contribute.  checksum: sum of the CSR values (keeps the work live). */
int hho_matrix_assembly_timed(const hho_mesh_params *mp, size_t row_begin, size_t row_end, hho_degrees di, int quad_kind,
                              int stab_kind, int rhs_fn, int bcs_fn, int rhs_di, int nthreads, double seconds[2],
                              size_t *nnz_out, double *checksum)
{
    size_t np = hho_mesh_num_points(mp), nc = hho_mesh_num_cells(mp), nf = hho_mesh_num_faces(mp);
    if (row_end > mp->Ny || row_begin >= row_end) return HHO_ERR_DEGREE;
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg), msize = cbs + 4 * fbs;
    size_t mm = (size_t)msize * msize, n = (row_end - row_begin) * mp->Nx, first = row_begin * mp->Nx;
    double *points = (double *)malloc(sizeof(double) * 2 * np);
    uint64_t *ptids = (uint64_t *)malloc(sizeof(uint64_t) * 4 * nc);
    hho_mesh_generate(mp, points, ptids);
    hho_scalar_fn f = hho_builtin_fn(rhs_fn), bf = hho_builtin_fn(bcs_fn);
    if (nthreads < 1) nthreads = 1;
    int status = HHO_OK;

    /* (a) operators + rhs only */
    double *lc = (double *)malloc(sizeof(double) * mm * n), *rhs = (double *)malloc(sizeof(double) * (size_t)cbs * n);
    double t0 = wall_seconds();
    status = hho_local_ops_batch_mt(points, ptids, first, n, di, quad_kind, stab_kind, f, NULL, rhs_di, lc, rhs, nthreads);
    seconds[0] = wall_seconds() - t0;
    if (status && status != HHO_ERR_NOT_SPD) { free(points); free(ptids); free(lc); free(rhs); return status; }

    /* (b) the whole span.  msh.faces belongs to the mesh (built by its constructor, outside the reference's timer);
     * make_assembler builds the compress table from it (hho.hpp:298-335): O(faces) once per assembly, here charged in
     * proportion to the sampled share of the mesh */
    uint64_t *faces = (uint64_t *)malloc(sizeof(uint64_t) * 2 * nf);
    uint8_t *is_bnd = (uint8_t *)malloc(nf);
    int64_t *compress = (int64_t *)malloc(sizeof(int64_t) * nf);
    hho_mesh_generate_faces(mp, faces, is_bnd);
    t0 = wall_seconds();
    size_t num_other = hho_assembler_compress_table(is_bnd, nf, compress);
    const double t_ctor = (wall_seconds() - t0) * (double)(row_end - row_begin) / (double)mp->Ny;
    t0 = wall_seconds();
    size_t system_size = hho_assembler_system_size(di, nc, num_other);
    int32_t *tr = (int32_t *)malloc(sizeof(int32_t) * mm * n), *tc = (int32_t *)malloc(sizeof(int32_t) * mm * n);
    double *tv = (double *)malloc(sizeof(double) * mm * n);
    int64_t *rrows = (int64_t *)malloc(sizeof(int64_t) * (size_t)msize * n);
    double *rvals = (double *)malloc(sizeof(double) * (size_t)msize * n);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (long long cc = 0; cc < (long long)n; cc++) {
        size_t c = first + (size_t)cc, ci = c % mp->Nx, cj = c / mp->Nx;
        double lcl[HHO_MAX_MSIZE * HHO_MAX_MSIZE], rl[HHO_MAX_RBS], dd[HHO_MAX_MSIZE];
        hho_local_ops_batch(points, ptids, c, 1, di, quad_kind, stab_kind, f, NULL, rhs_di, NULL, NULL, NULL, lcl, rl);
        uint64_t fids[4]; uint8_t fdir[4];
        memset(dd, 0, sizeof(dd));
        for (int lf = 0; lf < 4; lf++) {
            fids[lf] = hho_mesh_face_id(mp, ci, cj, lf);
            fdir[lf] = is_bnd[fids[lf]];
            if (fdir[lf]) {                                       /* hho.hpp:381-386 */
                const double *p0 = points + 2 * faces[2 * fids[lf]], *p1 = points + 2 * faces[2 * fids[lf] + 1];
                hho_dirichlet_face_data(p0, p1, di.face_deg, bf, NULL, dd + cbs + lf * fbs);
            }
        }
        /* assemble() pushes only the assembled pairs: the unused tail of the cell's slots is marked -1 */
        size_t nt = 0;
        int32_t *r_ = tr + (size_t)cc * mm, *c_ = tc + (size_t)cc * mm;
        hho_assembler_assemble_cell(di, c, nc, fids, fdir, compress, lcl, rl, dd, r_, c_, tv + (size_t)cc * mm, &nt,
                                    rrows + (size_t)cc * msize, rvals + (size_t)cc * msize);
        for (size_t t = nt; t < mm; t++) { r_[t] = -1; c_[t] = -1; }
    }
    double *RHS = (double *)calloc(system_size ? system_size : 1, sizeof(double));
    for (size_t t = 0; t < (size_t)msize * n; t++) if (rrows[t] >= 0) RHS[rrows[t]] += rvals[t];
    /* finalize (hho.hpp:451-455) */
    int64_t *rowptr = (int64_t *)malloc(sizeof(int64_t) * (system_size + 1));
    int32_t *colind = (int32_t *)malloc(sizeof(int32_t) * mm * n);
    double *values = (double *)malloc(sizeof(double) * mm * n);
    size_t nnz = hho_set_from_triplets(mm * n, tr, tc, tv, system_size, rowptr, colind, values, nthreads);
    seconds[1] = wall_seconds() - t0 + t_ctor;
    double cs = 0.0;
    for (size_t t = 0; t < nnz; t++) cs += values[t];
    for (size_t t = 0; t < system_size; t++) cs += RHS[t];
    for (size_t t = 0; t < mm * n; t++) cs += lc[t] * 0.0;
    if (checksum) *checksum = cs;
    if (nnz_out) *nnz_out = nnz;
    free(points); free(ptids); free(lc); free(rhs); free(faces); free(is_bnd); free(compress); free(tr); free(tc); free(tv);
    free(rrows); free(rvals); free(RHS); free(rowptr); free(colind); free(values);
    return status;
}
#endif /* HHO_FLOPCOUNT */
