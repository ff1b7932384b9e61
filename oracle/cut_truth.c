/*
 * cut_truth.c -- the cut-cell operators of apps/cuthho/cuthho_square.cpp evaluated in IEEE binary128
 * (__float128, 113-bit significand) from the double-valued geometry and quadrature lists.
 *
 * TEST INFRASTRUCTURE, like the rest of oracle/.  Purpose: the Nitsche-penalised rbs x rbs system of a
 * sliver cut is badly conditioned, so two correct double-precision evaluations (Eigen's, the C oracle's,
 * the HIP kernel's) differ from each other by cond * eps.  This file gives the tests a side to judge them
 * against: the SAME formulas (cuthho_square.cpp:308-388, 390-502, 566-621, 623-666; cuthho_utils.hpp:65-84)
 * with every arithmetic operation in binary128, so that its own rounding (eps = 1e-34) is negligible even at
 * cond = 1e12.  What is shared with the double side, and therefore held fixed, are the INPUTS: the mesh
 * points, the point ids, the level set's parameters and the quadrature lists (x, y, w) that
 * cut_cell_quadrature / cut_interface_quadrature / cut_face_quadrature (cuthho_oracle.c; restating
 * cuthho_geom.hpp:798-895) produce in double -- the same lists pa_cut_quadrature_points returns for the
 * product (bit-identical: tests/test_gpu_cuthho.py).  Everything downstream of those lists -- barycenter,
 * diameter, measure, normals, basis values, level-set normal, source / boundary functions, the sums, the
 * factorization, the products -- is binary128.
 *
 * It is pinned itself by the 50-digit mpmath fixtures of tests/golden/make_golden.py (cut cells:
 * tests/golden/cut_ops.npz), an independent implementation in another language.
 */
#include <quadmath.h>
#include <string.h>
#include <stdlib.h>
#include "cuthho_oracle.h"

typedef __float128 q_t;
#define IDX(i, j, ld) ((size_t)(j) * (size_t)(ld) + (size_t)(i))
#define QMAX_RBS HHO_MAX_RBS
#define QMAX_MS HHO_MAX_MSIZE

static const q_t Q_ETA = 5.0Q;                                   /* cell_eta, cuthho_square.cpp:301-306 */
/* error attribution (tests print it): 0 = everything in binary128; 1 = gr_lhs and gr_rhs ROUNDED TO DOUBLE once formed,
 * the factorization and the products still in binary128 -- what a perfect solver would return from double inputs */
static int g_round_inputs = 0;
void cut_truth_round_inputs(int on) { g_round_inputs = on; }
/* 1-norm condition number of the pinned (2 rbs - 1) system of the last cut_truth_laplacian_interface call (tests print it) */
static double g_last_iface_cond = -1.0;
double cut_truth_last_interface_cond(void) { return g_last_iface_cond; }

static void q_cell_pts(const cut_mesh *m, size_t c, q_t pts[8], uint64_t ids[4], double dpts[8])
{
    const double *P = cut_mesh_points(m);
    const uint64_t *T = cut_mesh_cell_ptids(m);
    for (int v = 0; v < 4; v++) {
        ids[v] = T[4 * c + v];
        dpts[2 * v] = P[2 * ids[v]]; dpts[2 * v + 1] = P[2 * ids[v] + 1];
        pts[2 * v] = dpts[2 * v]; pts[2 * v + 1] = dpts[2 * v + 1];
    }
}

static void q_barycenter(const q_t pts[8], q_t bar[2])          /* basic_geom.hpp:247-278 */
{
    q_t rx = 0, ry = 0, den = 0;
    for (int i = 2; i < 4; i++) {
        q_t ax = pts[2 * (i - 1)] - pts[0], ay = pts[2 * (i - 1) + 1] - pts[1];
        q_t bx = pts[2 * i] - pts[0], by = pts[2 * i + 1] - pts[1];
        q_t d = (ax * by - ay * bx) / 2;
        rx += (ax + bx) * d; ry += (ay + by) * d; den += d;
    }
    bar[0] = pts[0] + rx / (den * 3); bar[1] = pts[1] + ry / (den * 3);
}

static q_t q_diameter(const q_t pts[8])                         /* basic_geom.hpp:288-305 */
{
    q_t diam = 0;
    for (int i = 0; i < 4; i++)
        for (int j = i + 1; j < 4; j++) {
            q_t dx = pts[2 * j] - pts[2 * i], dy = pts[2 * j + 1] - pts[2 * i + 1];
            q_t d = sqrtq(dx * dx + dy * dy);
            if (d > diam) diam = d;
        }
    return diam;
}

static q_t q_measure(const q_t pts[8])                          /* basic_geom.hpp:317-334 */
{
    q_t acc = 0;
    for (int i = 1; i < 3; i++) {
        q_t ux = pts[2 * i] - pts[0], uy = pts[2 * i + 1] - pts[1];
        q_t vx = pts[2 * (i + 1)] - pts[0], vy = pts[2 * (i + 1) + 1] - pts[1];
        acc += fabsq(ux * vy - uy * vx) / 2;
    }
    return acc;
}

static void q_normals(const q_t pts[8], q_t n[8])               /* basic_geom.hpp:349-372 */
{
    for (int i = 0; i < 4; i++) {
        int k = (i + 1) % 4;
        q_t vx = pts[2 * k] - pts[2 * i], vy = pts[2 * k + 1] - pts[2 * i + 1];
        q_t nrm = sqrtq(vx * vx + vy * vy);
        n[2 * i] = vy / nrm; n[2 * i + 1] = -vx / nrm;
        if (g_round_inputs == 2) { n[2 * i] = (double)n[2 * i]; n[2 * i + 1] = (double)n[2 * i + 1]; }
    }
}

static void q_face_points(const q_t pts[8], const uint64_t ids[4], int f, q_t a[2], q_t b[2])
{
    int i0 = f, i1 = (f + 1) % 4;
    if (ids[i0] > ids[i1]) { int t = i0; i0 = i1; i1 = t; }     /* basic_geom.hpp:202-203 */
    a[0] = pts[2 * i0]; a[1] = pts[2 * i0 + 1]; b[0] = pts[2 * i1]; b[1] = pts[2 * i1 + 1];
}

static q_t q_ipow(q_t x, int n) { q_t r = 1; for (int i = 0; i < n; i++) r *= x; return r; }

static void q_cell_basis(const q_t bar[2], q_t h, int degree, q_t x, q_t y, q_t *phi, q_t *gx, q_t *gy)
{                                                               /* bases.hpp:85-190 */
    q_t bx = (x - bar[0]) / (h / 2), by = (y - bar[1]) / (h / 2), ih = 2 / h;
    if (g_round_inputs == 2) {      /* attribution: the scaled coordinates of a point (and 2/h) rounded to double, everything else exact */
        const double db = (double)bar[0], dc = (double)bar[1], dh = (double)h;
        bx = (q_t)(((double)x - db) * (1.0 / (0.5 * dh))); by = (q_t)(((double)y - dc) * (1.0 / (0.5 * dh))); ih = (q_t)(2.0 / dh);
    }
    int pos = 0;
    for (int k = 0; k <= degree; k++)
        for (int i = 0; i <= k; i++) {
            int px = k - i, py = i;
            if (phi) phi[pos] = q_ipow(bx, px) * q_ipow(by, py);
            if (gx) {
                gx[pos] = px == 0 ? 0 : px * ih * q_ipow(bx, px - 1) * q_ipow(by, py);
                gy[pos] = py == 0 ? 0 : py * ih * q_ipow(bx, px) * q_ipow(by, py - 1);
            }
            pos++;
        }
}

static void q_face_basis(const q_t p0[2], const q_t p1[2], int degree, q_t x, q_t y, q_t *phi)
{                                                               /* bases.hpp:253-280 */
    q_t barx = (p0[0] + p1[0]) / 2, bary = (p0[1] + p1[1]) / 2;
    q_t dx = p1[0] - p0[0], dy = p1[1] - p0[1];
    q_t h2 = dx * dx + dy * dy;
    q_t ep = 4 * ((barx - p0[0]) * (x - barx) + (bary - p0[1]) * (y - bary)) / h2;
    if (g_round_inputs == 2) ep = (double)ep;
    for (int i = 0; i <= degree; i++) phi[i] = q_ipow(ep, i);
}

static void q_ls_normal(const cut_level_set *ls, q_t x, q_t y, q_t n[2])
{                                                               /* cuthho_square.cpp:56-124 */
    if (ls->kind == CUT_LS_CIRCLE) {
        q_t gx = 2 * x - 2 * (q_t)ls->alpha, gy = 2 * y - 2 * (q_t)ls->beta;
        q_t nr = sqrtq(gx * gx + gy * gy);
        n[0] = gx / nr; n[1] = gy / nr;
        if (g_round_inputs == 2) { n[0] = (double)n[0]; n[1] = (double)n[1]; }
    } else { n[0] = 0; n[1] = 1; }
}

/* the built-in functions of hho_builtin_fn (hho_oracle.c), ids 1 and 2: 2 pi^2 sin(pi x) sin(pi y), sin(pi x) sin(pi y) */
static q_t q_fn(int id, q_t x, q_t y)
{
    q_t s = sinq(M_PIq * x) * sinq(M_PIq * y);
    return id == 1 ? 2 * M_PIq * M_PIq * s : s;
}

/* dense symmetric positive definite solve in binary128 (Cholesky); returns 0 or the 1-based index of a bad pivot */
static int q_llt_factor(q_t *A, int n)
{
    for (int j = 0; j < n; j++) {
        q_t d = A[IDX(j, j, n)];
        for (int k = 0; k < j; k++) d -= A[IDX(j, k, n)] * A[IDX(j, k, n)];
        if (!(d > 0)) return j + 1;
        d = sqrtq(d);
        A[IDX(j, j, n)] = d;
        for (int i = j + 1; i < n; i++) {
            q_t s = A[IDX(i, j, n)];
            for (int k = 0; k < j; k++) s -= A[IDX(i, k, n)] * A[IDX(j, k, n)];
            A[IDX(i, j, n)] = s / d;
        }
    }
    return 0;
}

static void q_llt_solve(const q_t *L, int n, q_t *B, int nrhs)
{
    for (int c = 0; c < nrhs; c++) {
        q_t *b = B + (size_t)c * n;
        for (int i = 0; i < n; i++) {
            q_t s = b[i];
            for (int k = 0; k < i; k++) s -= L[IDX(i, k, n)] * b[k];
            b[i] = s / L[IDX(i, i, n)];
        }
        for (int i = n - 1; i >= 0; i--) {
            q_t s = b[i];
            for (int k = i + 1; k < n; k++) s -= L[IDX(k, i, n)] * b[k];
            b[i] = s / L[IDX(i, i, n)];
        }
    }
}

/* make_hho_laplacian(msh, cl, level_set, di, where) for a CUT cell, cuthho_square.cpp:308-388.
 * oper rbs x msize, data msize x msize, column-major, rounded to double at the very end. */
int cut_truth_laplacian(const cut_mesh *m, const cut_level_set *ls, size_t c, hho_degrees di, int where,
                        double *oper, double *data)
{
    if (cut_mesh_cell_location(m)[c] != CUT_ON_INTERFACE) return HHO_ERR_ARG;
    q_t pts[8]; uint64_t ids[4]; double dpts[8];
    q_cell_pts(m, c, pts, ids, dpts);
    int recdeg = di.rec_deg, celdeg = di.cell_deg, facdeg = di.face_deg;
    int rbs = hho_cell_basis_size(recdeg), cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs;
    if (recdeg > HHO_MAX_RECDEG || cbs > rbs) return HHO_ERR_DEGREE;
    q_t bar[2]; q_barycenter(pts, bar);
    q_t h = q_diameter(pts), hT = q_measure(pts);
    q_t *stiff = calloc((size_t)rbs * rbs, sizeof(q_t)), *gr_rhs = calloc((size_t)rbs * msize, sizeof(q_t));
    q_t *L = malloc(sizeof(q_t) * rbs * rbs), *op = malloc(sizeof(q_t) * rbs * msize);
    double *qx = malloc(sizeof(double) * 3 * CUT_MAX_QPS), *qy = qx + CUT_MAX_QPS, *qw = qy + CUT_MAX_QPS;
    q_t gx[QMAX_RBS], gy[QMAX_RBS], phi[QMAX_RBS], fphi[HHO_MAX_FBS];
    int st = HHO_OK;

    int nq = cut_cell_quadrature(m, c, 2 * recdeg, where, qx, qy, qw, CUT_MAX_QPS);        /* :336-341 */
    if (nq < 0) { st = -nq; goto done; }
    for (int q = 0; q < nq; q++) {
        q_cell_basis(bar, h, recdeg, qx[q], qy[q], NULL, gx, gy);
        for (int j = 0; j < rbs; j++)
            for (int i = 0; i < rbs; i++) stiff[IDX(i, j, rbs)] += (q_t)qw[q] * (gx[i] * gx[j] + gy[i] * gy[j]);
    }
    nq = cut_interface_quadrature(m, c, 2 * recdeg, where, qx, qy, qw, CUT_MAX_QPS);       /* :347-360 */
    if (nq < 0) { st = -nq; goto done; }
    for (int q = 0; q < nq; q++) {
        q_t n[2];
        q_cell_basis(bar, h, recdeg, qx[q], qy[q], phi, gx, gy);
        q_ls_normal(ls, qx[q], qy[q], n);
        for (int j = 0; j < rbs; j++) {
            q_t dnj = gx[j] * n[0] + gy[j] * n[1];
            for (int i = 0; i < rbs; i++) {
                q_t dni = gx[i] * n[0] + gy[i] * n[1];
                const q_t eta_h = g_round_inputs == 2 ? (q_t)(5.0 / (double)hT) : Q_ETA / hT;
                stiff[IDX(i, j, rbs)] += (q_t)qw[q] * (phi[i] * phi[j] * eta_h - phi[i] * dnj - dni * phi[j]);
            }
        }
    }
    memcpy(L, stiff, sizeof(q_t) * rbs * rbs);                                              /* :362 */
    for (int j = 0; j < cbs; j++)                                                           /* :363 */
        for (int i = 0; i < rbs; i++) gr_rhs[IDX(i, j, rbs)] = stiff[IDX(i, j, rbs)];
    q_t nrm[8]; q_normals(pts, nrm);
    for (int f = 0; f < 4; f++) {                                                           /* :366-383 */
        q_t fp0[2], fp1[2];
        q_face_points(pts, ids, f, fp0, fp1);
        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
        int nfq = cut_face_quadrature(m, c, f, 2 * recdeg, where, fx, fy, fw, HHO_MAX_GAUSS);
        if (nfq < 0) { st = -nfq; goto done; }
        for (int q = 0; q < nfq; q++) {
            q_cell_basis(bar, h, recdeg, fx[q], fy[q], phi, gx, gy);
            q_face_basis(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            for (int i = 0; i < rbs; i++) {
                q_t wdn = (q_t)fw[q] * (gx[i] * nrm[2 * f] + gy[i] * nrm[2 * f + 1]);
                for (int j = 0; j < fbs; j++) gr_rhs[IDX(i, cbs + f * fbs + j, rbs)] += wdn * fphi[j];
                for (int j = 0; j < cbs; j++) gr_rhs[IDX(i, j, rbs)] -= wdn * phi[j];
            }
        }
    }
    if (g_round_inputs == 1) {
        for (int i = 0; i < rbs * rbs; i++) L[i] = (double)L[i];
        for (int i = 0; i < rbs * msize; i++) gr_rhs[i] = (double)gr_rhs[i];
    }
    if (q_llt_factor(L, rbs)) { st = HHO_ERR_NOT_SPD; goto done; }                           /* :385 */
    memcpy(op, gr_rhs, sizeof(q_t) * rbs * msize);
    q_llt_solve(L, rbs, op, msize);
    for (int i = 0; i < rbs * msize; i++) oper[i] = (double)op[i];
    for (int j = 0; j < msize; j++)                                                          /* :386 */
        for (int i = 0; i < msize; i++) {
            q_t s = 0;
            for (int k = 0; k < rbs; k++) s += gr_rhs[IDX(k, i, rbs)] * op[IDX(k, j, rbs)];
            data[IDX(i, j, msize)] = (double)s;
        }
done:
    free(stiff); free(gr_rhs); free(L); free(op); free(qx);
    return st;
}

/* make_hho_cut_stabilization for a CUT cell, cuthho_square.cpp:566-621 */
int cut_truth_stabilization(const cut_mesh *m, size_t c, hho_degrees di, int where, double *stab)
{
    if (cut_mesh_cell_location(m)[c] != CUT_ON_INTERFACE) return HHO_ERR_ARG;
    q_t pts[8]; uint64_t ids[4]; double dpts[8];
    q_cell_pts(m, c, pts, ids, dpts);
    int celdeg = di.cell_deg, facdeg = di.face_deg;
    int cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg), msize = cbs + 4 * fbs;
    q_t bar[2]; q_barycenter(pts, bar);
    q_t hd = q_diameter(pts), hT = q_measure(pts);                                           /* :589 */
    q_t *acc = calloc((size_t)msize * msize, sizeof(q_t));
    int st = HHO_OK;
    for (int f = 0; f < 4; f++) {
        q_t fp0[2], fp1[2];
        q_face_points(pts, ids, f, fp0, fp1);
        q_t op[HHO_MAX_FBS * QMAX_MS], mass[HHO_MAX_FBS * HHO_MAX_FBS], L[HHO_MAX_FBS * HHO_MAX_FBS], trace[HHO_MAX_FBS * QMAX_RBS];
        memset(op, 0, sizeof(op)); memset(mass, 0, sizeof(mass)); memset(trace, 0, sizeof(trace));
        for (int i = 0; i < fbs; i++) op[IDX(i, cbs + f * fbs + i, fbs)] = -1;
        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
        q_t cphi[QMAX_RBS], fphi[HHO_MAX_FBS];
        int nfq = cut_face_quadrature(m, c, f, 2 * facdeg, where, fx, fy, fw, HHO_MAX_GAUSS);   /* :602 */
        if (nfq < 0) { st = -nfq; break; }
        if (nfq == 0) continue;                                                              /* :612-613 */
        for (int q = 0; q < nfq; q++) {
            q_cell_basis(bar, hd, celdeg, fx[q], fy[q], cphi, NULL, NULL);
            q_face_basis(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            for (int j = 0; j < fbs; j++)
                for (int i = 0; i < fbs; i++) mass[IDX(i, j, fbs)] += (q_t)fw[q] * fphi[i] * fphi[j];
            for (int j = 0; j < cbs; j++)
                for (int i = 0; i < fbs; i++) trace[IDX(i, j, fbs)] += (q_t)fw[q] * fphi[i] * cphi[j];
        }
        memcpy(L, mass, sizeof(q_t) * fbs * fbs);
        if (q_llt_factor(L, fbs)) { st = HHO_ERR_NOT_SPD; break; }
        q_llt_solve(L, fbs, trace, cbs);
        memcpy(op, trace, sizeof(q_t) * fbs * cbs);
        for (int j = 0; j < msize; j++)
            for (int i = 0; i < msize; i++) {
                q_t s = 0;
                for (int k = 0; k < fbs; k++)
                    for (int l = 0; l < fbs; l++) s += op[IDX(l, i, fbs)] * mass[IDX(l, k, fbs)] * op[IDX(k, j, fbs)];
                acc[IDX(i, j, msize)] += s / hT;                                             /* :617 */
            }
    }
    for (int i = 0; i < msize * msize; i++) stab[i] = (double)acc[i];
    free(acc);
    return st;
}

/* cut make_rhs, cuthho_square.cpp:623-666, for a CUT cell; f_id / bcs_id as hho_builtin_fn (1: source, 2: solution) */
int cut_truth_rhs(const cut_mesh *m, const cut_level_set *ls, size_t c, int degree, int where, int f_id, int bcs_id, double *rhs)
{
    if (cut_mesh_cell_location(m)[c] != CUT_ON_INTERFACE) return HHO_ERR_ARG;
    if (f_id < 1 || f_id > 2 || bcs_id < 1 || bcs_id > 2) return HHO_ERR_ARG;
    q_t pts[8]; uint64_t ids[4]; double dpts[8];
    q_cell_pts(m, c, pts, ids, dpts);
    int cbs = hho_cell_basis_size(degree);
    q_t bar[2]; q_barycenter(pts, bar);
    q_t h = q_diameter(pts), hT = q_measure(pts);
    double *qx = malloc(sizeof(double) * 3 * CUT_MAX_QPS), *qy = qx + CUT_MAX_QPS, *qw = qy + CUT_MAX_QPS;
    q_t acc[QMAX_RBS], phi[QMAX_RBS], gx[QMAX_RBS], gy[QMAX_RBS];
    for (int i = 0; i < cbs; i++) acc[i] = 0;
    int st = HHO_OK;
    int nq = cut_cell_quadrature(m, c, 2 * degree, where, qx, qy, qw, CUT_MAX_QPS);          /* :639-644 */
    if (nq < 0) { st = -nq; goto done; }
    for (int q = 0; q < nq; q++) {
        q_cell_basis(bar, h, degree, qx[q], qy[q], phi, NULL, NULL);
        q_t fv = q_fn(f_id, qx[q], qy[q]);
        for (int i = 0; i < cbs; i++) acc[i] += (q_t)qw[q] * phi[i] * fv;
    }
    nq = cut_interface_quadrature(m, c, degree, where, qx, qy, qw, CUT_MAX_QPS);             /* :647: degree, not 2*degree */
    if (nq < 0) { st = -nq; goto done; }
    for (int q = 0; q < nq; q++) {
        q_t n[2];
        q_cell_basis(bar, h, degree, qx[q], qy[q], phi, gx, gy);
        q_ls_normal(ls, qx[q], qy[q], n);
        q_t bv = q_fn(bcs_id, qx[q], qy[q]);
        for (int i = 0; i < cbs; i++) acc[i] += (q_t)qw[q] * bv * (phi[i] * Q_ETA / hT - (gx[i] * n[0] + gy[i] * n[1]));   /* :654 */
    }
    for (int i = 0; i < cbs; i++) rhs[i] = (double)acc[i];
done:
    free(qx);
    return st;
}

/* make_hho_laplacian_interface, cuthho_square.cpp:390-502, for a CUT cell: data = gr_rhs^T gr_lhs^+ gr_rhs (2msize)^2.
 * gr_lhs (2rbs x 2rbs) is positive SEMI-definite with the kernel e_0 + e_rbs (the same constant on both sides), and the
 * columns of gr_rhs are orthogonal to it (row 0 + row rbs == 0 term by term): the solution set of gr_lhs X = gr_rhs is
 * X0 + (e_0 + e_rbs) t^T and data does not depend on t.  Here: X0 = the solution with X[0, :] = 0 (row / column 0 removed,
 * Cholesky of the rest in binary128).  oper (2rbs x 2msize) is that X0 -- the representative the product returns too. */
int cut_truth_laplacian_interface(const cut_mesh *m, const cut_level_set *ls, size_t c, hho_degrees di,
                                  const cut_params *parms, double *oper, double *data)
{
    if (cut_mesh_cell_location(m)[c] != CUT_ON_INTERFACE) return HHO_ERR_ARG;
    q_t pts[8]; uint64_t ids[4]; double dpts[8];
    q_cell_pts(m, c, pts, ids, dpts);
    int recdeg = di.rec_deg, celdeg = di.cell_deg, facdeg = di.face_deg;
    int rbs = hho_cell_basis_size(recdeg), cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs, n2 = 2 * rbs, m2 = 2 * msize;
    if (recdeg > HHO_MAX_RECDEG || cbs > rbs) return HHO_ERR_DEGREE;
    q_t bar[2]; q_barycenter(pts, bar);
    q_t h = q_diameter(pts), hT = q_measure(pts);
    q_t *stiff = calloc((size_t)n2 * n2, sizeof(q_t)), *gr_rhs = calloc((size_t)n2 * m2, sizeof(q_t));
    int n1 = n2 - 1;
    q_t *L = malloc(sizeof(q_t) * n1 * n1), *op = calloc((size_t)n2 * m2, sizeof(q_t)), *red = malloc(sizeof(q_t) * n1 * m2);
    double *qx = malloc(sizeof(double) * 3 * CUT_MAX_QPS), *qy = qx + CUT_MAX_QPS, *qw = qy + CUT_MAX_QPS;
    q_t gx[QMAX_RBS], gy[QMAX_RBS], phi[QMAX_RBS], fphi[HHO_MAX_FBS];
    const q_t kappa[2] = { parms->kappa_1, parms->kappa_2 };
    const q_t eta = parms->eta;
    int st = HHO_OK;

    for (int side = 0; side < 2; side++) {                                                   /* :419-432 */
        int nq = cut_cell_quadrature(m, c, 2 * recdeg, side == 0 ? CUT_NEG : CUT_POS, qx, qy, qw, CUT_MAX_QPS);
        if (nq < 0) { st = -nq; goto done; }
        int o = side * rbs;
        for (int q = 0; q < nq; q++) {
            q_cell_basis(bar, h, recdeg, qx[q], qy[q], NULL, gx, gy);
            for (int j = 0; j < rbs; j++)
                for (int i = 0; i < rbs; i++)
                    stiff[IDX(o + i, o + j, n2)] += kappa[side] * (q_t)qw[q] * (gx[i] * gx[j] + gy[i] * gy[j]);
        }
    }
    {
        int nq = cut_interface_quadrature(m, c, 2 * recdeg, CUT_NEG, qx, qy, qw, CUT_MAX_QPS);   /* :437 */
        if (nq < 0) { st = -nq; goto done; }
        for (int q = 0; q < nq; q++) {
            q_t n[2];
            q_cell_basis(bar, h, recdeg, qx[q], qy[q], phi, gx, gy);
            q_ls_normal(ls, qx[q], qy[q], n);
            for (int j = 0; j < rbs; j++) {
                q_t dnj = gx[j] * n[0] + gy[j] * n[1];
                for (int i = 0; i < rbs; i++) {
                    q_t dni = gx[i] * n[0] + gy[i] * n[1];
                    q_t a = kappa[0] * (q_t)qw[q] * phi[i] * dnj;                            /* :444 */
                    q_t b = kappa[0] * (q_t)qw[q] * dni * phi[j];                            /* :445 */
                    q_t cc = kappa[0] * (q_t)qw[q] * phi[i] * phi[j] * eta / hT;             /* :446 */
                    stiff[IDX(i, j, n2)] += cc - a - b;                                      /* :448-457 */
                    stiff[IDX(rbs + i, j, n2)] += a - cc;
                    stiff[IDX(i, rbs + j, n2)] += b - cc;
                    stiff[IDX(rbs + i, rbs + j, n2)] += cc;
                }
            }
        }
    }
    for (int j = 0; j < cbs; j++)                                                            /* :462-463 */
        for (int i = 0; i < n2; i++) {
            gr_rhs[IDX(i, j, n2)] = stiff[IDX(i, j, n2)];
            gr_rhs[IDX(i, cbs + j, n2)] = stiff[IDX(i, rbs + j, n2)];
        }
    q_t nrm[8]; q_normals(pts, nrm);
    for (int f = 0; f < 4; f++) {                                                            /* :465-495 */
        q_t fp0[2], fp1[2];
        q_face_points(pts, ids, f, fp0, fp1);
        for (int side = 0; side < 2; side++) {
            double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
            int nfq = cut_face_quadrature(m, c, f, 2 * recdeg, side == 0 ? CUT_NEG : CUT_POS, fx, fy, fw, HHO_MAX_GAUSS);
            if (nfq < 0) { st = -nfq; goto done; }
            int ro = side * rbs, co_cell = side * cbs, co_face = 2 * cbs + side * 4 * fbs + f * fbs;   /* :480,492 */
            for (int q = 0; q < nfq; q++) {
                q_cell_basis(bar, h, recdeg, fx[q], fy[q], phi, gx, gy);
                q_face_basis(fp0, fp1, facdeg, fx[q], fy[q], fphi);
                for (int i = 0; i < rbs; i++) {
                    q_t wdn = kappa[side] * (q_t)fw[q] * (gx[i] * nrm[2 * f] + gy[i] * nrm[2 * f + 1]);
                    for (int j = 0; j < cbs; j++) gr_rhs[IDX(ro + i, co_cell + j, n2)] -= wdn * phi[j];
                    for (int j = 0; j < fbs; j++) gr_rhs[IDX(ro + i, co_face + j, n2)] += wdn * fphi[j];
                }
            }
        }
    }
    for (int j = 0; j < n1; j++)                                                             /* :498, unknown 0 pinned */
        for (int i = 0; i < n1; i++) L[IDX(i, j, n1)] = stiff[IDX(i + 1, j + 1, n2)];
    for (int j = 0; j < m2; j++)
        for (int i = 0; i < n1; i++) red[IDX(i, j, n1)] = gr_rhs[IDX(i + 1, j, n2)];
    q_t na = 0;
    for (int j = 0; j < n1; j++) {
        q_t sa = 0;
        for (int i = 0; i < n1; i++) sa += fabsq(L[IDX(i, j, n1)]);
        if (sa > na) na = sa;
    }
    if (q_llt_factor(L, n1)) { st = HHO_ERR_NOT_SPD; goto done; }
    {
        q_t *I = calloc((size_t)n1 * n1, sizeof(q_t)), ni = 0;
        for (int i = 0; i < n1; i++) I[IDX(i, i, n1)] = 1;
        q_llt_solve(L, n1, I, n1);
        for (int j = 0; j < n1; j++) {
            q_t si = 0;
            for (int i = 0; i < n1; i++) si += fabsq(I[IDX(i, j, n1)]);
            if (si > ni) ni = si;
        }
        g_last_iface_cond = (double)(na * ni);
        free(I);
    }
    q_llt_solve(L, n1, red, m2);
    for (int j = 0; j < m2; j++)
        for (int i = 0; i < n1; i++) op[IDX(i + 1, j, n2)] = red[IDX(i, j, n1)];
    if (oper) for (int i = 0; i < n2 * m2; i++) oper[i] = (double)op[i];
    for (int j = 0; j < m2; j++)                                                             /* :499 */
        for (int i = 0; i < m2; i++) {
            q_t s = 0;
            for (int k = 0; k < n2; k++) s += gr_rhs[IDX(k, i, n2)] * op[IDX(k, j, n2)];
            data[IDX(i, j, m2)] = (double)s;
        }
done:
    free(stiff); free(gr_rhs); free(L); free(op); free(red); free(qx);
    return st;
}

/* make_rhs(msh, cl, degree, where, f), cuthho_utils.hpp:65-84 (one side of a cut cell) */
int cut_truth_rhs_side(const cut_mesh *m, size_t c, int degree, int where, int f_id, double *rhs)
{
    q_t pts[8]; uint64_t ids[4]; double dpts[8];
    q_cell_pts(m, c, pts, ids, dpts);
    int cbs = hho_cell_basis_size(degree);
    q_t bar[2]; q_barycenter(pts, bar);
    q_t h = q_diameter(pts);
    double *qx = malloc(sizeof(double) * 3 * CUT_MAX_QPS), *qy = qx + CUT_MAX_QPS, *qw = qy + CUT_MAX_QPS;
    q_t acc[QMAX_RBS], phi[QMAX_RBS];
    for (int i = 0; i < cbs; i++) acc[i] = 0;
    int nq = cut_cell_quadrature(m, c, 2 * degree, where, qx, qy, qw, CUT_MAX_QPS);
    if (nq < 0) { free(qx); return -nq; }
    for (int q = 0; q < nq; q++) {
        q_cell_basis(bar, h, degree, qx[q], qy[q], phi, NULL, NULL);
        q_t fv = q_fn(f_id, qx[q], qy[q]);
        for (int i = 0; i < cbs; i++) acc[i] += (q_t)qw[q] * phi[i] * fv;
    }
    for (int i = 0; i < cbs; i++) rhs[i] = (double)acc[i];
    free(qx);
    return HHO_OK;
}

/* condition number estimate of the cut reconstruction system in the 1-norm (|A|_1 |A^-1|_1, exact inverse in binary128):
 * what the tests print next to a sliver's error */
double cut_truth_laplacian_cond(const cut_mesh *m, const cut_level_set *ls, size_t c, hho_degrees di, int where)
{
    (void)ls;
    /* rebuilt from the operator: cheap enough (tests only) */
    int rbs = hho_cell_basis_size(di.rec_deg);
    q_t pts[8]; uint64_t ids[4]; double dpts[8];
    q_cell_pts(m, c, pts, ids, dpts);
    q_t bar[2]; q_barycenter(pts, bar);
    q_t h = q_diameter(pts), hT = q_measure(pts);
    q_t *A = calloc((size_t)rbs * rbs, sizeof(q_t)), *L = malloc(sizeof(q_t) * rbs * rbs), *I = calloc((size_t)rbs * rbs, sizeof(q_t));
    double *qx = malloc(sizeof(double) * 3 * CUT_MAX_QPS), *qy = qx + CUT_MAX_QPS, *qw = qy + CUT_MAX_QPS;
    q_t gx[QMAX_RBS], gy[QMAX_RBS], phi[QMAX_RBS];
    double cond = -1.0;
    int nq = cut_cell_quadrature(m, c, 2 * di.rec_deg, where, qx, qy, qw, CUT_MAX_QPS);
    if (nq < 0) goto done;
    for (int q = 0; q < nq; q++) {
        q_cell_basis(bar, h, di.rec_deg, qx[q], qy[q], NULL, gx, gy);
        for (int j = 0; j < rbs; j++)
            for (int i = 0; i < rbs; i++) A[IDX(i, j, rbs)] += (q_t)qw[q] * (gx[i] * gx[j] + gy[i] * gy[j]);
    }
    nq = cut_interface_quadrature(m, c, 2 * di.rec_deg, where, qx, qy, qw, CUT_MAX_QPS);
    if (nq < 0) goto done;
    for (int q = 0; q < nq; q++) {
        q_t n[2];
        q_cell_basis(bar, h, di.rec_deg, qx[q], qy[q], phi, gx, gy);
        q_ls_normal(ls, qx[q], qy[q], n);
        for (int j = 0; j < rbs; j++) {
            q_t dnj = gx[j] * n[0] + gy[j] * n[1];
            for (int i = 0; i < rbs; i++) {
                q_t dni = gx[i] * n[0] + gy[i] * n[1];
                A[IDX(i, j, rbs)] += (q_t)qw[q] * (phi[i] * phi[j] * Q_ETA / hT - phi[i] * dnj - dni * phi[j]);
            }
        }
    }
    memcpy(L, A, sizeof(q_t) * rbs * rbs);
    if (q_llt_factor(L, rbs)) goto done;
    for (int i = 0; i < rbs; i++) I[IDX(i, i, rbs)] = 1;
    q_llt_solve(L, rbs, I, rbs);
    {
        q_t na = 0, ni = 0;
        for (int j = 0; j < rbs; j++) {
            q_t sa = 0, si = 0;
            for (int i = 0; i < rbs; i++) { sa += fabsq(A[IDX(i, j, rbs)]); si += fabsq(I[IDX(i, j, rbs)]); }
            if (sa > na) na = sa;
            if (si > ni) ni = si;
        }
        cond = (double)(na * ni);
    }
done:
    free(A); free(L); free(I); free(qx);
    return cond;
}
