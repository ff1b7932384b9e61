/*
 * cuthho_oracle.c -- CPU restatement of the cutHHO preprocessing, cut quadrature and cut local
 * operators (see cuthho_oracle.h).  TEST INFRASTRUCTURE ONLY.
 */
#include "cuthho_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, ld) ((size_t)(i) + (size_t)(j) * (size_t)(ld))

struct cut_mesh {
    hho_mesh_params mp;
    size_t np, nc, nf;
    double *points;              /* np x 2 */
    uint64_t *ptids;             /* nc x 4 */
    uint64_t *faces;             /* nf x 2 */
    uint8_t *face_bnd;
    size_t *cell_faces;          /* nc x 4 */
    int8_t *node_loc;
    uint8_t *node_displaced;
    double *node_disp;           /* np x 2 */
    int8_t *face_loc;
    uint8_t *face_node_inside;
    double *face_ip;             /* nf x 2 */
    int8_t *cell_loc;
    double *cell_p0p1;           /* nc x 4 */
    size_t niface;               /* points per interface polyline */
    double *iface;               /* nc x niface x 2 (only cut cells filled) */
};

double cut_ls_eval(const cut_level_set *ls, double x, double y)
{
    if (ls->kind == CUT_LS_CIRCLE)                                     /* cuthho_square.cpp:65-71 */
        return (x - ls->alpha) * (x - ls->alpha) + (y - ls->beta) * (y - ls->beta) - ls->radius * ls->radius;
    return y - ls->cut_y;                                               /* :100-106 */
}

void cut_ls_normal(const cut_level_set *ls, double x, double y, double n[2])
{
    double gx, gy;
    if (ls->kind == CUT_LS_CIRCLE) { gx = 2 * x - 2 * ls->alpha; gy = 2 * y - 2 * ls->beta; }   /* :73-79 */
    else { gx = 0; gy = 1; }
    double nrm = sqrt(gx * gx + gy * gy);
    n[0] = gx / nrm; n[1] = gy / nrm;                                  /* :81-88 */
}

cut_mesh *cut_mesh_create(size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y)
{
    cut_mesh *m = (cut_mesh *)calloc(1, sizeof(cut_mesh));
    m->mp.Nx = Nx; m->mp.Ny = Ny; m->mp.min_x = min_x; m->mp.max_x = max_x; m->mp.min_y = min_y; m->mp.max_y = max_y;
    m->np = hho_mesh_num_points(&m->mp); m->nc = hho_mesh_num_cells(&m->mp); m->nf = hho_mesh_num_faces(&m->mp);
    m->points = (double *)calloc(2 * m->np, sizeof(double));
    m->ptids = (uint64_t *)calloc(4 * m->nc, sizeof(uint64_t));
    m->faces = (uint64_t *)calloc(2 * m->nf, sizeof(uint64_t));
    m->face_bnd = (uint8_t *)calloc(m->nf, 1);
    m->cell_faces = (size_t *)calloc(4 * m->nc, sizeof(size_t));
    hho_mesh_generate(&m->mp, m->points, m->ptids);
    hho_mesh_generate_faces(&m->mp, m->faces, m->face_bnd);
    for (size_t j = 0; j < Ny; j++)
        for (size_t i = 0; i < Nx; i++)
            for (int lf = 0; lf < 4; lf++) m->cell_faces[4 * (j * Nx + i) + lf] = hho_mesh_face_id(&m->mp, i, j, lf);
    m->node_loc = (int8_t *)malloc(m->np);
    memset(m->node_loc, CUT_UNDEF, m->np);
    m->node_displaced = (uint8_t *)calloc(m->np, 1);
    m->node_disp = (double *)calloc(2 * m->np, sizeof(double));
    m->face_loc = (int8_t *)malloc(m->nf);
    memset(m->face_loc, CUT_UNDEF, m->nf);
    m->face_node_inside = (uint8_t *)calloc(m->nf, 1);
    m->face_ip = (double *)calloc(2 * m->nf, sizeof(double));
    m->cell_loc = (int8_t *)malloc(m->nc);
    memset(m->cell_loc, CUT_UNDEF, m->nc);
    m->cell_p0p1 = (double *)calloc(4 * m->nc, sizeof(double));
    m->niface = 2;
    m->iface = NULL;
    return m;
}

void cut_mesh_free(cut_mesh *m)
{
    if (!m) return;
    free(m->points); free(m->ptids); free(m->faces); free(m->face_bnd); free(m->cell_faces);
    free(m->node_loc); free(m->node_displaced); free(m->node_disp); free(m->face_loc);
    free(m->face_node_inside); free(m->face_ip); free(m->cell_loc); free(m->cell_p0p1); free(m->iface);
    free(m);
}

size_t cut_mesh_num_points(const cut_mesh *m) { return m->np; }
size_t cut_mesh_num_cells(const cut_mesh *m) { return m->nc; }
size_t cut_mesh_num_faces(const cut_mesh *m) { return m->nf; }
const double *cut_mesh_points(const cut_mesh *m) { return m->points; }
const uint64_t *cut_mesh_cell_ptids(const cut_mesh *m) { return m->ptids; }
const uint64_t *cut_mesh_faces(const cut_mesh *m) { return m->faces; }
const uint8_t *cut_mesh_face_boundary(const cut_mesh *m) { return m->face_bnd; }
const int8_t *cut_mesh_node_location(const cut_mesh *m) { return m->node_loc; }
const int8_t *cut_mesh_face_location(const cut_mesh *m) { return m->face_loc; }
const double *cut_mesh_face_intersection(const cut_mesh *m) { return m->face_ip; }
const int8_t *cut_mesh_cell_location(const cut_mesh *m) { return m->cell_loc; }
size_t cut_mesh_interface_points(const cut_mesh *m) { return m->niface; }
const double *cut_mesh_cell_interface(const cut_mesh *m, size_t cell)
{
    if (m->cell_loc[cell] != CUT_ON_INTERFACE || !m->iface) return NULL;
    return m->iface + cell * m->niface * 2;
}
size_t cut_mesh_cell_face(const cut_mesh *m, size_t cell, int lf) { return m->cell_faces[4 * cell + lf]; }

/* cuthho_geom.hpp:68-116 */
static void find_zero_crossing(const double p0[2], const double p1[2], const cut_level_set *ls, double threshold, double out[2])
{
    double pa[2] = {p0[0], p0[1]}, pb[2] = {p1[0], p1[1]};
    double pm[2] = {(pa[0] + pb[0]) / 2.0, (pa[1] + pb[1]) / 2.0};
    double pm_prev[2];
    double xd, yd;
    size_t max_iter = 30;
    int cont;
    do {
        double lb = cut_ls_eval(ls, pb[0], pb[1]);
        double lm = cut_ls_eval(ls, pm[0], pm[1]);
        if ((lb >= 0 && lm >= 0) || (lb < 0 && lm < 0)) {      /* intersection is between pa and pm */
            pm_prev[0] = pm[0]; pm_prev[1] = pm[1];
            pb[0] = pm[0]; pb[1] = pm[1];
        } else {                                                 /* between pm and pb */
            pm_prev[0] = pm[0]; pm_prev[1] = pm[1];
            pa[0] = pm[0]; pa[1] = pm[1];
        }
        pm[0] = (pa[0] + pb[0]) / 2.0; pm[1] = (pa[1] + pb[1]) / 2.0;
        xd = (pm_prev[0] - pm[0]) * (pm_prev[0] - pm[0]);
        yd = (pm_prev[1] - pm[1]) * (pm_prev[1] - pm[1]);
        cont = (sqrt(xd + yd) > threshold) && (max_iter-- != 0);
    } while (cont);
    out[0] = pm[0]; out[1] = pm[1];
}

/* cuthho_geom.hpp:132-161 */
static void detect_cut_faces(cut_mesh *m, const cut_level_set *ls)
{
    for (size_t f = 0; f < m->nf; f++) {
        const double *p0 = &m->points[2 * m->faces[2 * f]], *p1 = &m->points[2 * m->faces[2 * f + 1]];
        double l0 = cut_ls_eval(ls, p0[0], p0[1]), l1 = cut_ls_eval(ls, p1[0], p1[1]);
        if (l0 >= 0 && l1 >= 0) { m->face_loc[f] = CUT_POS; continue; }
        if (l0 < 0 && l1 < 0) { m->face_loc[f] = CUT_NEG; continue; }
        double dx = p1[0] - p0[0], dy = p1[1] - p0[1];
        double threshold = sqrt(dx * dx + dy * dy) / 1e4;
        find_zero_crossing(p0, p1, ls, threshold, &m->face_ip[2 * f]);
        m->face_node_inside[f] = (l0 < 0) ? 0 : 1;
        m->face_loc[f] = CUT_ON_INTERFACE;
    }
}

/* cuthho_geom.hpp:466-543 (the version compiled: USE_OLD_DISPLACEMENT is off) */
static int move_nodes(cut_mesh *m)
{
    const double closeness_thresh = 0.4;
    for (size_t f = 0; f < m->nf; f++) {
        if (m->face_loc[f] != CUT_ON_INTERFACE) continue;
        uint64_t n0 = m->faces[2 * f], n1 = m->faces[2 * f + 1];
        const double *p0 = &m->points[2 * n0], *p1 = &m->points[2 * n1];
        double bar[2] = {(p1[0] + p0[0]) / 2.0, (p1[1] + p0[1]) / 2.0};
        double lf = sqrt((p1[0] - p0[0]) * (p1[0] - p0[0]) + (p1[1] - p0[1]) * (p1[1] - p0[1]));
        const double *ip = &m->face_ip[2 * f];
        double dp = sqrt((ip[0] - p0[0]) * (ip[0] - p0[0]) + (ip[1] - p0[1]) * (ip[1] - p0[1]));
        double closeness = dp / lf;
        uint64_t ntc;
        if (closeness < closeness_thresh) ntc = n0;
        else if (closeness > 1.0 - closeness_thresh) ntc = n1;
        else continue;
        double delta[2] = {(bar[0] - ip[0]) / 2, (bar[1] - ip[1]) / 2};
        m->node_disp[2 * ntc] = m->node_disp[2 * ntc] - delta[0];
        m->node_disp[2 * ntc + 1] = m->node_disp[2 * ntc + 1] - delta[1];
        m->node_displaced[ntc] = 1;
    }
    for (size_t n = 0; n < m->np; n++)
        if (m->node_displaced[n]) {
            m->points[2 * n] = m->points[2 * n] + m->node_disp[2 * n];
            m->points[2 * n + 1] = m->points[2 * n + 1] + m->node_disp[2 * n + 1];
        }
    for (size_t c = 0; c < m->nc; c++) {                      /* concavity check :517-541 */
        int distorted = 0;
        for (int v = 0; v < 4; v++) distorted |= m->node_displaced[m->ptids[4 * c + v]];
        if (!distorted) continue;
        for (int i = 0; i < 4; i++) {
            const double *pa = &m->points[2 * m->ptids[4 * c + i]];
            const double *pb = &m->points[2 * m->ptids[4 * c + (i + 1) % 4]];
            const double *pc = &m->points[2 * m->ptids[4 * c + (i + 2) % 4]];
            double v1x = pb[0] - pa[0], v1y = pb[1] - pa[1], v2x = pc[0] - pb[0], v2y = pc[1] - pb[1];
            if (v1x * v2y - v2x * v1y < 0) return 11;         /* "concave poly" */
        }
    }
    return 0;
}

/* cuthho_geom.hpp:275-340 */
static int detect_cut_cells(cut_mesh *m, const cut_level_set *ls)
{
    for (size_t c = 0; c < m->nc; c++) {
        int k = 0;
        double cutp[2][2];
        for (int i = 0; i < 4; i++) {
            size_t f = m->cell_faces[4 * c + i];
            if (m->face_loc[f] == CUT_ON_INTERFACE) {
                if (k < 2) { cutp[k][0] = m->face_ip[2 * f]; cutp[k][1] = m->face_ip[2 * f + 1]; }
                k++;
            }
        }
        if (k == 0) {
            int all_pos = 1;
            for (int v = 0; v < 4; v++) {
                const double *p = &m->points[2 * m->ptids[4 * c + v]];
                if (!(cut_ls_eval(ls, p[0], p[1]) > 0)) all_pos = 0;
            }
            m->cell_loc[c] = all_pos ? CUT_POS : CUT_NEG;
        } else if (k == 2) {
            m->cell_loc[c] = CUT_ON_INTERFACE;
            double ptx = cutp[1][0] - cutp[0][0], pty = cutp[1][1] - cutp[0][1];
            double pnx = cutp[0][0] + (-pty), pny = cutp[0][1] + ptx;
            double *pp = &m->cell_p0p1[4 * c];
            if (cut_ls_eval(ls, pnx, pny) >= 0) { pp[0] = cutp[1][0]; pp[1] = cutp[1][1]; pp[2] = cutp[0][0]; pp[3] = cutp[0][1]; }
            else { pp[0] = cutp[0][0]; pp[1] = cutp[0][1]; pp[2] = cutp[1][0]; pp[3] = cutp[1][1]; }
        } else return 12;                                       /* "invalid number of cuts in cell" */
    }
    return 0;
}

static double cell_diameter_of(const cut_mesh *m, size_t c)
{
    double pts[8];
    for (int v = 0; v < 4; v++) { pts[2 * v] = m->points[2 * m->ptids[4 * c + v]]; pts[2 * v + 1] = m->points[2 * m->ptids[4 * c + v] + 1]; }
    return hho_cell_diameter(pts);
}

/* cuthho_geom.hpp:609-651 */
static int refine_interface_rec(cut_mesh *m, size_t c, const cut_level_set *ls, size_t min, size_t max)
{
    if ((max - min) < 2) return 0;
    double *ifc = m->iface + c * m->niface * 2;
    size_t mid = (max + min) / 2;
    double p0[2] = {ifc[2 * min], ifc[2 * min + 1]}, p1[2] = {ifc[2 * max], ifc[2 * max + 1]};
    double pm[2] = {(p0[0] + p1[0]) / 2.0, (p0[1] + p1[1]) / 2.0};
    double ptx = p1[0] - p0[0], pty = p1[1] - p0[1];
    double pn[2] = {-pty, ptx};
    double ps1[2] = {pm[0] + pn[0], pm[1] + pn[1]}, ps2[2] = {pm[0] - pn[0], pm[1] - pn[1]};
    double lm = cut_ls_eval(ls, pm[0], pm[1]), ls1 = cut_ls_eval(ls, ps1[0], ps1[1]), ls2 = cut_ls_eval(ls, ps2[0], ps2[1]);
    double ip[2];
    if (!((lm >= 0 && ls1 >= 0) || (lm < 0 && ls1 < 0))) {
        find_zero_crossing(pm, ps1, ls, cell_diameter_of(m, c) / 10000.0, ip);
    } else if (!((lm >= 0 && ls2 >= 0) || (lm < 0 && ls2 < 0))) {
        find_zero_crossing(pm, ps2, ls, cell_diameter_of(m, c) / 10000.0, ip);
    } else return 13;                                           /* "interface not found in search range" */
    ifc[2 * mid] = ip[0]; ifc[2 * mid + 1] = ip[1];
    int st = refine_interface_rec(m, c, ls, min, mid);
    if (st) return st;
    return refine_interface_rec(m, c, ls, mid, max);
}

static int preprocess(cut_mesh *m, const cut_level_set *ls, int refsteps, int displace);

int cut_mesh_preprocess(cut_mesh *m, const cut_level_set *ls, int refsteps) { return preprocess(m, ls, refsteps, 1); }
/* the agglomeration branch of cuthho_square.cpp:2039-2044 (-A): no node displacement */
int cut_mesh_preprocess_agglomeration(cut_mesh *m, const cut_level_set *ls, int refsteps) { return preprocess(m, ls, refsteps, 0); }

/* detect_cell_agglo_set cuthho_geom.hpp:163-273 */
void cut_mesh_agglo_set(const cut_mesh *m, int8_t *agglo)
{
    const double threshold = 0.3;                                /* :170 */
    for (size_t c = 0; c < m->nc; c++) {
        agglo[c] = CUT_AGGLO_UNDEF;
        const size_t *fcs = m->cell_faces + 4 * c;
        const uint64_t *ids = m->ptids + 4 * c;
        int cut[4];
        double d[4][2];                                          /* d[n][0]: node n to the cut of face n-1, d[n][1]: to the cut of face n */
        for (int i = 0; i < 4; i++) cut[i] = m->face_loc[fcs[i]] == CUT_ON_INTERFACE;
        for (int n = 0; n < 4; n++)
            for (int s = 0; s < 2; s++) {
                size_t f = fcs[s == 0 ? (n == 0 ? 3 : n - 1) : n];
                const double *p0 = &m->points[2 * m->faces[2 * f]], *p1 = &m->points[2 * m->faces[2 * f + 1]];
                const double *pn = &m->points[2 * ids[n]], *ip = &m->face_ip[2 * f];
                double ma = sqrt((p1[0] - p0[0]) * (p1[0] - p0[0]) + (p1[1] - p0[1]) * (p1[1] - p0[1]));
                d[n][s] = sqrt((pn[0] - ip[0]) * (pn[0] - ip[0]) + (pn[1] - ip[1]) * (pn[1] - ip[1])) / ma;
            }
        for (int i = 0; i < 4; i++) {                            /* :243-252 -> agglo_set_single_node(n) :182-206 */
            int f1 = i, f2 = (i + 1) % 4, n = (i + 1) % 4;
            if (!(cut[f1] && cut[f2])) continue;
            double da = d[n][0], db = d[n][1];
            if ((da < db ? da : db) > threshold) agglo[c] = CUT_AGGLO_T_OK;
            else agglo[c] = m->node_loc[ids[n]] == CUT_NEG ? CUT_AGGLO_T_KO_NEG : CUT_AGGLO_T_KO_POS;
        }
        for (int f1 = 0; f1 < 2; f1++) {                         /* :254-258 -> agglo_set_double_node(f1, f2) :208-241 */
            int f2 = f1 + 2;
            if (!(cut[f1] && cut[f2])) continue;
            int n1 = f1, n2 = (f2 + 1) % 4;
            double da = d[n1][1];                                /* node n1 against face f1 == n1 */
            double db = d[n2][0];                                /* node n2 against face f2 == n2 - 1 */
            double m1 = da > db ? da : db, m2 = (1 - da) > (1 - db) ? (1 - da) : (1 - db);
            if ((m1 < m2 ? m1 : m2) > threshold) { agglo[c] = CUT_AGGLO_T_OK; continue; }
            if (m->node_loc[ids[n1]] == CUT_NEG) agglo[c] = (m1 <= threshold) ? CUT_AGGLO_T_KO_NEG : CUT_AGGLO_T_KO_POS;
            else agglo[c] = (m2 <= threshold) ? CUT_AGGLO_T_KO_NEG : CUT_AGGLO_T_KO_POS;
        }
    }
}

/* make_neighbors_info cuthho_geom.hpp:343-370, literally: all pairs, "share a point id" (O(cells^2)).
 * neighbors: nc x 8, ascending, -1 padded. */
void cut_mesh_neighbors(const cut_mesh *m, int32_t *neighbors)
{
    for (size_t i = 0; i < m->nc * 8; i++) neighbors[i] = -1;
    for (size_t i = 0; i < m->nc; i++) {
        int k = 0;
        for (size_t j = 0; j < m->nc; j++) {
            if (i == j) continue;
            int share = 0;
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++)
                    if (m->ptids[4 * i + a] == m->ptids[4 * j + b]) share = 1;
            if (share && k < 8) neighbors[8 * i + k++] = (int32_t)j;
        }
    }
}

static int preprocess(cut_mesh *m, const cut_level_set *ls, int refsteps, int displace)
{
    for (size_t n = 0; n < m->np; n++)                           /* detect_node_position :118-130 */
        m->node_loc[n] = cut_ls_eval(ls, m->points[2 * n], m->points[2 * n + 1]) < 0 ? CUT_NEG : CUT_POS;
    detect_cut_faces(m, ls);
    int st = 0;
    if (displace) {
        st = move_nodes(m);
        if (st) return st;
        detect_cut_faces(m, ls);                                 /* again, to update the intersection points :2048 */
    }
    st = detect_cut_cells(m, ls);
    if (st) return st;
    /* refine_interface :653-673 (levels == 0: the interface stays [p0, p1]) */
    size_t ipts = (size_t)1 << refsteps;
    m->niface = ipts + 1;
    free(m->iface);
    m->iface = (double *)calloc(m->nc * m->niface * 2, sizeof(double));
    for (size_t c = 0; c < m->nc; c++) {
        if (m->cell_loc[c] != CUT_ON_INTERFACE) continue;
        double *ifc = m->iface + c * m->niface * 2;
        ifc[0] = m->cell_p0p1[4 * c]; ifc[1] = m->cell_p0p1[4 * c + 1];
        ifc[2 * ipts] = m->cell_p0p1[4 * c + 2]; ifc[2 * ipts + 1] = m->cell_p0p1[4 * c + 3];
        if (refsteps > 0) {
            st = refine_interface_rec(m, c, ls, 0, ipts);
            if (st) return st;
        }
    }
    return 0;
}

/* ---- cut geometry ------------------------------------------------------------------------ */
static void cell_pts(const cut_mesh *m, size_t c, double pts[8], uint64_t ids[4])
{
    for (int v = 0; v < 4; v++) {
        ids[v] = m->ptids[4 * c + v];
        pts[2 * v] = m->points[2 * ids[v]]; pts[2 * v + 1] = m->points[2 * ids[v] + 1];
    }
}

/* collect_triangulation_points cuthho_geom.hpp:675-728 ; returns count, tp: n x 2 */
static int collect_tp(const cut_mesh *m, size_t c, int where, double *tp)
{
    const double *ifc = m->iface + c * m->niface * 2;
    int loc[4];
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    for (int v = 0; v < 4; v++) loc[v] = m->node_loc[ids[v]];
    int n = 0;
#define PUSH(x, y) do { tp[2 * n] = (x); tp[2 * n + 1] = (y); n++; } while (0)
#define INSERT_INTERFACE() do {                                                                   \
        if (where == CUT_NEG) for (size_t i = 0; i < m->niface; i++) PUSH(ifc[2 * i], ifc[2 * i + 1]);      \
        else for (size_t i = m->niface; i-- > 0;) PUSH(ifc[2 * i], ifc[2 * i + 1]);                       \
    } while (0)
    int case1 = loc[0] == where && loc[3] != where;
    int case2 = loc[0] != where && loc[3] == where;
    int case3 = loc[0] != where && loc[3] != where;
    if (case1 || case2 || case3) {
        for (int i = 0; i < 4; i++)
            if (loc[i] == where) PUSH(pts[2 * i], pts[2 * i + 1]);
        INSERT_INTERFACE();
    } else {
        int i = 0;
        while (i < 4 && loc[i] == where) { PUSH(pts[2 * i], pts[2 * i + 1]); i++; }
        INSERT_INTERFACE();
        while (i < 4 && loc[i] != where) i++;
        while (i < 4 && loc[i] == where) { PUSH(pts[2 * i], pts[2 * i + 1]); i++; }
    }
#undef PUSH
#undef INSERT_INTERFACE
    return n;
}

/* barycenter(begin, end) basic_geom.hpp:247-270 for n points */
static void poly_barycenter(const double *tp, int n, double bar[2])
{
    double rx = 0.0, ry = 0.0, den = 0.0;
    for (int i = 2; i < n; i++) {
        double ax = tp[2 * (i - 1)] - tp[0], ay = tp[2 * (i - 1) + 1] - tp[1];
        double bx = tp[2 * i] - tp[0], by = tp[2 * i + 1] - tp[1];
        double d = (ax * by - ay * bx) / 2.0;
        rx = rx + (ax + bx) * d; ry = ry + (ay + by) * d; den += d;
    }
    bar[0] = tp[0] + rx / (den * 3); bar[1] = tp[1] + ry / (den * 3);
}

#define CUT_MAX_TP 80

int cut_cell_quadrature(const cut_mesh *m, size_t c, int degree, int where, double *qx, double *qy, double *qw, int cap)
{
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    if (m->cell_loc[c] != CUT_ON_INTERFACE) {                   /* :802-803 */
        if (cap < HHO_MAX_CELL_QPS) return -HHO_ERR_ARG;
        return hho_cell_quadrature(pts, HHO_QUAD_FAN, degree, qx, qy, qw);
    }
    double tp[2 * CUT_MAX_TP], bar[2];
    if (m->niface + 4 > CUT_MAX_TP) return -HHO_ERR_ARG;
    int n = collect_tp(m, c, where, tp);
    poly_barycenter(tp, n, bar);
    int k = 0;
    for (int i = 0; i < n; i++) {                               /* triangulate :754-777 + :806-812 */
        const double *p1 = &tp[2 * i], *p2 = &tp[2 * ((i + 1) % n)];
        if (k + 16 > cap) return -HHO_ERR_ARG;
        int nq = hho_triangle_quadrature(bar, p1, p2, degree, qx + k, qy + k, qw + k);
        if (nq < 0) return nq;
        k += nq;
    }
    return k;
}

double cut_cell_measure(const cut_mesh *m, size_t c, int where)
{
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    if (m->cell_loc[c] != CUT_ON_INTERFACE) return hho_cell_measure(pts);
    double tp[2 * CUT_MAX_TP], bar[2];
    int n = collect_tp(m, c, where, tp);
    poly_barycenter(tp, n, bar);
    double tot = 0.0;
    for (int i = 0; i < n; i++) {
        const double *p1 = &tp[2 * i], *p2 = &tp[2 * ((i + 1) % n)];
        double v1x = p1[0] - bar[0], v1y = p1[1] - bar[1], v2x = p2[0] - bar[0], v2y = p2[1] - bar[1];
        tot += fabs(v1x * v2y - v2x * v1y) / 2.0;
    }
    return tot;
}

/* points(msh, fc, where) cuthho_geom.hpp:546-569 ; returns 0 ok, <0 error */
static int face_points_where(const cut_mesh *m, size_t f, int where, double p0[2], double p1[2])
{
    if (m->face_loc[f] != where && m->face_loc[f] != CUT_ON_INTERFACE) return -1;
    uint64_t n0 = m->faces[2 * f], n1 = m->faces[2 * f + 1];
    p0[0] = m->points[2 * n0]; p0[1] = m->points[2 * n0 + 1];
    p1[0] = m->points[2 * n1]; p1[1] = m->points[2 * n1 + 1];
    if (m->face_loc[f] != CUT_ON_INTERFACE) return 0;
    if (m->node_loc[n0] == where && m->node_loc[n1] != where) { p1[0] = m->face_ip[2 * f]; p1[1] = m->face_ip[2 * f + 1]; }
    else if (m->node_loc[n0] != where && m->node_loc[n1] == where) { p0[0] = m->face_ip[2 * f]; p0[1] = m->face_ip[2 * f + 1]; }
    else return -2;                                             /* "Invalid point configuration" */
    return 0;
}

int cut_face_quadrature(const cut_mesh *m, size_t c, int lf, int degree, int where, double *qx, double *qy, double *qw, int cap)
{
    size_t f = m->cell_faces[4 * c + lf];
    if (cap < HHO_MAX_GAUSS) return -HHO_ERR_ARG;
    if (m->face_loc[f] != where && m->face_loc[f] != CUT_ON_INTERFACE) return 0;     /* :823-824 */
    double p0[2], p1[2];
    if (m->face_loc[f] != CUT_ON_INTERFACE) {                                        /* :826-827 */
        p0[0] = m->points[2 * m->faces[2 * f]]; p0[1] = m->points[2 * m->faces[2 * f] + 1];
        p1[0] = m->points[2 * m->faces[2 * f + 1]]; p1[1] = m->points[2 * m->faces[2 * f + 1] + 1];
        return hho_face_quadrature(p0, p1, degree, qx, qy, qw);
    }
    if (face_points_where(m, f, where, p0, p1) < 0) return -HHO_ERR_ARG;
    return hho_face_quadrature(p0, p1, degree, qx, qy, qw);                          /* :829-846: same formulas */
}

int cut_interface_quadrature(const cut_mesh *m, size_t c, int degree, int where, double *qx, double *qy, double *qw, int cap)
{
    if (m->cell_loc[c] != CUT_ON_INTERFACE) return -HHO_ERR_ARG;
    const double *ifc = m->iface + c * m->niface * 2;
    double tp[2 * CUT_MAX_TP], bar[2];
    int n = collect_tp(m, c, where, tp);                       /* barycenter(msh, cl, where) :592-606 */
    poly_barycenter(tp, n, bar);
    double vax = ifc[0] - bar[0], vay = ifc[1] - bar[1];
    double vtx = ifc[2] - ifc[0], vty = ifc[3] - ifc[1];
    double vbx = vty, vby = -vtx;
    double int_sign = (vax * vbx + vay * vby) < 0 ? -1.0 : +1.0;
    double nd[HHO_MAX_GAUSS], wt[HHO_MAX_GAUSS];
    int ng = hho_gauss_legendre(degree, nd, wt);
    if (ng < 0) return ng;
    int k = 0;
    for (size_t i = 1; i < m->niface; i++) {
        const double *p0 = &ifc[2 * (i - 1)], *p1 = &ifc[2 * i];
        double sx = p1[0] - p0[0], sy = p1[1] - p0[1];
        double meas = sqrt(sx * sx + sy * sy);
        for (int q = 0; q < ng; q++) {
            if (k >= cap) return -HHO_ERR_ARG;
            double t = nd[q];
            qx[k] = 0.5 * (1 - t) * p0[0] + 0.5 * (1 + t) * p1[0];
            qy[k] = 0.5 * (1 - t) * p0[1] + 0.5 * (1 + t) * p1[1];
            qw[k] = int_sign * wt[q] * meas * 0.5;
            k++;
        }
    }
    return k;
}

/* ---- cut operators ------------------------------------------------------------------------- */
static const double CELL_ETA = 5.0;                             /* cuthho_square.cpp:301-306 */

int cut_make_hho_laplacian(const cut_mesh *m, const cut_level_set *ls, size_t c, hho_degrees di, int where,
                           double *oper, double *data, int *oper_rows)
{
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    int recdeg = di.rec_deg, celdeg = di.cell_deg, facdeg = di.face_deg;
    int rbs = hho_cell_basis_size(recdeg), cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs;
    if (m->cell_loc[c] != CUT_ON_INTERFACE) {                   /* :316-317 */
        *oper_rows = rbs - 1;
        return hho_make_laplacian(pts, ids, di, HHO_QUAD_FAN, oper, data);
    }
    if (recdeg > HHO_MAX_RECDEG || cbs > rbs) return HHO_ERR_DEGREE;
    *oper_rows = rbs;
    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);
    double stiff[HHO_MAX_RBS * HHO_MAX_RBS], gr_lhs[HHO_MAX_RBS * HHO_MAX_RBS], gr_rhs[HHO_MAX_RBS * HHO_MAX_MSIZE];
    memset(stiff, 0, sizeof(double) * rbs * rbs);
    memset(gr_rhs, 0, sizeof(double) * rbs * msize);
    static double qx[CUT_MAX_QPS], qy[CUT_MAX_QPS], qw[CUT_MAX_QPS];
    double gx[HHO_MAX_RBS], gy[HHO_MAX_RBS], phi[HHO_MAX_RBS], fphi[HHO_MAX_FBS];

    int nq = cut_cell_quadrature(m, c, 2 * recdeg, where, qx, qy, qw, CUT_MAX_QPS);     /* :336-341 */
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        hho_cell_basis_grad(bar, h, recdeg, qx[q], qy[q], gx, gy);
        for (int j = 0; j < rbs; j++)
            for (int i = 0; i < rbs; i++) stiff[IDX(i, j, rbs)] += (qw[q] * gx[i]) * gx[j] + (qw[q] * gy[i]) * gy[j];
    }
    double hT = hho_cell_measure(pts);                          /* :344 the WHOLE cell's area */
    nq = cut_interface_quadrature(m, c, 2 * recdeg, where, qx, qy, qw, CUT_MAX_QPS);    /* :347-360 */
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        double n[2];
        hho_cell_basis_eval(bar, h, recdeg, qx[q], qy[q], phi);
        hho_cell_basis_grad(bar, h, recdeg, qx[q], qy[q], gx, gy);
        cut_ls_normal(ls, qx[q], qy[q], n);                     /* not flipped for the positive side, :352-355 */
        for (int j = 0; j < rbs; j++) {
            double dnj = gx[j] * n[0] + gy[j] * n[1];
            for (int i = 0; i < rbs; i++) {
                double dni = gx[i] * n[0] + gy[i] * n[1];
                stiff[IDX(i, j, rbs)] -= (qw[q] * phi[i]) * dnj;
                stiff[IDX(i, j, rbs)] -= (qw[q] * dni) * phi[j];
                stiff[IDX(i, j, rbs)] += (qw[q] * phi[i]) * phi[j] * CELL_ETA / hT;
            }
        }
    }
    memcpy(gr_lhs, stiff, sizeof(double) * rbs * rbs);          /* :362 */
    for (int j = 0; j < cbs; j++)                               /* :363 */
        for (int i = 0; i < rbs; i++) gr_rhs[IDX(i, j, rbs)] = stiff[IDX(i, j, rbs)];
    double nrm[8]; hho_cell_normals(pts, nrm);
    for (int f = 0; f < 4; f++) {                               /* :366-383 */
        double fp0[2], fp1[2];
        hho_cell_face_points(pts, ids, f, fp0, fp1);            /* face_basis of the WHOLE face */
        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
        int nfq = cut_face_quadrature(m, c, f, 2 * recdeg, where, fx, fy, fw, HHO_MAX_GAUSS);
        if (nfq < 0) return -nfq;
        for (int q = 0; q < nfq; q++) {
            hho_cell_basis_eval(bar, h, recdeg, fx[q], fy[q], phi);
            hho_cell_basis_grad(bar, h, recdeg, fx[q], fy[q], gx, gy);
            hho_face_basis_eval(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            for (int i = 0; i < rbs; i++) {
                double wdn = fw[q] * (gx[i] * nrm[2 * f] + gy[i] * nrm[2 * f + 1]);
                for (int j = 0; j < fbs; j++) gr_rhs[IDX(i, cbs + f * fbs + j, rbs)] += wdn * fphi[j];
                for (int j = 0; j < cbs; j++) gr_rhs[IDX(i, j, rbs)] -= wdn * phi[j];
            }
        }
    }
    int bad = hho_llt_factor(gr_lhs, rbs);                      /* :385 */
    memcpy(oper, gr_rhs, sizeof(double) * rbs * msize);
    hho_llt_solve_inplace(gr_lhs, rbs, oper, msize);
    for (int j = 0; j < msize; j++)                             /* :386 */
        for (int i = 0; i < msize; i++) {
            double s = 0.0;
            for (int k = 0; k < rbs; k++) s += gr_rhs[IDX(k, i, rbs)] * oper[IDX(k, j, rbs)];
            data[IDX(i, j, msize)] = s;
        }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

int cut_make_hho_cut_stabilization(const cut_mesh *m, size_t c, hho_degrees di, int where, double *data)
{
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    if (m->cell_loc[c] != CUT_ON_INTERFACE) return hho_make_naive_stabilization(pts, ids, di, data);   /* :572-573 */
    int celdeg = di.cell_deg, facdeg = di.face_deg;
    int cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg), msize = cbs + 4 * fbs;
    memset(data, 0, sizeof(double) * msize * msize);
    double bar[2]; hho_cell_barycenter(pts, bar);
    double hd = hho_cell_diameter(pts);
    double hT = hho_cell_measure(pts);                          /* :589 */
    int bad = 0;
    for (int f = 0; f < 4; f++) {
        double fp0[2], fp1[2];
        hho_cell_face_points(pts, ids, f, fp0, fp1);
        double oper[HHO_MAX_FBS * HHO_MAX_MSIZE], mass[HHO_MAX_FBS * HHO_MAX_FBS], L[HHO_MAX_FBS * HHO_MAX_FBS];
        double trace[HHO_MAX_FBS * HHO_MAX_RBS];
        memset(oper, 0, sizeof(double) * fbs * msize);
        memset(mass, 0, sizeof(double) * fbs * fbs);
        memset(trace, 0, sizeof(double) * fbs * cbs);
        for (int i = 0; i < fbs; i++) oper[IDX(i, cbs + f * fbs + i, fbs)] = -1.0;
        double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS], cphi[HHO_MAX_RBS], fphi[HHO_MAX_FBS];
        int nfq = cut_face_quadrature(m, c, f, 2 * facdeg, where, fx, fy, fw, HHO_MAX_GAUSS);       /* :602 */
        if (nfq < 0) return -nfq;
        for (int q = 0; q < nfq; q++) {
            hho_cell_basis_eval(bar, hd, celdeg, fx[q], fy[q], cphi);
            hho_face_basis_eval(fp0, fp1, facdeg, fx[q], fy[q], fphi);
            for (int j = 0; j < fbs; j++)
                for (int i = 0; i < fbs; i++) mass[IDX(i, j, fbs)] += (fw[q] * fphi[i]) * fphi[j];
            for (int j = 0; j < cbs; j++)
                for (int i = 0; i < fbs; i++) trace[IDX(i, j, fbs)] += (fw[q] * fphi[i]) * cphi[j];
        }
        if (nfq == 0) continue;                                 /* :612-613 */
        memcpy(L, mass, sizeof(double) * fbs * fbs);
        if (hho_llt_factor(L, fbs)) bad = 1;
        hho_llt_solve_inplace(L, fbs, trace, cbs);
        memcpy(oper, trace, sizeof(double) * fbs * cbs);
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
                data[IDX(i, j, msize)] += s * (1. / hT);        /* :617 */
            }
    }
    return bad ? HHO_ERR_NOT_SPD : HHO_OK;
}

int cut_make_rhs(const cut_mesh *m, const cut_level_set *ls, size_t c, int degree, int where,
                 hho_scalar_fn f, hho_scalar_fn bcs, void *user, double *rhs)
{
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    int cbs = hho_cell_basis_size(degree);
    if (m->cell_loc[c] == where) return hho_cell_rhs(pts, HHO_QUAD_FAN, degree, 0, f, user, rhs);   /* :628-629 */
    memset(rhs, 0, sizeof(double) * cbs);
    if (m->cell_loc[c] != CUT_ON_INTERFACE) return HHO_OK;                                           /* :659-664 */
    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);
    double hT = hho_cell_measure(pts);
    static double qx[CUT_MAX_QPS], qy[CUT_MAX_QPS], qw[CUT_MAX_QPS];
    double phi[HHO_MAX_RBS], gx[HHO_MAX_RBS], gy[HHO_MAX_RBS];
    int nq = cut_cell_quadrature(m, c, 2 * degree, where, qx, qy, qw, CUT_MAX_QPS);                  /* :639-644 */
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        hho_cell_basis_eval(bar, h, degree, qx[q], qy[q], phi);
        double fv = f(qx[q], qy[q], user);
        for (int i = 0; i < cbs; i++) rhs[i] += (qw[q] * phi[i]) * fv;
    }
    nq = cut_interface_quadrature(m, c, degree, where, qx, qy, qw, CUT_MAX_QPS);                     /* :647: degree, not 2*degree */
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        double n[2];
        hho_cell_basis_eval(bar, h, degree, qx[q], qy[q], phi);
        hho_cell_basis_grad(bar, h, degree, qx[q], qy[q], gx, gy);
        cut_ls_normal(ls, qx[q], qy[q], n);
        double bv = bcs(qx[q], qy[q], user);
        for (int i = 0; i < cbs; i++)
            rhs[i] += (qw[q] * bv) * (phi[i] * CELL_ETA / hT - (gx[i] * n[0] + gy[i] * n[1]));      /* :654 */
    }
    return HHO_OK;
}

/* ---- two-sided interface problem ----------------------------------------------------------- */
/* LDL^T with symmetric diagonal pivoting (what Eigen's LDLT computes, cuthho_square.cpp:498):
 * P A P^T = L D L^T, at step k the largest remaining |diagonal| entry is moved to position k.
 * solve: x = P^T L^-T D^+ L^-1 P b, D^+ the pseudo-inverse of D (tolerance: see ldlt_solve). */
static void ldlt_factor(double *A, int n, int *perm)
{
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int i = k + 1; i < n; i++)
            if (fabs(A[IDX(i, i, n)]) > fabs(A[IDX(p, p, n)])) p = i;
        if (p != k) {                                       /* symmetric swap of rows/columns k and p (full storage) */
            for (int j = 0; j < n; j++) { double t = A[IDX(k, j, n)]; A[IDX(k, j, n)] = A[IDX(p, j, n)]; A[IDX(p, j, n)] = t; }
            for (int i = 0; i < n; i++) { double t = A[IDX(i, k, n)]; A[IDX(i, k, n)] = A[IDX(i, p, n)]; A[IDX(i, p, n)] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        double d = A[IDX(k, k, n)];
        if (d == 0.0) continue;
        for (int i = k + 1; i < n; i++) A[IDX(i, k, n)] /= d;                     /* column of L */
        for (int j = k + 1; j < n; j++)                                             /* trailing update */
            for (int i = j; i < n; i++) {
                A[IDX(i, j, n)] -= A[IDX(i, k, n)] * d * A[IDX(j, k, n)];
                A[IDX(j, i, n)] = A[IDX(i, j, n)];
            }
    }
}

static void ldlt_solve(const double *F, int n, const int *perm, double *B, int nrhs)
{
    double y[2 * HHO_MAX_RBS];
    /* pseudo-inverse of D.  Tolerance of Eigen <= 3.2 (max|D| * epsilon).  Eigen 3.3+ uses
     * numeric_limits::min(): the semi-definite system's last pivot is rounding noise (1e-31 .. 1e-15
     * here, rows 0 and rbs of gr_lhs being bitwise negatives of each other), which that tolerance
     * inverts into an O(1e15) multiple of the kernel vector and destroys `data`; the committed
     * cuthho.xlsx numbers are only reproduced with the kernel component suppressed. */
    double dmax = 0.0;
    for (int i = 0; i < n; i++) dmax = fmax(dmax, fabs(F[IDX(i, i, n)]));
    const double tol = fmax(dmax * DBL_EPSILON, DBL_MIN);
    for (int c = 0; c < nrhs; c++) {
        double *b = B + (size_t)c * n;
        for (int i = 0; i < n; i++) y[i] = b[perm[i]];
        for (int i = 0; i < n; i++)
            for (int k = 0; k < i; k++) y[i] -= F[IDX(i, k, n)] * y[k];
        for (int i = 0; i < n; i++) y[i] = fabs(F[IDX(i, i, n)]) > tol ? y[i] / F[IDX(i, i, n)] : 0.0;
        for (int i = n - 1; i >= 0; i--)
            for (int k = i + 1; k < n; k++) y[i] -= F[IDX(k, i, n)] * y[k];
        for (int i = 0; i < n; i++) b[perm[i]] = y[i];
    }
}

int cut_make_hho_laplacian_interface(const cut_mesh *m, const cut_level_set *ls, size_t c, hho_degrees di,
                                     const cut_params *parms, double *oper, double *data)
{
    if (m->cell_loc[c] != CUT_ON_INTERFACE) return HHO_ERR_ARG;        /* :397-398 "The cell is not cut" */
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    int recdeg = di.rec_deg, celdeg = di.cell_deg, facdeg = di.face_deg;
    int rbs = hho_cell_basis_size(recdeg), cbs = hho_cell_basis_size(celdeg), fbs = hho_face_basis_size(facdeg);
    int msize = cbs + 4 * fbs, n2 = 2 * rbs, m2 = 2 * msize;
    if (recdeg > HHO_MAX_RECDEG || cbs > rbs) return HHO_ERR_DEGREE;
    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);
    static double stiff[4 * HHO_MAX_RBS * HHO_MAX_RBS], gr_lhs[4 * HHO_MAX_RBS * HHO_MAX_RBS];
    static double gr_rhs[4 * HHO_MAX_RBS * HHO_MAX_MSIZE];
    memset(stiff, 0, sizeof(double) * n2 * n2);
    memset(gr_rhs, 0, sizeof(double) * n2 * m2);
    static double qx[CUT_MAX_QPS], qy[CUT_MAX_QPS], qw[CUT_MAX_QPS];
    double gx[HHO_MAX_RBS], gy[HHO_MAX_RBS], phi[HHO_MAX_RBS], fphi[HHO_MAX_FBS];
    const double kappa[2] = { parms->kappa_1, parms->kappa_2 };

    for (int side = 0; side < 2; side++) {                              /* :419-432 */
        int nq = cut_cell_quadrature(m, c, 2 * recdeg, side == 0 ? CUT_NEG : CUT_POS, qx, qy, qw, CUT_MAX_QPS);
        if (nq < 0) return -nq;
        int o = side * rbs;
        for (int q = 0; q < nq; q++) {
            hho_cell_basis_grad(bar, h, recdeg, qx[q], qy[q], gx, gy);
            for (int j = 0; j < rbs; j++)
                for (int i = 0; i < rbs; i++)
                    stiff[IDX(o + i, o + j, n2)] += kappa[side] * ((qw[q] * gx[i]) * gx[j] + (qw[q] * gy[i]) * gy[j]);
        }
    }
    double hT = hho_cell_measure(pts);                                  /* :434 */
    int nq = cut_interface_quadrature(m, c, 2 * recdeg, CUT_NEG, qx, qy, qw, CUT_MAX_QPS);   /* :437 */
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        double n[2];
        hho_cell_basis_eval(bar, h, recdeg, qx[q], qy[q], phi);
        hho_cell_basis_grad(bar, h, recdeg, qx[q], qy[q], gx, gy);
        cut_ls_normal(ls, qx[q], qy[q], n);
        for (int j = 0; j < rbs; j++) {
            double dnj = gx[j] * n[0] + gy[j] * n[1];
            for (int i = 0; i < rbs; i++) {
                double dni = gx[i] * n[0] + gy[i] * n[1];
                double a = parms->kappa_1 * qw[q] * phi[i] * dnj;                 /* :444 */
                double b = parms->kappa_1 * qw[q] * dni * phi[j];                 /* :445 */
                double cc = parms->kappa_1 * qw[q] * phi[i] * phi[j] * parms->eta / hT;   /* :446 */
                stiff[IDX(i, j, n2)] -= a;           stiff[IDX(rbs + i, j, n2)] += a;     /* :448-449 */
                stiff[IDX(i, j, n2)] -= b;           stiff[IDX(i, rbs + j, n2)] += b;     /* :451-452 */
                stiff[IDX(i, j, n2)] += cc;          stiff[IDX(i, rbs + j, n2)] -= cc;    /* :454-457 */
                stiff[IDX(rbs + i, j, n2)] -= cc;    stiff[IDX(rbs + i, rbs + j, n2)] += cc;
            }
        }
    }
    memcpy(gr_lhs, stiff, sizeof(double) * n2 * n2);                    /* :461 */
    for (int j = 0; j < cbs; j++)                                       /* :462-463 */
        for (int i = 0; i < n2; i++) {
            gr_rhs[IDX(i, j, n2)] = stiff[IDX(i, j, n2)];
            gr_rhs[IDX(i, cbs + j, n2)] = stiff[IDX(i, rbs + j, n2)];
        }
    double nrm[8]; hho_cell_normals(pts, nrm);
    for (int f = 0; f < 4; f++) {                                       /* :465-495 */
        double fp0[2], fp1[2];
        hho_cell_face_points(pts, ids, f, fp0, fp1);
        for (int side = 0; side < 2; side++) {
            double fx[HHO_MAX_GAUSS], fy[HHO_MAX_GAUSS], fw[HHO_MAX_GAUSS];
            int nfq = cut_face_quadrature(m, c, f, 2 * recdeg, side == 0 ? CUT_NEG : CUT_POS, fx, fy, fw, HHO_MAX_GAUSS);
            if (nfq < 0) return -nfq;
            int ro = side * rbs, co_cell = side * cbs, co_face = 2 * cbs + side * 4 * fbs + f * fbs;   /* :480,492 */
            for (int q = 0; q < nfq; q++) {
                hho_cell_basis_eval(bar, h, recdeg, fx[q], fy[q], phi);
                hho_cell_basis_grad(bar, h, recdeg, fx[q], fy[q], gx, gy);
                hho_face_basis_eval(fp0, fp1, facdeg, fx[q], fy[q], fphi);
                for (int i = 0; i < rbs; i++) {
                    double wdn = kappa[side] * fw[q] * (gx[i] * nrm[2 * f] + gy[i] * nrm[2 * f + 1]);
                    for (int j = 0; j < cbs; j++) gr_rhs[IDX(ro + i, co_cell + j, n2)] -= wdn * phi[j];
                    for (int j = 0; j < fbs; j++) gr_rhs[IDX(ro + i, co_face + j, n2)] += wdn * fphi[j];
                }
            }
        }
    }
    int perm[2 * HHO_MAX_RBS];
    ldlt_factor(gr_lhs, n2, perm);                                       /* :498 */
    memcpy(oper, gr_rhs, sizeof(double) * n2 * m2);
    ldlt_solve(gr_lhs, n2, perm, oper, m2);
    for (int j = 0; j < m2; j++)                                         /* :499 */
        for (int i = 0; i < m2; i++) {
            double s = 0.0;
            for (int k = 0; k < n2; k++) s += gr_rhs[IDX(k, i, n2)] * oper[IDX(k, j, n2)];
            data[IDX(i, j, m2)] = s;
        }
    return HHO_OK;
}

int cut_make_rhs_side(const cut_mesh *m, size_t c, int degree, int where, hho_scalar_fn f, void *user, double *rhs)
{
    double pts[8]; uint64_t ids[4];
    cell_pts(m, c, pts, ids);
    int cbs = hho_cell_basis_size(degree);
    memset(rhs, 0, sizeof(double) * cbs);
    double bar[2]; hho_cell_barycenter(pts, bar);
    double h = hho_cell_diameter(pts);
    static double qx[CUT_MAX_QPS], qy[CUT_MAX_QPS], qw[CUT_MAX_QPS];
    double phi[HHO_MAX_RBS];
    int nq = cut_cell_quadrature(m, c, 2 * degree, where, qx, qy, qw, CUT_MAX_QPS);    /* cuthho_utils.hpp:75 */
    if (nq < 0) return -nq;
    for (int q = 0; q < nq; q++) {
        hho_cell_basis_eval(bar, h, degree, qx[q], qy[q], phi);
        double fv = f(qx[q], qy[q], user);
        for (int i = 0; i < cbs; i++) rhs[i] += (qw[q] * phi[i]) * fv;
    }
    return HHO_OK;
}

void cut_interface_tables(const cut_mesh *m, int64_t *cell_table, int64_t *face_table,
                          size_t *num_all_cells, size_t *num_other_faces)
{
    size_t nc = 0;
    for (size_t c = 0; c < m->nc; c++) {                             /* :1142-1150 */
        cell_table[c] = (int64_t)nc;
        nc += m->cell_loc[c] == CUT_ON_INTERFACE ? 2 : 1;
    }
    size_t co = 0;
    for (size_t f = 0; f < m->nf; f++) {                             /* :1167-1178 */
        if (m->face_bnd[f]) { face_table[f] = -1; continue; }            /* every boundary face is Dirichlet */
        face_table[f] = (int64_t)co;
        co += m->face_loc[f] == CUT_ON_INTERFACE ? 2 : 1;
    }
    *num_all_cells = nc;
    *num_other_faces = co;   /* == num_all_faces - num_dirichlet_faces (:1162-1163): cut faces are never on the boundary */
}

int cut_interface_assemble(const cut_mesh *m, hho_degrees di, size_t c, const int64_t *cell_table,
                           const int64_t *face_table, size_t num_all_cells,
                           const double *lhs, const double *rhs, const double *dirichlet_data,
                           int32_t *trip_rows, int32_t *trip_cols, double *trip_vals, size_t *ntrip,
                           int64_t *rhs_rows, double *rhs_vals)
{
    int cbs = hho_cell_basis_size(di.cell_deg), fbs = hho_face_basis_size(di.face_deg);
    int msize = cbs + 4 * fbs;
    int cut = m->cell_loc[c] == CUT_ON_INTERFACE;
    int n = cut ? 2 * msize : msize;
    int64_t idx[2 * HHO_MAX_MSIZE];
    int assem[2 * HHO_MAX_MSIZE];
    int64_t cell_LHS_offset = cell_table[c] * cbs;                       /* :1223, :1291 */
    int ncd = cut ? 2 * cbs : cbs;
    for (int i = 0; i < ncd; i++) { idx[i] = cell_LHS_offset + i; assem[i] = 1; }
    for (int pass = 0; pass < (cut ? 2 : 1); pass++)                     /* :1230-1247 ; :1296-1333 */
        for (int f = 0; f < 4; f++) {
            size_t fid = m->cell_faces[4 * c + f];
            int dirichlet = m->face_bnd[fid];
            if (cut && dirichlet) return HHO_ERR_ARG;                    /* "Dirichlet boundary on cut cell not supported." */
            int64_t d = (pass == 1 && m->face_loc[fid] == CUT_ON_INTERFACE) ? fbs : 0;    /* :1319 */
            int64_t face_LHS_offset = (int64_t)num_all_cells * cbs + (dirichlet ? 0 : face_table[fid]) * fbs + d;
            for (int i = 0; i < fbs; i++) {
                idx[ncd + pass * 4 * fbs + f * fbs + i] = face_LHS_offset + i;
                assem[ncd + pass * 4 * fbs + f * fbs + i] = !dirichlet;
            }
        }
    size_t nt = 0;
    for (int i = 0; i < n; i++) { rhs_rows[i] = assem[i] ? idx[i] : -1; rhs_vals[i] = 0.0; }
    for (int i = 0; i < n; i++) {                                        /* :1251-1263 ; :1337-1347 */
        if (!assem[i]) continue;
        for (int j = 0; j < n; j++) {
            double v = lhs[IDX(i, j, n)];
            if (assem[j]) { trip_rows[nt] = (int32_t)idx[i]; trip_cols[nt] = (int32_t)idx[j]; trip_vals[nt] = v; nt++; }
            else rhs_vals[i] -= v * dirichlet_data[j];
        }
    }
    for (int i = 0; i < ncd; i++) rhs_vals[i] += rhs[i];                 /* :1265 ; :1349 */
    *ntrip = nt;
    return HHO_OK;
}

size_t cut_interface_cell_offset(const cut_mesh *m, hho_degrees di, size_t c, const int64_t *cell_table, int where)
{
    int cbs = hho_cell_basis_size(di.cell_deg);
    size_t o = (size_t)cell_table[c] * cbs;                              /* :1368-1379 */
    if (m->cell_loc[c] == CUT_ON_INTERFACE && where == CUT_POS) o += cbs;
    return o;
}
