/*
 * cuthho_oracle.h -- CPU restatement (plain C) of the cutHHO part of the hot path:
 * mesh tagging / node displacement / interface refinement (src/methods/cuthho_bits/
 * cuthho_geom.hpp:68-161, 275-340, 466-543, 609-673), cut quadrature (cuthho_geom.hpp:546-895)
 * and the cut local operators of the fictitious-domain driver
 * (apps/cuthho/cuthho_square.cpp:301-388, 566-666).
 *
 * TEST INFRASTRUCTURE, like hho_oracle.h.  Pinned end to end by the energy errors of
 * apps/cuthho/cuthho.xlsx (sheet 1, F.D. table) in tests/test_oracle_cuthho.py.
 */
#ifndef CUTHHO_ORACLE_H
#define CUTHHO_ORACLE_H

#include "hho_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { CUT_NEG = 0, CUT_POS = 1, CUT_ON_INTERFACE = 2, CUT_UNDEF = 3 };   /* element_location, cuthho_mesh.hpp:31-36 */
enum { CUT_LS_CIRCLE = 0, CUT_LS_LINE = 1 };

/* circle_level_set / line_level_set, cuthho_square.cpp:56-124 */
typedef struct { int kind; double radius, alpha, beta; double cut_y; } cut_level_set;
double cut_ls_eval(const cut_level_set *ls, double x, double y);
void cut_ls_normal(const cut_level_set *ls, double x, double y, double n[2]);

typedef struct cut_mesh cut_mesh;

/* cuthho_poly_mesh(mip) (basic_mesh.hpp:321-403: same generator as quad_mesh) */
cut_mesh *cut_mesh_create(size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y);
void cut_mesh_free(cut_mesh *m);
/* cuthho_square.cpp:2036-2052 with node displacement (-D, the default):
 * detect_node_position, detect_cut_faces, move_nodes, detect_cut_faces, detect_cut_cells,
 * refine_interface(refsteps).  Returns 0, or an error code (invalid number of cuts, concave
 * polygon, interface not found). */
int cut_mesh_preprocess(cut_mesh *m, const cut_level_set *ls, int refsteps);

size_t cut_mesh_num_points(const cut_mesh *m);
size_t cut_mesh_num_cells(const cut_mesh *m);
size_t cut_mesh_num_faces(const cut_mesh *m);
const double *cut_mesh_points(const cut_mesh *m);            /* np x 2 (after displacement)     */
const uint64_t *cut_mesh_cell_ptids(const cut_mesh *m);      /* nc x 4                          */
const uint64_t *cut_mesh_faces(const cut_mesh *m);           /* nf x 2                          */
const uint8_t *cut_mesh_face_boundary(const cut_mesh *m);
const int8_t *cut_mesh_node_location(const cut_mesh *m);     /* tags of the UNDISPLACED mesh    */
const int8_t *cut_mesh_face_location(const cut_mesh *m);
const double *cut_mesh_face_intersection(const cut_mesh *m); /* nf x 2                          */
const int8_t *cut_mesh_cell_location(const cut_mesh *m);
size_t cut_mesh_interface_points(const cut_mesh *m);         /* 2^refsteps + 1                  */
const double *cut_mesh_cell_interface(const cut_mesh *m, size_t cell);   /* NULL if not cut     */
size_t cut_mesh_cell_face(const cut_mesh *m, size_t cell, int lf);       /* offset(msh, faces(msh,cl)[lf]) */

/* integrate(msh, cl, degree, where) cuthho_geom.hpp:798-815 (uncut: fan quadrature) */
int cut_cell_quadrature(const cut_mesh *m, size_t cell, int degree, int where, double *qx, double *qy, double *qw, int cap);
/* integrate(msh, fc, degree, where) cuthho_geom.hpp:817-849; face = local face lf of `cell` */
int cut_face_quadrature(const cut_mesh *m, size_t cell, int lf, int degree, int where, double *qx, double *qy, double *qw, int cap);
/* integrate_interface cuthho_geom.hpp:851-895 */
int cut_interface_quadrature(const cut_mesh *m, size_t cell, int degree, int where, double *qx, double *qy, double *qw, int cap);
/* measure(msh, cl, where) cuthho_geom.hpp:779-796 */
double cut_cell_measure(const cut_mesh *m, size_t cell, int where);

#define CUT_MAX_QPS 1024

/* make_hho_laplacian(msh, cl, level_set, di, where) cuthho_square.cpp:308-388.
 * *oper_rows = rbs for cut cells, rbs - 1 otherwise (quirk 8 of SURVEY appendix B). */
int cut_make_hho_laplacian(const cut_mesh *m, const cut_level_set *ls, size_t cell, hho_degrees di, int where,
                           double *oper, double *data, int *oper_rows);
/* make_hho_cut_stabilization cuthho_square.cpp:566-621 */
int cut_make_hho_cut_stabilization(const cut_mesh *m, size_t cell, hho_degrees di, int where, double *stab);
/* make_rhs(msh, cl, degree, f, where, level_set, bcs) cuthho_square.cpp:623-666 */
int cut_make_rhs(const cut_mesh *m, const cut_level_set *ls, size_t cell, int degree, int where,
                 hho_scalar_fn f, hho_scalar_fn bcs, void *user, double *rhs);

#ifdef __cplusplus
}
#endif
#endif
