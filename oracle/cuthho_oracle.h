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

/* the -A branch of cuthho_square.cpp:2039-2044: detect_node_position, detect_cut_faces,
 * detect_cut_cells (no node displacement), refine_interface */
int cut_mesh_preprocess_agglomeration(cut_mesh *m, const cut_level_set *ls, int refsteps);
/* detect_cell_agglo_set cuthho_geom.hpp:163-273 (threshold 0.3): per cell, the reference's
 * cell_agglo_set in its declaration order */
enum { CUT_AGGLO_UNDEF = 0, CUT_AGGLO_T_OK = 1, CUT_AGGLO_T_KO_NEG = 2, CUT_AGGLO_T_KO_POS = 3 };
void cut_mesh_agglo_set(const cut_mesh *m, int8_t *agglo);
/* make_neighbors_info cuthho_geom.hpp:343-370 (all pairs of cells): nc x 8 ids, ascending, -1 padded */
void cut_mesh_neighbors(const cut_mesh *m, int32_t *neighbors);

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

/* ---- two-sided interface problem (`cuthho_square -i`) ------------------------------------- */
typedef struct { double kappa_1, kappa_2, eta; } cut_params;   /* params<T>, cuthho_square.cpp:293-299: 1, 1, 5 */

/* make_hho_laplacian_interface cuthho_square.cpp:390-502 for a CUT cell: unknowns ordered
 * [cell-, cell+, faces-, faces+].  oper 2rbs x 2msize, data 2msize x 2msize (column-major).
 * gr_lhs is symmetric positive SEMI-definite (kernel: the same constant on both sides) and the
 * reference solves with Eigen's LDLT (diagonal pivoting, :498): restated here as textbook LDL^T with
 * symmetric diagonal pivoting and a pseudo-inverse of the rounding-level pivot (max|D| * eps, the
 * tolerance of Eigen <= 3.2; see ldlt_solve in the .c file for why not Eigen 3.3's).  `oper` is
 * therefore the solution with zero component on the last pivot; `data = gr_rhs^T oper` (:499),
 * the only output run_cuthho_interface uses (:1692), does not depend on the kernel component. */
int cut_make_hho_laplacian_interface(const cut_mesh *m, const cut_level_set *ls, size_t cell, hho_degrees di,
                                     const cut_params *parms, double *oper, double *data);
/* make_rhs(msh, cl, degree, where, f) src/methods/cuthho_bits/cuthho_utils.hpp:65-84 */
int cut_make_rhs_side(const cut_mesh *m, size_t cell, int degree, int where, hho_scalar_fn f, void *user, double *rhs);

/* interface_assembler cuthho_square.cpp:1091-1443.  Tables of the constructor (:1137-1185):
 * cell_table[c] = first unknown block of cell c (cut cells own two), face_table[f] = first
 * block of non-Dirichlet face f (cut faces own two; -1 for Dirichlet faces). */
void cut_interface_tables(const cut_mesh *m, int64_t *cell_table, int64_t *face_table,
                          size_t *num_all_cells, size_t *num_other_faces);
/* assemble (:1203-1269, uncut cell, lhs msize^2) / assemble_cut (:1271-1354, cut cell, lhs
 * (2 msize)^2, rhs 2cbs): triplets in push order and per-local-row right-hand-side updates, like
 * hho_assembler_assemble_cell.  dirichlet_data: msize values (uncut only). */
int cut_interface_assemble(const cut_mesh *m, hho_degrees di, size_t cell, const int64_t *cell_table,
                           const int64_t *face_table, size_t num_all_cells,
                           const double *lhs, const double *rhs, const double *dirichlet_data,
                           int32_t *trip_rows, int32_t *trip_cols, double *trip_vals, size_t *ntrip,
                           int64_t *rhs_rows, double *rhs_vals);
/* cell unknowns read back by take_local_data (:1356-1379): offset of the cell block of `where` */
size_t cut_interface_cell_offset(const cut_mesh *m, hho_degrees di, size_t cell, const int64_t *cell_table, int where);

/* ---- cut_truth.c: the same cut operators in IEEE binary128 (__float128) from the double-valued geometry and
 * quadrature lists above; the side sliver cells are judged against (cond * eps separates any two double evaluations).
 * Outputs are rounded to double at the very end.  Cut cells only (HHO_ERR_ARG otherwise). */
int cut_truth_laplacian(const cut_mesh *m, const cut_level_set *ls, size_t cell, hho_degrees di, int where,
                        double *oper, double *data);                     /* oper rbs x msize */
int cut_truth_stabilization(const cut_mesh *m, size_t cell, hho_degrees di, int where, double *stab);
int cut_truth_rhs(const cut_mesh *m, const cut_level_set *ls, size_t cell, int degree, int where, int f_id, int bcs_id, double *rhs);
/* data = gr_rhs^T gr_lhs^+ gr_rhs; oper (may be NULL) = the solution with the constant of the negative side pinned to 0 */
int cut_truth_laplacian_interface(const cut_mesh *m, const cut_level_set *ls, size_t cell, hho_degrees di,
                                  const cut_params *parms, double *oper, double *data);
double cut_truth_last_interface_cond(void);   /* 1-norm condition number of the pinned system of the last call above */
int cut_truth_rhs_side(const cut_mesh *m, size_t cell, int degree, int where, int f_id, double *rhs);
/* error attribution: 1 = gr_lhs / gr_rhs of cut_truth_laplacian rounded to double once formed (the solve stays binary128) */
void cut_truth_round_inputs(int on);
/* 1-norm condition number of the cut cell's rbs x rbs reconstruction system (exact inverse in binary128) */
double cut_truth_laplacian_cond(const cut_mesh *m, const cut_level_set *ls, size_t cell, hho_degrees di, int where);

#ifdef __cplusplus
}
#endif
#endif
