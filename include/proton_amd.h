/*
 * proton_amd.h -- C ABI of the MI355X-native HHO local-operator assembly path.
 *
 * The reference (OmarDuran/ProtoN) is a header-only C++ template library with no
 * FFI; its interface for this path is a set of per-cell function templates.  This
 * ABI is the batched, device-side equivalent that a binding of those templates
 * calls.  Each entry point cites the reference interface it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only; no
 * C++/torch types; no exceptions cross this boundary (int status codes).
 *
 * Pointers named d_* are DEVICE pointers (hipMalloc'd or e.g. torch tensors'
 * data_ptr()); everything else is host memory.  All matrices are column-major
 * (Eigen's default, which the reference returns), batched cell-major:
 * entry (row i, col j) of cell c lives at  base[c * rows*cols + j*rows + i].
 *
 * Threading: one context per device/stream; calls on a context are ordered by
 * its HIP stream and are asynchronous with respect to the host unless stated.
 */
#ifndef PROTON_AMD_H
#define PROTON_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_ABI_VERSION 2

/* status codes (reference: C++ exceptions / silent NaNs) */
enum {
    PA_OK = 0,
    PA_ERR_INVALID_ARG = 1,
    PA_ERR_INVALID_DEGREE = 2,     /* hho_degree_info pair not instantiated                       */
    PA_ERR_QUADRATURE = 3,         /* "Quadrature order too high" quadratures.hpp:245-246, incl. the
                                      rules[8] hole of the fan quadrature (quadratures_dunavant.hpp:129) */
    PA_ERR_HIP = 4,                /* a HIP runtime call failed; see pa_last_error()               */
    PA_ERR_NO_MESH = 5,
    PA_ERR_NOT_SPD = 6,            /* some cell had a non-positive Cholesky pivot (Eigen LLT: silent) */
    PA_ERR_COMM = 7                /* RCCL missing or an RCCL call failed; see pa_comm_last_error()  */
};

/* integrate() overloads: quadratures.hpp:311-375 (quad_mesh, tensor Gauss) and
 * quadratures.hpp:377-402 (poly_mesh, 4-triangle fan + Dunavant). */
enum { PA_QUAD_TENSOR = 0, PA_QUAD_FAN = 1 };
/* stabilization choice: none / make_hho_naive_stabilization hho.hpp:99-148 /
 * make_hho_fancy_stabilization hho.hpp:155-237 */
enum { PA_STAB_NONE = 0, PA_STAB_NAIVE = 1, PA_STAB_FANCY = 2 };
/* built-in source terms evaluated on the device (the reference passes a C++ lambda) */
enum {
    PA_FN_SAMPLED = 0,             /* values supplied per quadrature point by the caller            */
    PA_FN_SIN_SIN_RHS = 1,         /* 2 pi^2 sin(pi x) sin(pi y)   convergence_test.cpp:100-102,
                                      cuthho_square.cpp:846-848                                     */
    PA_FN_SIN_SIN_SOL = 2,         /* sin(pi x) sin(pi y)          convergence_test.cpp:104-106     */
    PA_FN_OBSTACLE_RHS = 3,        /* obstacle.cpp:67-74  (r0 = 0.7)                                */
    PA_FN_OBSTACLE_SOL = 4,        /* obstacle.cpp:76-81                                            */
    PA_FN_ONE = 5
};

typedef struct pa_context pa_context;

/* hho_degree_info (utils.hpp:62-111) as plain ints */
typedef struct { int32_t cell_deg, face_deg, rec_deg; } pa_degree_info;
/* hho_degree_info(size_t) utils.hpp:71-73 */
pa_degree_info pa_degree_info_equal(int degree);
/* hho_degree_info(size_t cd, size_t fd) utils.hpp:75-95; *fell_back (may be NULL) is set
 * when the pair is invalid and the reference "reverts to equal-order" */
pa_degree_info pa_degree_info_make(int cd, int fd, int *fell_back);

/* sizes: cell_basis::size bases.hpp:191-194, face_basis::size bases.hpp:287-290 */
typedef struct {
    int32_t rbs, cbs, fbs, msize;      /* msize = cbs + 4 fbs                                   */
    int32_t oper_rows;                 /* rbs - 1 (hho.hpp:52-53)                               */
    int32_t cell_qps, face_qps;        /* points of integrate(cell, 2 recdeg) / (face, 2 facdeg) */
} pa_sizes;
int pa_sizes_for(pa_degree_info di, int quad_kind, pa_sizes *out);

/* ---- context ------------------------------------------------------------------------ */
/* `stream` is the hipStream_t every call of the context is enqueued on; NULL is HIP's default
 * (null) stream -- which is what torch.cuda.current_stream().cuda_stream reports for torch's
 * default stream.  With own_stream != 0 `stream` is ignored and the context creates (and later
 * destroys) a non-blocking stream of its own. */
int pa_context_create(int device, void *stream, int own_stream, pa_context **out);
int pa_context_destroy(pa_context *ctx);
int pa_context_synchronize(pa_context *ctx);
/* on != 0: pa_cut_local_ops_batch / pa_cut_rhs_sampled_batch run on a side stream of the context, next to whatever
 * is enqueued on its main stream AFTER them -- call them before pa_local_ops_batch so that the few cut cells
 * (one wavefront each, a long serial chain) overlap the uncut cells' kernels (cuthho_square.cpp:883-900 handles
 * both kinds in one loop).  Their outputs are ordered only before pa_cut_merge and pa_context_synchronize.
 * Off by default: everything on the one stream, in call order.  (Pays only when the main stream's kernels leave
 * room: next to the persistent local-operator grid of a 512^2 mesh it measured 8 % slower than the plain sequence.) */
int pa_context_set_cut_overlap(pa_context *ctx, int on);
const char *pa_last_error(pa_context *ctx);      /* text of the last HIP failure, "" if none  */
/* The record buffer of the local-operator calls (see pa_local_ops_batch) grows on demand up to a cap -- 4 GiB, or what
 * pa_context_set_record_cap says (>= 1 MiB; larger ranges of cells run in equal pieces); calls that write local matrices
 * run in pieces of at most 192 Ki cells anyway (135 MB of records at k = 2: a piece's records stay in the Infinity Cache
 * between its two kernels) -- and stays with the
 * context for reuse.  pa_context_trim waits for the context's stream and gives it back to the device. */
int pa_context_set_record_cap(pa_context *ctx, size_t bytes);
int pa_context_trim(pa_context *ctx);
int pa_abi_version(void);

/* device-memory helpers so a non-torch host (the C++ header, a cgo/JNI caller) needs no HIP */
int pa_malloc(pa_context *ctx, size_t bytes, void **d_out);
int pa_free(pa_context *ctx, void *d_ptr);
int pa_memcpy_h2d(pa_context *ctx, void *d_dst, const void *src, size_t bytes);   /* synchronous */
int pa_memcpy_d2h(pa_context *ctx, void *dst, const void *d_src, size_t bytes);   /* synchronous */
int pa_memset(pa_context *ctx, void *d_dst, int value, size_t bytes);

/* ---- mesh ----------------------------------------------------------------------------
 * Replaces the mesh<T,4,...> storage the kernels consume: msh.points (basic_mesh.hpp:222)
 * and cell.ptids (basic_mesh.hpp:51).  points: np x 2 doubles (x,y), cell_ptids: nc x 4
 * uint32 in the reference's CCW order.  The arrays are copied to the device. */
int pa_mesh_upload(pa_context *ctx, const double *points, size_t npoints,
                   const uint32_t *cell_ptids, size_t ncells);
/* Same, but the arrays already live on the device and are borrowed (not copied, not freed). */
int pa_mesh_attach_device(pa_context *ctx, const double *d_points, size_t npoints,
                          const uint32_t *d_cell_ptids, size_t ncells);
/* Structured generator mesh_impl<T,4>(mesh_init_params) basic_mesh.hpp:230-298, built on
 * the device: points (min + i*h), cells {p, p+1, p+Nx+2, p+Nx+1}, cell id = j*Nx + i.
 * row_begin/row_end select the block of cell rows [row_begin,row_end) this context owns
 * (multi-GPU partition); cell index 0 of the context is global cell row_begin*Nx. */
int pa_mesh_generate(pa_context *ctx, size_t Nx, size_t Ny,
                     double min_x, double max_x, double min_y, double max_y,
                     size_t row_begin, size_t row_end);
int pa_mesh_counts(pa_context *ctx, size_t *npoints, size_t *ncells);
/* Replaces the coordinates of a mesh the context owns (pa_mesh_upload / pa_mesh_generate) and keeps its connectivity and
 * numbering: msh.points edited in place (the commented-out node perturbation of convergence_test.cpp:176-187; a general
 * quadrilateral mesh with the generator's closed-form face numbering, whole or as a slab of cell rows -- d_points then holds
 * the slab's node rows row_begin .. row_end).  d_points: npoints x 2 doubles on the device, copied on the context's stream. */
int pa_mesh_set_points(pa_context *ctx, const double *d_points, size_t npoints);

/* ---- the hot path --------------------------------------------------------------------
 * For cells [first, first+n) of the uploaded mesh computes, per cell,
 *   (oper, data) = make_hho_laplacian(msh, cl, di)                 hho.hpp:32-96
 *   stab         = make_hho_{naive,fancy}_stabilization(...)       hho.hpp:99-148 / 155-237
 *   lc           = data + stab                                     convergence_test.cpp:212
 * Any of the d_* outputs may be NULL (not written).  Shapes per cell:
 *   d_oper (rbs-1) x msize, d_data/d_stab/d_lc msize x msize, d_info one int32
 *   (0, or 1+index of the first non-positive Cholesky pivot: 100+ for the cell-mass
 *   factorization of the fancy stabilization).
 * Asynchronous on the context's stream.  Two kernels per call: a one-thread-per-cell pre-pass and the
 * cooperative kernel, which meet in a record buffer the CONTEXT owns (grown on demand, at most 4 GiB --
 * pa_context_set_record_cap -- larger ranges run in equal pieces; released by pa_context_trim / pa_context_destroy).  Calls on one
 * context are serialized by its stream, so the buffer needs no locking; use one context per stream. */
int pa_local_ops_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind,
                       size_t first, size_t n,
                       double *d_oper, double *d_data, double *d_stab, double *d_lc,
                       int32_t *d_info);

/* make_rhs(msh, cl, degree, f, di) utils.hpp:153-174 for every cell in [first, first+n):
 * d_rhs[c][cbs(degree)].  fn = PA_FN_*; with PA_FN_SAMPLED d_fvals holds f at the
 * quadrature points of integrate(cell, 2*(degree+dinc)), n x nqp doubles in the
 * reference's point order (pa_cell_quadrature_points returns the points). */
int pa_cell_rhs_batch(pa_context *ctx, int degree, int dinc, int quad_kind, int fn,
                      const double *d_fvals, size_t first, size_t n, double *d_rhs);
/* integrate(msh, cl, degree) quadratures.hpp:311-402: d_xyw[c][nqp][3] = (x, y, weight) */
int pa_cell_quadrature_points(pa_context *ctx, int degree, int quad_kind,
                              size_t first, size_t n, double *d_xyw, int32_t *nqp_out);

/* Static condensation of the cell unknowns (the reference has none):
 *   S = A_FF - A_FT A_TT^-1 A_TF,  g = -A_FT A_TT^-1 f_T,
 *   rec = [ A_TT^-1 f_T | -A_TT^-1 A_TF ]   (cbs x (1 + 4 fbs)),  u_T = rec[:,0] + rec[:,1:] u_F.
 * d_lc: n x msize^2, d_rhs: n x cbs (may be NULL = 0).  Outputs may be NULL. */
int pa_static_condensation_batch(pa_context *ctx, pa_degree_info di, size_t n,
                                 const double *d_lc, const double *d_rhs,
                                 double *d_S, double *d_g, double *d_rec, int32_t *d_info);
/* Same, with the symmetric Schur complement stored as its upper triangle, column-packed:
 * d_Sp[c][j*(j+1)/2 + i] = S(i,j), i <= j, nf*(nf+1)/2 values per cell (the multi-GPU exchange
 * format: values only, indices closed-form). */
int pa_static_condensation_packed_batch(pa_context *ctx, pa_degree_info di, size_t n,
                                        const double *d_lc, const double *d_rhs,
                                        double *d_Sp, double *d_g, int32_t *d_info);

/* ---- condensed mode: static condensation fused into the local-operator pass ---------------------
 * The north star's hot path ends in the static condensation of the cell unknowns (SURVEY section 8 row
 * A15; the reference has none: its assemblers keep cell AND face unknowns, hho.hpp:331).  In this mode
 * lc = data + stab never reaches HBM: the kernel eliminates the cbs cell unknowns of every cell on chip
 * and writes, per cell, one packed record of  nf (nf + 1) / 2 + nf  doubles (nf = 4 fbs):
 *   the upper triangle of  S = A_FF - A_FT A_TT^-1 A_TF,  column-packed (S(i,j), i <= j, at j(j+1)/2 + i),
 *   then  g = -A_FT A_TT^-1 f_T.
 * d_rhs: f_T = make_rhs (utils.hpp:153-174), n x cbs, or NULL (= 0).  d_cond: n x (nf(nf+1)/2 + nf).
 * d_info as pa_local_ops_batch; 200 + j flags pivot j of A_TT.  stab_kind must not be PA_STAB_NONE
 * (A_TT = data_TT is singular on the constants, hho.hpp:93). */
int pa_condensed_ops_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind,
                           size_t first, size_t n, const double *d_rhs, double *d_cond, int32_t *d_info);
/* The same pass run for the recovery of the eliminated unknowns once the face unknowns are known:
 *   u_T = A_TT^-1 (f_T - A_TF u_F)      d_uF: n x nf (pa_condensed_take_faces), d_uT: n x cbs. */
int pa_condensed_recover_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind,
                               size_t first, size_t n, const double *d_rhs, const double *d_uF,
                               double *d_uT, int32_t *d_info);

/* The face-only global system the condensed records assemble into: the reference's numbering
 * (hho.hpp:305-331, 362-379) without its cbs * ncells block of cell unknowns -- unknown k of
 * non-Dirichlet face F sits at compress_table[F] * fbs + k. */
typedef struct {
    uint64_t system_size;        /* fbs * num_other_faces                                            */
    uint64_t num_other_faces;
    int32_t nf, cond_doubles;    /* 4 fbs; nf (nf + 1) / 2 + nf                                     */
    /* row partition: the context OWNS the rows of the faces of its cell rows' blocks (for every cell
     * row its bottom and vertical faces; the horizontals closing the slab on top belong to the slab
     * above): the contiguous row range [row_begin, row_end) of the global system */
    uint64_t row_begin, row_end;
    uint64_t nnz_owned;          /* stored entries of the owned rows                                 */
    uint64_t halo_cells;         /* cells whose top face belongs to the slab above (Nx, or 0 for the
                                    topmost slab): pa_condensed_halo_pack writes that many rows      */
    int32_t halo_doubles;        /* doubles per halo row: fbs * (nf + 1)                             */
    int32_t has_below;           /* != 0: the slab's bottom faces also receive the contribution of the
                                    slab below: pa_condensed_csr_fill expects d_halo_below            */
} pa_condensed_info;
int pa_condensed_query(pa_context *ctx, pa_degree_info di, pa_condensed_info *out);
/* The row partition alone, in closed form, for the slab [row_begin, row_end) of the Nx x Ny generator mesh: what
 * pa_condensed_query reports for such a context, without a context or a device (nnz_owned is left 0).  Lets a driver
 * size its buffers and its exchange before any rank has touched its GPU. */
int pa_condensed_partition_info(size_t Nx, size_t Ny, size_t row_begin, size_t row_end, pa_degree_info di,
                                pa_condensed_info *out);

/* assembler::assemble (hho.hpp:344-406) on the condensed blocks, cells [first, first+n): per cell nf^2
 * triplet slots in the reference's push order restricted to the face unknowns (slot i*nf + j for face
 * rows i, j of the local matrix), row = col = -1 where either face is Dirichlet; d_rhs_rows / d_rhs_vals
 * n x nf: g_i minus the Dirichlet columns times the boundary data (hho.hpp:401).  d_g from
 * pa_dirichlet_data_batch or NULL. */
int pa_condensed_triplets_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                                const double *d_cond, const double *d_g,
                                int32_t *d_rows, int32_t *d_cols, double *d_vals,
                                int32_t *d_rhs_rows, double *d_rhs_vals);
/* The same system built directly in CSR from the mesh's face adjacency -- no triplets, no sort.
 * Symbolic phase, once per mesh and degree: d_rowptr (row_end - row_begin + 1 entries, int64, starting
 * at 0) and d_colind (nnz_owned global column ids, ascending within a row). */
int pa_condensed_csr_pattern(pa_context *ctx, pa_degree_info di, int64_t *d_rowptr, int32_t *d_colind);
/* Numeric phase, once per assembly: d_values (nnz_owned) and d_rhs (row_end - row_begin) from the
 * records of ALL the context's cells (d_cond: ncells x cond_doubles).  Contributions of the two cells of
 * a face are added lower cell id first, the order setFromTriplets sums the triplets above in: values are
 * bit-identical to pa_csr_from_triplets of pa_condensed_triplets_batch.  d_halo_below: the rows packed by
 * the slab below (has_below), else NULL. */
int pa_condensed_csr_fill(pa_context *ctx, pa_degree_info di, const double *d_cond, const double *d_g,
                          const double *d_halo_below, double *d_values, double *d_rhs);
/* What the slab above needs of this slab: for every cell of the top cell row the fbs rows of its top
 * face -- fbs x nf values S(2 fbs + k, :) with Dirichlet columns already moved to the right-hand side,
 * then the fbs right-hand-side contributions; halo_cells x halo_doubles doubles.  These rows are the
 * whole multi-GPU exchange of an assembly step (pa_comm_* below). */
int pa_condensed_halo_pack(pa_context *ctx, pa_degree_info di, const double *d_cond, const double *d_g, double *d_halo);
/* take_local_data (hho.hpp:408-449) restricted to the faces: d_uF n x nf from the face-only solution
 * (system_size values), Dirichlet faces from d_g. */
int pa_condensed_take_faces(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                            const double *d_solution, const double *d_g, double *d_uF);
/* The reference's full solution vector (cells first, then compressed faces; hho.hpp:331) from the
 * recovered cell unknowns (ncells x cbs, the context's cells) and the face-only solution: what
 * assembler::take_local_data and the postprocessing expect.  d_full: cbs * ncells_global + system_size;
 * a slab writes its own cells and (d_xF != NULL) the face part. */
int pa_condensed_expand_solution(pa_context *ctx, pa_degree_info di, const double *d_uT, const double *d_xF,
                                 double *d_full);

/* ---- assembler<Mesh> (hho.hpp:252-463) ---------------------------------------------------
 * Face connectivity.  pa_mesh_generate builds it in closed form; for an uploaded mesh supply
 * msh.faces as the reference holds them (basic_mesh.hpp:114-137, sorted by (lo,hi) point ids):
 * cell_faces[c][4] = offset(msh, fc) of faces(msh, cl) (basic_geom.hpp:183-212, order bottom,
 * right, top, left), face_pts[f][2] = fc.ptids (lo, hi), face_is_dirichlet[f] =
 * fc.is_boundary && fc.bndtype == DIRICHLET.  Host arrays; copied. */
int pa_mesh_set_faces(pa_context *ctx, const uint32_t *cell_faces, const uint32_t *face_pts,
                      const uint8_t *face_is_dirichlet, size_t nfaces);

typedef struct {
    uint64_t system_size;        /* cbs * ncells + fbs * num_other_faces        hho.hpp:331      */
    uint64_t ncells_global;      /* cells of the whole mesh                                       */
    uint64_t cell_base;          /* global id of the context's cell 0 (row partition)             */
    uint64_t nfaces_local;       /* rows of the context's face tables (d_g has nfaces_local x fbs) */
    uint64_t face_base;          /* global id of local face 0                                      */
    uint64_t num_other_faces;    /* non-Dirichlet faces of the whole mesh       hho.hpp:305-307   */
} pa_assembler_info;
int pa_assembler_query(pa_context *ctx, pa_degree_info di, pa_assembler_info *out);

/* Dirichlet data of every boundary face, mass.llt().solve(rhs) of the boundary function
 * (hho.hpp:381-386): d_g[f][fbs], zeros for non-Dirichlet faces.  fn = PA_FN_*; with
 * PA_FN_SAMPLED d_fvals holds the function at the face quadrature points (nfaces_local x
 * (face_deg+1), the points come from pa_face_quadrature_points). */
int pa_dirichlet_data_batch(pa_context *ctx, int face_deg, int fn, const double *d_fvals, double *d_g);
/* integrate(msh, fc, 2*face_deg) quadratures.hpp:404-432: d_xyw[f][face_deg+1][3]; face_deg <= 7: closed-form rules to
 * five nodes, golub_welsch's (quadratures.hpp:32-75, ascending nodes) to eight
 * (pass face_deg + di for the points of a degree-increased rule, utils.hpp:185) */
int pa_face_quadrature_points(pa_context *ctx, int face_deg, double *d_xyw);

/* assembler::assemble (hho.hpp:344-406) for cells [first, first+n): per cell msize^2 triplet
 * slots in the reference's push order (slot i*msize + j for local row i, column j); a slot whose
 * row or column belongs to a Dirichlet face holds row = col = -1.  Row/column indices are the
 * reference's int triplet indices (Eigen::Triplet<T>).  The right-hand-side updates of
 * hho.hpp:401,405 are returned per local row: d_rhs_rows[c][msize] (global row or -1) and
 * d_rhs_vals[c][msize]; RHS = scatter-add of those.  d_lc: n x msize^2, d_rhs: n x cbs or NULL,
 * d_g from pa_dirichlet_data_batch or NULL (homogeneous). */
int pa_triplets_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                      const double *d_lc, const double *d_rhs, const double *d_g,
                      int32_t *d_rows, int32_t *d_cols, double *d_vals,
                      int32_t *d_rhs_rows, double *d_rhs_vals);

/* The same system -- assembler<Mesh>'s own numbering, cell AND face unknowns (system_size = cbs * ncells + fbs *
 * num_other_faces, hho.hpp:331), i.e. what assemble (hho.hpp:344-406) + finalize = setFromTriplets (hho.hpp:451-455)
 * leave in `LHS` / `RHS` -- built DIRECTLY in CSR from the mesh's face adjacency: no triplets, no sort.  The span the
 * reference's drivers print as "Matrix assembly" (apps/cuthho/cuthho_square.cpp:881-905,
 * apps/convergence_test/convergence_test.cpp:201-217) is pa_local_ops_batch + pa_cell_rhs_batch + pa_assembler_csr_fill.
 * Whole-mesh contexts only (a slab of a partitioned mesh assembles the face-only system: pa_condensed_*).
 * Symbolic phase, once per mesh and degree: nrows = system_size, nnz; d_rowptr nrows + 1 (int64), d_colind nnz (int32,
 * ascending within a row; may be NULL).  Numeric phase, once per assembly: d_values nnz, d_RHS nrows (may be NULL) from
 * d_lc (ncells x msize^2), d_rhs (ncells x cbs or NULL), d_g (pa_dirichlet_data_batch or NULL).  Structure and values
 * bit-identical to pa_csr_from_triplets of pa_triplets_batch, d_RHS to the scatter-add of its d_rhs_vals in cell order. */
typedef struct { uint64_t nrows, nnz; } pa_assembler_csr_info;
int pa_assembler_csr_query(pa_context *ctx, pa_degree_info di, pa_assembler_csr_info *out);
int pa_assembler_csr_pattern(pa_context *ctx, pa_degree_info di, int64_t *d_rowptr, int32_t *d_colind);
int pa_assembler_csr_fill(pa_context *ctx, pa_degree_info di, const double *d_lc, const double *d_rhs, const double *d_g,
                          double *d_values, double *d_RHS);

/* assembler::take_local_data (hho.hpp:408-449) for cells [first, first+n): d_out n x msize =
 * the cell's dofs of `d_solution` (system_size values), Dirichlet faces filled from d_g
 * (pa_dirichlet_data_batch; NULL = homogeneous). */
int pa_take_local_data_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                             const double *d_solution, const double *d_g, double *d_out);

/* project_function(msh, cl, hdi, f, di) (utils.hpp:199-227) for cells [first, first+n):
 * d_out n x msize = L2 projection of f on the cell basis (mass and rhs at degree
 * 2*(cell_deg + dinc)) followed by the projections on the four faces (degree
 * 2*(face_deg + dinc)).  fn = PA_FN_*; with PA_FN_SAMPLED d_cell_fvals holds f at the points of
 * pa_cell_quadrature_points(2*(cell_deg+dinc)) for those cells and d_face_fvals at the points
 * of pa_face_quadrature_points(face_deg + dinc) for every face of the context. d_info (n, may be
 * NULL): first non-positive pivot of the cell mass matrix (+1), 0 if SPD. */
int pa_project_function_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int dinc, int fn,
                              const double *d_cell_fvals, const double *d_face_fvals,
                              size_t first, size_t n, double *d_out, int32_t *d_info);
/* d_out[c] = (u_c - v_c)^T lc_c (u_c - v_c) for n cells (the energy-error summand of
 * convergence_test.cpp:283-300 / obstacle.cpp:202-213); d_v may be NULL (v = 0). */
int pa_energy_form_batch(pa_context *ctx, pa_degree_info di, size_t n, const double *d_lc, const double *d_u,
                         const double *d_v, double *d_out);

/* ---- obstacle_assembler<Mesh> (hho.hpp:471-751) ---------------------------------------------
 * These entry points need the whole mesh on the context (no row slab).
 * Constructor tables (hho.hpp:538-578): d_in_A ncells flags (is_in_set_A), d_A_ct / d_B_ct
 * ncells int32: position among the cells outside / inside the active set, -1 otherwise.
 * *num_I / *num_A (host) receive the two counts. */
int pa_obstacle_tables(pa_context *ctx, const uint8_t *d_in_A, int32_t *d_A_ct, int32_t *d_B_ct,
                       size_t *num_I, size_t *num_A);
/* obstacle_assembler::assemble (hho.hpp:609-695) for cells [first, first+n).  Slot layout:
 * msize^2 + 1 slots per cell, slot i*msize + j = lhs(i,j) in the reference's push order, the last
 * slot the multiplier coupling (row cell*cbs, column num_I*cbs + num_other*fbs + B_ct[cell], 1.0)
 * of active cells (hho.hpp:688-693); rows/cols = -1 for slots the reference does not push.
 * Rows are not compressed (cell rows at cell + i, hho.hpp:631: exact for cbs = 1, the only case
 * obstacle.cpp:51 uses), columns are (hho.hpp:625,645).  d_gamma: one value per cell
 * (hho.hpp:677).  d_rhs_rows / d_rhs_vals: n x msize as in pa_triplets_batch. */
int pa_obstacle_triplets_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                               const double *d_lc, const double *d_rhs, const double *d_g, const double *d_gamma,
                               const uint8_t *d_in_A, const int32_t *d_A_ct, const int32_t *d_B_ct, size_t num_I,
                               int32_t *d_rows, int32_t *d_cols, double *d_vals,
                               int32_t *d_rhs_rows, double *d_rhs_vals);
/* obstacle_assembler::expand_solution (hho.hpp:698-744): d_alpha ncells*cbs + nfaces*fbs (cells,
 * then ALL faces, Dirichlet ones from d_g), d_beta ncells*cbs multipliers. */
int pa_obstacle_expand_solution(pa_context *ctx, pa_degree_info di, const double *d_solution, const double *d_g,
                                const double *d_gamma, const uint8_t *d_in_A, const int32_t *d_A_ct,
                                const int32_t *d_B_ct, size_t num_I, double *d_alpha, double *d_beta);
/* free take_local_data(msh, cl, di, expanded_solution) (hho.hpp:753-782): d_out n x msize */
int pa_obstacle_take_local_data_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                                      const double *d_expanded, double *d_out);

/* SparseMatrix::setFromTriplets (hho.hpp:451-455, :746-750; cuthho_square.cpp:1437-1441) on the
 * device: nslots triplet slots (a negative row or column = a slot the assembler did not push) ->
 * CSR with duplicates summed in push order.  d_rowptr nrows+1 (int64), d_colind / d_values with room
 * for nslots entries (the worst case); *nnz (host) receives the number of stored entries.
 * Column indices are ascending within a row.  nslots < 2^31. */
int pa_csr_from_triplets(pa_context *ctx, size_t nslots, const int32_t *d_rows, const int32_t *d_cols, const double *d_vals,
                         size_t nrows, int64_t *d_rowptr, int32_t *d_colind, double *d_values, size_t *nnz);

/* conjugated_gradient(A, b, x, cg_params) (src/core/core_bits/solver_cg.hpp:45-144; the solver of
 * run_cuthho_interface, cuthho_square.cpp:1737-1743) on the CSR matrix of pa_csr_from_triplets.  Same
 * recurrences and exit tests, in the reference's order: *exit_reason = 0 converged (relative residual
 * below convergence_threshold), 2 max_iter reached (iter > max_iter), 1 diverged (relative residual above
 * divergence_threshold).  apply_preconditioner: Jacobi.  x starts from zero (solver_cg.hpp:73).  The dot
 * products are tree reductions, not Eigen's sequential sums: iterates agree to rounding, not bitwise. */
int pa_conjugated_gradient(pa_context *ctx, size_t nrows, const int64_t *d_rowptr, const int32_t *d_colind,
                           const double *d_values, const double *d_b, double *d_x,
                           double convergence_threshold, double divergence_threshold, size_t max_iter,
                           int apply_preconditioner, int32_t *exit_reason, size_t *iterations, double *relative_residual);

/* ---- cutHHO (fictitious domain, `cuthho_square -f`) -----------------------------------------
 * circle_level_set / line_level_set, apps/cuthho/cuthho_square.cpp:56-124 */
typedef struct { int32_t kind; double radius, alpha, beta, cut_y; } pa_level_set;   /* kind 0 circle, 1 line */
enum { PA_LOC_NEGATIVE = 0, PA_LOC_POSITIVE = 1, PA_LOC_ON_INTERFACE = 2 };          /* element_location */

/* Builds the generator mesh (cuthho_poly_mesh, basic_mesh.hpp:321-403) and runs the host
 * preprocessing of cuthho_square.cpp:2036-2052 with node displacement (-D, the default):
 * detect_node_position, detect_cut_faces, move_nodes, detect_cut_faces, detect_cut_cells,
 * refine_interface(refsteps).  The displaced points become the context's mesh.  Errors the
 * reference throws ("invalid number of cuts in cell", "concave poly", "interface not found in
 * search range") are returned as PA_ERR_INVALID_ARG with the text in pa_last_error(). */
int pa_cut_preprocess(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                      const pa_level_set *ls, int refsteps);
/* The same for the slab of cell rows [row_begin, row_end) of a row partition ("cut cells are distributed
 * by the same row rule"): the host preprocessing runs over the WHOLE mesh on every rank (deterministic: every rank tags and
 * displaces the same nodes), the context keeps the slab -- its displaced node rows, the tags of its cells and its own cut
 * cells, in ascending cell order.  Cell index 0 of the context is global cell row_begin * Nx (as pa_mesh_generate);
 * pa_cut_query reports the slab's cells; pa_cut_query_tags still reports the whole mesh.  Everything per cell works on a
 * slab (pa_cut_local_ops_batch, pa_cut_uncut_rhs_batch, pa_cut_merge, pa_cut_merge_condensed, pa_cut_interface_ops_batch,
 * pa_cut_interface_uncut_batch); the interface_assembler's numbering (pa_interface_*) and pa_cut_agglo_query need a
 * whole-mesh context and return PA_ERR_INVALID_ARG on a slab. */
int pa_cut_preprocess_rows(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                           const pa_level_set *ls, int refsteps, size_t row_begin, size_t row_end);
/* The agglomeration branch of the same preprocessing (`-A`, cuthho_square.cpp:2039-2044): no node
 * displacement.  The reference then only CLASSIFIES the cut cells (detect_cell_agglo_set); its
 * agglomerate_cells is dead code (cuthho_square.cpp:1525-1621).  All operators of this library
 * work on the resulting mesh as on the displaced one. */
int pa_cut_preprocess_agglomeration(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y,
                                    double max_y, const pa_level_set *ls, int refsteps);
/* detect_cell_agglo_set (cuthho_geom.hpp:163-273, threshold 0.3): agglo_set ncells (host),
 * 0 UNDEF, 1 T_OK, 2 T_KO_NEG, 3 T_KO_POS (cuthho_mesh.hpp's cell_agglo_set order), and
 * make_neighbors_info (cuthho_geom.hpp:343-370; O(cells^2) there, closed form here): neighbors
 * ncells x 8 cell ids sharing a point, ascending, -1 padded.  Either may be NULL. */
int pa_cut_agglo_query(pa_context *ctx, int8_t *agglo_set, int32_t *neighbors);
/* cell tags (element_location per cell) and, for cut cells, their index in the cut-cell batch
 * (-1 otherwise).  Host arrays of ncells entries; either may be NULL.  *ncut may be NULL. */
int pa_cut_query(pa_context *ctx, size_t *ncut, int8_t *cell_location, int32_t *cut_index);
/* The cut cells' local operators (cuthho_square.cpp:308-388, 566-621, 623-666) with
 * hho_degree_info(face_deg + 1, face_deg) (cuthho_square.cpp:871), batched over the ncut cut
 * cells in ascending cell order: d_oper ncut x rbs x msize (cut cells keep the constant mode:
 * rbs rows), d_data/d_stab/d_lc ncut x msize^2, d_rhs ncut x cbs, d_info ncut.  Any may be NULL.
 * `where` = PA_LOC_NEGATIVE or PA_LOC_POSITIVE; rhs_fn / bcs_fn = PA_FN_* built-ins. */
int pa_cut_local_ops_batch(pa_context *ctx, int face_deg, const pa_level_set *ls, int where, int rhs_fn, int bcs_fn,
                           double *d_oper, double *d_data, double *d_stab, double *d_lc, double *d_rhs,
                           int32_t *d_info);
/* Tags and displaced coordinates left by pa_cut_preprocess (host arrays; any may be NULL):
 * node_location npoints (of the UNDISPLACED mesh: detect_node_position is not re-run after
 * move_nodes, cuthho_square.cpp:2036-2049), face_location nfaces, points npoints x 2. */
int pa_cut_query_tags(pa_context *ctx, int8_t *node_location, int8_t *face_location, double *points);
/* Quadrature lists of the cut cells as the cut integrate() overloads produce them
 * (cuthho_geom.hpp:798-815, 851-895), for callers that sample their own functions:
 * which = 0: integrate(msh, cl, 2*recdeg, where); 1: integrate_interface(msh, cl, 2*recdeg, where);
 * 2: integrate_interface(msh, cl, recdeg, where) (the rule cut make_rhs uses, :647).  HOST outputs:
 * h_offsets ncut+1, h_xyw count x 3; pass NULL to query *count. */
int pa_cut_quadrature_points(pa_context *ctx, int face_deg, int where, int which, uint32_t *h_offsets, double *h_xyw,
                             size_t *count);
/* cut make_rhs (:623-666) with caller-sampled functions: d_rhs_vals at the points of list 0,
 * d_bcs_vals at the points of list 2.  d_rhs ncut x cbs. */
int pa_cut_rhs_sampled_batch(pa_context *ctx, int face_deg, const pa_level_set *ls, int where, const double *d_rhs_vals,
                             const double *d_bcs_vals, double *d_rhs);
/* make_rhs(msh, cl, degree, where, f) of the fictitious-domain driver for the UNCUT cells (cuthho_square.cpp:623-631, fan
 * quadrature of degree 2*degree): the cells on the `where` side get utils.hpp:163-171, every other cell -- outside the domain,
 * or cut: pa_cut_merge puts the cut kernel's right-hand side there -- gets zeros without being integrated (at 512 x 512 with
 * the circle of radius 0.35 that is 61 % of the cells).  d_rhs: ncells x cbs(degree); the same values pa_cell_rhs_batch
 * followed by pa_cut_merge's zeroing leaves. */
int pa_cut_uncut_rhs_batch(pa_context *ctx, int degree, int where, int fn, double *d_rhs);
/* Merge for the assembly loop of cuthho_square.cpp:883-900: rows of the cut cells in the
 * cell-major d_lc / d_rhs (all cells, from pa_local_ops_batch(PA_QUAD_FAN, PA_STAB_NAIVE) and
 * pa_cell_rhs_batch) are replaced by the cut operators; the right-hand side of uncut cells
 * outside `where` is zeroed (cuthho_square.cpp:659-664). */
int pa_cut_merge(pa_context *ctx, int face_deg, int where, const double *d_cut_lc, const double *d_cut_rhs,
                 double *d_lc, double *d_rhs);
/* The merge of the condensed mode: the packed records of the cut cells -- pa_static_condensation_packed_batch of their local
 * matrices and right-hand sides (pa_cut_local_ops_batch), d_cut_Sp ncut x nf(nf+1)/2 and d_cut_g ncut x nf -- replace the
 * cut cells' records in d_cond (pa_condensed_ops_batch over all cells with the uncut formulas, n x (nf(nf+1)/2 + nf)). */
int pa_cut_merge_condensed(pa_context *ctx, int face_deg, const double *d_cut_Sp, const double *d_cut_g, double *d_cond);

/* ---- cutHHO two-sided interface problem (`cuthho_square -i`, run_cuthho_interface
 * cuthho_square.cpp:1625-1846).  hho_degree_info(face_deg + 1, face_deg) (:1662). */
typedef struct { double kappa_1, kappa_2, eta; } pa_interface_params;      /* params<T> :293-299: 1, 1, 5 */
/* Cut cells, in ascending cell order.  Unknowns of a cut cell: [cell-, cell+, faces-, faces+]
 * (2*msize).  d_oper ncut x (2rbs x 2msize) and d_data ncut x (2msize)^2: make_hho_laplacian_interface
 * (:390-502); gr_lhs is semi-definite (the reference uses Eigen's pivoted LDLT): `oper` comes back
 * with the constant of the negative side pinned to zero, `data` does not depend on that choice.
 * d_lc ncut x (2msize)^2: data + kappa_1 * make_hho_cut_stabilization(NEGATIVE) + kappa_2 * (POSITIVE)
 * scattered as :1694-1705.  d_rhs ncut x 2cbs: make_rhs(msh, cl, degree, where, f) of both sides
 * (cuthho_utils.hpp:65-84, :1710-1711).  Any output may be NULL. */
int pa_cut_interface_ops_batch(pa_context *ctx, int face_deg, const pa_level_set *ls, const pa_interface_params *parms,
                               int rhs_fn, double *d_oper, double *d_data, double *d_lc, double *d_rhs, int32_t *d_info);
/* Uncut cells (:1668-1681): d_lc ncells x msize^2 = kappa(side) * make_hho_laplacian.second +
 * make_hho_naive_stabilization (fan quadrature), d_rhs ncells x cbs = make_rhs(f).  Rows of cut
 * cells hold the uncut formulas too and are ignored by pa_interface_triplets_batch. */
int pa_cut_interface_uncut_batch(pa_context *ctx, int face_deg, const pa_interface_params *parms, int rhs_fn,
                                 double *d_lc, double *d_rhs, int32_t *d_info);
/* interface_assembler (:1091-1443): sizes of the system with duplicated unknowns */
typedef struct {
    uint64_t num_all_cells;      /* cell blocks, cut cells counted twice              :1142-1150 */
    uint64_t num_other_faces;    /* non-Dirichlet face blocks, cut faces counted twice :1152-1163 */
    uint64_t system_size;        /* cbs * num_all_cells + fbs * num_other_faces        :1185      */
    uint64_t ncut;
} pa_interface_info;
int pa_interface_assembler_query(pa_context *ctx, int face_deg, pa_interface_info *out);
/* interface_assembler::assemble (:1203-1269) for the uncut cells and assemble_cut (:1271-1354)
 * for the cut cells.  Uncut: d_rows/d_cols/d_vals ncells x msize^2 (slot i*msize + j; -1 for
 * dropped slots and for every slot of a cut cell), d_rhs_rows/d_rhs_vals ncells x msize.  Cut:
 * d_rows_cut/... ncut x (2msize)^2, d_rhs_rows_cut/d_rhs_vals_cut ncut x 2msize.  d_g: Dirichlet
 * data (pa_dirichlet_data_batch) or NULL. */
int pa_interface_triplets_batch(pa_context *ctx, int face_deg, const double *d_lc, const double *d_rhs, const double *d_g,
                                const double *d_lc_cut, const double *d_rhs_cut,
                                int32_t *d_rows, int32_t *d_cols, double *d_vals,
                                int32_t *d_rows_cut, int32_t *d_cols_cut, double *d_vals_cut,
                                int32_t *d_rhs_rows, double *d_rhs_vals, int32_t *d_rhs_rows_cut, double *d_rhs_vals_cut);
/* cell dofs read back by interface_assembler::take_local_data (:1356-1379): d_offsets ncells x 2 =
 * offset in the solution of the cell block of the negative / positive side (equal for uncut cells) */
int pa_interface_cell_offsets(pa_context *ctx, int face_deg, int64_t *d_offsets);

/* ---- multi-GPU exchange: one process per GPU, RCCL over xGMI ------------
 * The reference is a single process without any communication.  Cells shard by rows (pa_mesh_generate's
 * row_begin / row_end); every rank assembles the CSR rows of the faces it owns (pa_condensed_csr_fill), and the only
 * data a step moves between ranks are the packed top-face rows of a slab's top cell row (pa_condensed_halo_pack),
 * one slab up: rank r -> rank r + 1, halo_cells x halo_doubles doubles.  RCCL is bound at run time (dlopen; a copy
 * the process already holds, e.g. PyTorch's, is used), so the library loads without it.
 * Bootstrap as with NCCL: rank 0 calls pa_comm_unique_id and hands the PA_COMM_ID_BYTES to the other ranks (MPI,
 * a file, torch.distributed's store ...); every rank then calls pa_comm_create with its context.
 * The *_start calls are ordered after everything already enqueued on the context's stream and run on a stream of the
 * communicator, next to whatever the context enqueues afterwards; pa_comm_wait orders the context's stream behind
 * them.  Ranks in slab order: rank r owns the cell rows below those of rank r + 1. */
#define PA_COMM_ID_BYTES 128
typedef struct pa_comm pa_comm;
int pa_comm_unique_id(void *id_out, size_t bytes);
int pa_comm_create(pa_context *ctx, int nranks, int rank, const void *unique_id, pa_comm **out);
int pa_comm_destroy(pa_comm *comm);
int pa_comm_info(pa_comm *comm, int *nranks, int *rank);
const char *pa_comm_last_error(pa_comm *comm);
/* d_send_up: this rank's pa_condensed_halo_pack output (ignored on the last rank); d_recv_below: where the rows of
 * the rank below land (ignored on rank 0); counts in doubles */
int pa_comm_halo_exchange_start(pa_comm *comm, const double *d_send_up, size_t send_count,
                                double *d_recv_below, size_t recv_count);
/* every rank's bytes_per_rank bytes to every rank, in rank order (the north star's all-gather of the face-dof
 * blocks, for a caller that wants the whole system on one device) */
int pa_comm_allgather_start(pa_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank);
/* in-place sum over the ranks (the dot products of a distributed solve) */
int pa_comm_allreduce_sum_start(pa_comm *comm, double *d_buf, size_t count);
int pa_comm_wait(pa_comm *comm);
/* both neighbours at once (the halo of a vector in a row-partitioned solve): my first n_send_lo doubles to rank - 1, my last
 * n_send_hi to rank + 1; rank - 1's message into d_recv_lo, rank + 1's into d_recv_hi.  Counts of a missing neighbour are
 * ignored. */
int pa_comm_neighbour_exchange_start(pa_comm *comm, const double *d_send_lo, size_t n_send_lo, const double *d_send_hi, size_t n_send_hi,
                                     double *d_recv_lo, size_t n_recv_lo, double *d_recv_hi, size_t n_recv_hi);

/* ---- the reference's conjugated_gradient on a ROW-PARTITIONED system -------------------------------------------
 * (solver_cg.hpp:45-144 has one process; this is the solve of the face-only condensed system where it was assembled:
 * every rank holds the CSR rows [row_begin, row_end) it owns, pa_condensed_csr_fill, with GLOBAL column indices.)
 * Same recurrences and exit tests in the same order as pa_conjugated_gradient; the dot products are local partial sums
 * added over the ranks, the search direction's entries a rank's rows read beyond its own range -- the matrix is symmetric
 * and banded by the mesh's row structure, so they belong to ranks r - 1 and r + 1 -- are refreshed once per iteration.
 * The transport is three callbacks; NULL = one rank (then the call equals pa_conjugated_gradient, bit for bit).
 *   allreduce_sum:    sum of n host doubles over the ranks, in place, the same bits on every rank;
 *   halo:             one neighbour exchange of device ranges (pa_comm_neighbour_exchange_start's arguments; complete or
 *                     ordered on `stream` when it returns);
 *   neighbour_counts: at setup, tell the neighbours how many of their entries this rank reads (need_lo below, need_hi
 *                     above) and learn how many of this rank's first / last entries they read (give_lo, give_hi).
 * Callbacks return 0 on success.  d_b, d_x: the rank's own rows (row_end - row_begin doubles); x starts from zero.
 * *transport_status: 0 ok, 1 a callback failed on this rank, 2 this rank's rows read beyond what the neighbours can give (or,
 * without a transport, beyond the own range), 3 ANOTHER rank failed, 4 a HIP error on this rank.  Every exit is collective: a
 * rank's local failure rides as a flag on the next all-reduce (one extra addend: allreduce_sum is called with up to 3 values)
 * and all ranks return from the same reduction -- PA_ERR_COMM (1, 3), PA_ERR_INVALID_ARG (2), PA_ERR_HIP (4) --, none is left
 * inside a collective or a neighbour exchange. */
typedef struct {
    void *user;
    int (*allreduce_sum)(void *user, double *vals, int n);
    int (*halo)(void *user, const double *d_send_lo, size_t n_send_lo, const double *d_send_hi, size_t n_send_hi,
                double *d_recv_lo, size_t n_recv_lo, double *d_recv_hi, size_t n_recv_hi, void *stream);
    int (*neighbour_counts)(void *user, int64_t need_lo, int64_t need_hi, int64_t *give_lo, int64_t *give_hi);
} pa_cg_transport;
int pa_conjugated_gradient_rows(pa_context *ctx, const pa_cg_transport *transport, int64_t row_begin, int64_t row_end,
                                const int64_t *d_rowptr, const int32_t *d_colind, const double *d_values, const double *d_b, double *d_x,
                                double convergence_threshold, double divergence_threshold, size_t max_iter, int apply_preconditioner,
                                int32_t *exit_reason, size_t *iterations, double *relative_residual, int32_t *transport_status);
/* plain copies ordered on the context's stream and complete on return (what a host-staged transport needs) */
int pa_copy_to_host(pa_context *ctx, void *host_dst, const void *d_src, size_t bytes);
int pa_copy_to_device(pa_context *ctx, void *d_dst, const void *host_src, size_t bytes);
/* the RCCL transport of a communicator (user = the communicator) */
int pa_comm_cg_transport(pa_comm *comm, pa_cg_transport *out);

/* occupancy / launch facts of the dominant kernel for the roofline bookkeeping */
typedef struct {
    int32_t lanes_per_cell, cells_per_block, block_threads, lds_bytes_per_block;
    int32_t grid_blocks;
    const char *kernel_name;
} pa_launch_info;
int pa_local_ops_launch_info(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind,
                             size_t n, pa_launch_info *out);
/* the same for the condensed-mode instance (pa_condensed_ops_batch) */
int pa_condensed_launch_info(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind,
                             size_t n, pa_launch_info *out);

#ifdef __cplusplus
}
#endif
#endif
