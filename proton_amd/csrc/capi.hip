// capi.hip -- implementation of the C ABI declared in include/proton_amd.h.
// Thin: argument validation, kernel selection, launches on the context's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/proton_amd.h"
#include "cut_device.hpp"
#include "cut_interface_device.hpp"
#include "condensed.hpp"
#include "assembler_csr.hpp"
#include "cut_host.hpp"
#include "hho_assembly.hpp"
#include "hho_aux.hpp"
#include "hho_launch.hpp"
#include "quad_tables.hpp"

// ---- registry of instantiated kernels (one getter per translation unit, see pa_configs.def) ----
#define PA_CONFIG(cd, fd, q, gmin) \
    extern "C" __attribute__((visibility("hidden"))) const pa::KernelEntry *pa_entries_##cd##_##fd##_##q(int *count);
#include "pa_configs.def"
#undef PA_CONFIG

namespace {

typedef const pa::KernelEntry *(*entries_getter)(int *);
struct ConfigRow { int cd, fd, quad, gmin; entries_getter get; };
const ConfigRow k_configs[] = {
#define PA_CONFIG(cd, fd, q, gmin) {cd, fd, q, gmin, &pa_entries_##cd##_##fd##_##q},
#include "pa_configs.def"
#undef PA_CONFIG
};

const pa::KernelEntry *find_kernel(int cd, int fd, int quad, int stab, int lanes)
{
    for (const ConfigRow &row : k_configs) {
        if (row.cd != cd || row.fd != fd || row.quad != quad) continue;
        int n = 0;
        const pa::KernelEntry *e = row.get(&n);
        for (int i = 0; i < n; ++i)
            if (e[i].stab == stab && e[i].lanes_per_cell == lanes) return &e[i];
    }
    return nullptr;
}

int min_lanes(int cd, int fd, int quad)
{
    for (const ConfigRow &row : k_configs)
        if (row.cd == cd && row.fd == fd && row.quad == quad) return row.gmin;
    return 0;
}

}  // namespace

namespace pa {
hipError_t csr_from_triplets(hipStream_t stream, size_t n, const int32_t *d_rows, const int32_t *d_cols, const double *d_vals,
                             size_t nrows, int64_t *d_rowptr, int32_t *d_colind, double *d_values, size_t *nnz_out);   // csr.hip
}

namespace pa {
hipError_t conjugated_gradient(hipStream_t stream, size_t n, const int64_t *rowptr, const int32_t *colind, const double *values,
                               const double *b, double *x, double convergence_threshold, double divergence_threshold,
                               size_t max_iter, int precond, int *exit_reason, size_t *iterations, double *relative_residual);   // solver.hip
struct CgTransport {
    void *user;
    int (*allreduce_sum)(void *user, double *vals, int n);
    int (*halo)(void *user, const double *send_lo, size_t n_send_lo, const double *send_hi, size_t n_send_hi, double *recv_lo,
                size_t n_recv_lo, double *recv_hi, size_t n_recv_hi, void *stream);
    int (*neighbour_counts)(void *user, int64_t need_lo, int64_t need_hi, int64_t *give_lo, int64_t *give_hi);
};
hipError_t conjugated_gradient_rows(hipStream_t stream, const CgTransport *tp, int64_t row_begin, int64_t row_end, const int64_t *rowptr,
                                    const int32_t *colind, const double *values, const double *b, double *x,
                                    double convergence_threshold, double divergence_threshold, size_t max_iter, int precond,
                                    int *exit_reason, size_t *iterations, double *relative_residual, int *transport_status);   // solver.hip
}

#ifndef PA_PIECE_CELLS
#define PA_PIECE_CELLS ((size_t)192 * 1024)
#endif

struct pa_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    int num_cus = 256;
    pa::QuadTables host_tab;
    pa::QuadTables *d_tab = nullptr;
    // mesh
    double *d_points = nullptr;
    uint32_t *d_ptids = nullptr;
    size_t npoints = 0, ncells = 0;
    bool owns_mesh = false;
    // face connectivity for the assembler
    uint32_t *d_cell_faces = nullptr, *d_face_pts = nullptr;
    uint8_t *d_face_dir = nullptr;
    int32_t *d_face_compress = nullptr;
    size_t nfaces_local = 0, face_base = 0, num_other_faces = 0, ncells_global = 0, cell_base = 0;
    // generator mesh (pa_mesh_generate / pa_cut_preprocess): the slab in closed form
    pa::StructuredMesh sm = {0, 0, 0, 0};
    bool structured = false;
    // condensed (face-only) assembly: face adjacency and the symbolic records of the owned faces, built on first use
    int32_t *d_adj = nullptr;
    pa::CondFace *d_cfaces = nullptr;
    pa::CondFaceLean *d_cfaces_lean = nullptr;
    uint32_t *d_ncols = nullptr, *d_prefix = nullptr;
    bool cond_ready = false;
    uint32_t cond_nown = 0, cond_owned_range = 0;
    int32_t cond_p0 = 0;
    uint64_t cond_total_cols = 0;             // sum of the column-face counts of the owned faces
    // direct CSR of the assembler's own system (assembler_csr.hip): non-Dirichlet faces per cell / cells per face and their prefixes
    uint32_t *d_asm_nfc = nullptr, *d_asm_cprefix = nullptr, *d_asm_nfcell = nullptr, *d_asm_fprefix = nullptr;
    bool asm_ready = false;
    uint64_t asm_cell_faces_total = 0, asm_face_cells_total = 0;
    // cutHHO state (host tags + device copies)
    pa::CutMeshHost *cut = nullptr;
    // device copies of the cut quadrature lists, built once per (face degree, side)
    struct CutListsDev {
        int face_deg = -1, where = -1;
        uint32_t *co = nullptr, *io = nullptr, *ro = nullptr;
        double *cx = nullptr, *ix = nullptr, *rx = nullptr, *fl = nullptr, *fs = nullptr;
        int32_t *flc = nullptr, *fsc = nullptr;
    } cl[2];                                  // one slot per side (PA_LOC_NEGATIVE / PA_LOC_POSITIVE)
    uint32_t *d_cut_cells = nullptr;
    int8_t *d_cell_loc = nullptr, *d_face_loc = nullptr;
    int32_t *d_cut_index = nullptr;
    // interface_assembler tables (cuthho_square.cpp:1137-1185)
    int32_t *d_if_cell_table = nullptr, *d_if_face_table = nullptr;
    size_t if_num_all_cells = 0, if_num_other_faces = 0;
    // scratch of pa_cut_interface_ops_batch ([data | stab- | stab+] of the cut cells), kept between calls
    double *d_if_scratch = nullptr;
    size_t if_scratch_cap = 0;                // doubles
    // records of the per-cell pre-pass (hho_pre.hpp), grown on demand, reused by every local-operator call
    double *d_pre = nullptr;
    size_t pre_capacity = 0;                  // doubles
    size_t pre_cap_bytes = (size_t)4 << 30;   // pa_context_set_record_cap
    // pa_context_set_cut_overlap: the cut-cell kernel runs on a side stream next to the uncut cells' kernels
    hipStream_t side = nullptr;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    bool cut_overlap = false, side_pending = false;
    std::string last_error;
};

#define PA_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return PA_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

static void release_faces(pa_context *ctx)
{
    if (ctx->d_cell_faces) (void)hipFree(ctx->d_cell_faces);
    if (ctx->d_face_pts) (void)hipFree(ctx->d_face_pts);
    if (ctx->d_face_dir) (void)hipFree(ctx->d_face_dir);
    if (ctx->d_face_compress) (void)hipFree(ctx->d_face_compress);
    ctx->d_cell_faces = ctx->d_face_pts = nullptr; ctx->d_face_dir = nullptr; ctx->d_face_compress = nullptr;
    ctx->nfaces_local = ctx->face_base = ctx->num_other_faces = 0;
    if (ctx->d_adj) (void)hipFree(ctx->d_adj);
    if (ctx->d_cfaces) (void)hipFree(ctx->d_cfaces);
    if (ctx->d_cfaces_lean) (void)hipFree(ctx->d_cfaces_lean);
    ctx->d_cfaces_lean = nullptr;
    if (ctx->d_ncols) (void)hipFree(ctx->d_ncols);
    if (ctx->d_prefix) (void)hipFree(ctx->d_prefix);
    ctx->d_adj = nullptr; ctx->d_cfaces = nullptr; ctx->d_ncols = ctx->d_prefix = nullptr;
    ctx->cond_ready = false; ctx->cond_nown = ctx->cond_owned_range = 0; ctx->cond_p0 = 0; ctx->cond_total_cols = 0;
    for (uint32_t **p : {&ctx->d_asm_nfc, &ctx->d_asm_cprefix, &ctx->d_asm_nfcell, &ctx->d_asm_fprefix}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    ctx->asm_ready = false; ctx->asm_cell_faces_total = ctx->asm_face_cells_total = 0;
    ctx->structured = false;
}

static void release_cut_lists(pa_context *ctx, int slot)
{
    auto &c = ctx->cl[slot];
    (void)hipFree(c.co); (void)hipFree(c.io); (void)hipFree(c.ro); (void)hipFree(c.cx); (void)hipFree(c.ix);
    (void)hipFree(c.rx); (void)hipFree(c.fl); (void)hipFree(c.fs); (void)hipFree(c.flc); (void)hipFree(c.fsc);
    c = pa_context::CutListsDev();
}

static void release_cut(pa_context *ctx)
{
    release_cut_lists(ctx, 0);
    release_cut_lists(ctx, 1);
    delete ctx->cut; ctx->cut = nullptr;
    if (ctx->d_cut_cells) (void)hipFree(ctx->d_cut_cells);
    if (ctx->d_cell_loc) (void)hipFree(ctx->d_cell_loc);
    if (ctx->d_face_loc) (void)hipFree(ctx->d_face_loc);
    if (ctx->d_cut_index) (void)hipFree(ctx->d_cut_index);
    if (ctx->d_if_cell_table) (void)hipFree(ctx->d_if_cell_table);
    if (ctx->d_if_face_table) (void)hipFree(ctx->d_if_face_table);
    if (ctx->d_if_scratch) (void)hipFree(ctx->d_if_scratch);
    ctx->d_if_scratch = nullptr; ctx->if_scratch_cap = 0;
    ctx->d_cut_cells = nullptr; ctx->d_cell_loc = nullptr; ctx->d_face_loc = nullptr; ctx->d_cut_index = nullptr;
    ctx->d_if_cell_table = ctx->d_if_face_table = nullptr;
    ctx->if_num_all_cells = ctx->if_num_other_faces = 0;
}

static void release_mesh(pa_context *ctx)
{
    release_cut(ctx);
    release_faces(ctx);
    if (ctx->owns_mesh) {
        if (ctx->d_points) (void)hipFree(ctx->d_points);
        if (ctx->d_ptids) (void)hipFree(ctx->d_ptids);
    }
    ctx->d_points = nullptr; ctx->d_ptids = nullptr; ctx->npoints = ctx->ncells = 0; ctx->owns_mesh = false;
}

template <typename T>
static hipError_t upload_vec(const std::vector<T> &v, T **d, hipStream_t s)
{
    hipError_t e = hipMalloc((void **)d, (v.size() ? v.size() : 1) * sizeof(T));
    if (e == hipSuccess && !v.empty()) e = hipMemcpyAsync(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
}

// accessors for comm.hip (internal to the library)
extern "C" __attribute__((visibility("hidden"))) void *pa_context_stream_(pa_context *ctx) { return (void *)ctx->stream; }
extern "C" __attribute__((visibility("hidden"))) int pa_context_device_(pa_context *ctx) { return ctx->device; }

extern "C" {

int pa_abi_version(void) { return PA_ABI_VERSION; }

pa_degree_info pa_degree_info_equal(int degree)
{
    pa_degree_info d = {degree, degree, degree + 1};
    return d;
}

pa_degree_info pa_degree_info_make(int cd, int fd, int *fell_back)
{
    // utils.hpp:75-95
    const bool c1 = fd > 0 && (cd == fd - 1 || cd == fd || cd == fd + 1);
    const bool c2 = fd == 0 && (cd == fd || cd == fd + 1);
    pa_degree_info d;
    if (c1 || c2) { d.cell_deg = cd; d.face_deg = fd; d.rec_deg = fd + 1; }
    else { d.cell_deg = fd; d.face_deg = fd; d.rec_deg = fd + 1; }     // "Reverting to equal-order"
    if (fell_back) *fell_back = !(c1 || c2);
    return d;
}

int pa_sizes_for(pa_degree_info di, int quad_kind, pa_sizes *out)
{
    if (!out || di.cell_deg < 0 || di.face_deg < 0 || di.rec_deg != di.face_deg + 1) return PA_ERR_INVALID_ARG;
    if (quad_kind != PA_QUAD_TENSOR && quad_kind != PA_QUAD_FAN) return PA_ERR_INVALID_ARG;
    out->rbs = pa::P2(di.rec_deg);
    out->cbs = pa::P2(di.cell_deg);
    out->fbs = di.face_deg + 1;
    out->msize = out->cbs + 4 * out->fbs;
    out->oper_rows = out->rbs - 1;
    const int qdeg = 2 * di.rec_deg;
    if (quad_kind == PA_QUAD_TENSOR) {
        const int n = pa::gauss_nodes(qdeg);
        if (n > 5) return PA_ERR_QUADRATURE;          // the local-operator kernels are instantiated for recdeg <= 4 (k <= 3: at most 5 nodes)
        out->cell_qps = n * n;
    } else {
        // quadratures.hpp:245-246 throws above 8; degree 8 itself selects the empty rules[8]
        // (quadratures_dunavant.hpp:129): zero points, singular gr_lhs, NaNs in the reference
        if (qdeg > 8 || pa::dunavant_points(qdeg) == 0) return PA_ERR_QUADRATURE;
        out->cell_qps = 4 * pa::dunavant_points(qdeg);
    }
    out->face_qps = pa::gauss_nodes(2 * di.face_deg);
    return PA_OK;
}

int pa_context_create(int device, void *stream, int own_stream, pa_context **out)
{
    if (!out) return PA_ERR_INVALID_ARG;
    *out = nullptr;
    pa_context *ctx = new (std::nothrow) pa_context();
    if (!ctx) return PA_ERR_INVALID_ARG;
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        if (!own_stream) { ctx->stream = (hipStream_t)stream; ctx->owns_stream = false; }
        else { e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking); ctx->owns_stream = true; }
    }
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
        pa::fill_gauss(ctx->host_tab);
        pa::fill_dunavant(ctx->host_tab);
        pa::fill_face_tables(ctx->host_tab);
        e = hipMalloc((void **)&ctx->d_tab, sizeof(pa::QuadTables));
    }
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tab, &ctx->host_tab, sizeof(pa::QuadTables), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        std::fprintf(stderr, "proton_amd: pa_context_create failed: %s\n", hipGetErrorString(e));
        if (ctx->d_tab) (void)hipFree(ctx->d_tab);
        if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return PA_ERR_HIP;
    }
    *out = ctx;
    return PA_OK;
}

int pa_context_destroy(pa_context *ctx)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);
    release_mesh(ctx);
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->d_pre) (void)hipFree(ctx->d_pre);
    if (ctx->ev_main) (void)hipEventDestroy(ctx->ev_main);
    if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PA_OK;
}

int pa_context_synchronize(pa_context *ctx)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->side) { PA_HIP(ctx, hipStreamSynchronize(ctx->side)); ctx->side_pending = false; }
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

int pa_context_trim(pa_context *ctx)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_pre) (void)hipFree(ctx->d_pre);
    ctx->d_pre = nullptr; ctx->pre_capacity = 0;
    return PA_OK;
}

int pa_context_set_record_cap(pa_context *ctx, size_t bytes)
{
    if (!ctx || bytes < ((size_t)1 << 20)) return PA_ERR_INVALID_ARG;
    ctx->pre_cap_bytes = bytes;
    return PA_OK;
}

int pa_context_set_cut_overlap(pa_context *ctx, int on)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipSetDevice(ctx->device);
    if (on && !ctx->side) {
        PA_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        PA_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
        PA_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
    }
    if (!on && ctx->side_pending) {                      // join what is still out on the side stream
        PA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        ctx->side_pending = false;
    }
    ctx->cut_overlap = on != 0;
    return PA_OK;
}

const char *pa_last_error(pa_context *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int pa_malloc(pa_context *ctx, size_t bytes, void **d_out)
{
    if (!ctx || !d_out) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    PA_HIP(ctx, hipSetDevice(ctx->device));
    PA_HIP(ctx, hipMalloc(d_out, bytes ? bytes : 1));
    return PA_OK;
}

int pa_free(pa_context *ctx, void *d_ptr)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (d_ptr) PA_HIP(ctx, hipFree(d_ptr));
    return PA_OK;
}

int pa_memcpy_h2d(pa_context *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || (!d_dst && bytes) || (!src && bytes)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    PA_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

int pa_memcpy_d2h(pa_context *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || (!dst && bytes) || (!d_src && bytes)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    PA_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

int pa_memset(pa_context *ctx, void *d_dst, int value, size_t bytes)
{
    if (!ctx || (!d_dst && bytes)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    PA_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return PA_OK;
}

// ---- mesh ---------------------------------------------------------------------------------
int pa_mesh_upload(pa_context *ctx, const double *points, size_t npoints, const uint32_t *cell_ptids, size_t ncells)
{
    if (!ctx || !points || !cell_ptids || npoints == 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    for (size_t i = 0; i < 4 * ncells; ++i)
        if (cell_ptids[i] >= npoints) return PA_ERR_INVALID_ARG;      // the kernels gather points[ptid] unchecked
    PA_HIP(ctx, hipSetDevice(ctx->device));
    release_mesh(ctx);
    ctx->owns_mesh = true;
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_points, npoints * 2 * sizeof(double)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_ptids, (ncells ? ncells : 1) * 4 * sizeof(uint32_t)));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_points, points, npoints * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_ptids, cell_ptids, ncells * 4 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->npoints = npoints; ctx->ncells = ncells;
    ctx->ncells_global = ncells; ctx->cell_base = 0;
    return PA_OK;
}

int pa_mesh_attach_device(pa_context *ctx, const double *d_points, size_t npoints, const uint32_t *d_cell_ptids, size_t ncells)
{
    if (!ctx || !d_points || !d_cell_ptids || npoints == 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    release_mesh(ctx);
    ctx->d_points = const_cast<double *>(d_points);
    ctx->d_ptids = const_cast<uint32_t *>(d_cell_ptids);
    ctx->npoints = npoints; ctx->ncells = ncells; ctx->owns_mesh = false;
    ctx->ncells_global = ncells; ctx->cell_base = 0;
    return PA_OK;
}

int pa_mesh_generate(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                     size_t row_begin, size_t row_end)
{
    if (!ctx || Nx == 0 || Ny == 0 || row_begin >= row_end || row_end > Ny) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const size_t rows = row_end - row_begin;
    const size_t np = (Nx + 1) * (rows + 1), nc = Nx * rows;
    if (np >= ((size_t)1 << 32)) return PA_ERR_INVALID_ARG;
    PA_HIP(ctx, hipSetDevice(ctx->device));
    release_mesh(ctx);
    ctx->owns_mesh = true;
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_points, np * 2 * sizeof(double)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_ptids, nc * 4 * sizeof(uint32_t)));
    const double hx = (max_x - min_x) / (double)Nx, hy = (max_y - min_y) / (double)Ny;    // basic_mesh.hpp:190-196
    const int block = 256;
    const int grid = (int)((np + block - 1) / block < 65535 ? (np + block - 1) / block : 65535);
    hipLaunchKernelGGL(pa::mesh_generate_kernel, dim3(grid), dim3(block), 0, ctx->stream, ctx->d_points, ctx->d_ptids,
                       Nx, row_begin, row_end, min_x, hx, min_y, hy);
    PA_HIP(ctx, hipGetLastError());
    ctx->npoints = np; ctx->ncells = nc;
    ctx->ncells_global = Nx * Ny; ctx->cell_base = row_begin * Nx;
    // face connectivity in closed form (basic_mesh.hpp:266-297)
    pa::StructuredMesh sm = {(uint32_t)Nx, (uint32_t)Ny, (uint32_t)row_begin, (uint32_t)row_end};
    const uint32_t nfl = pa::sm_faces_local(sm);
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_cell_faces, nc * 4 * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_face_pts, (size_t)nfl * 2 * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_face_dir, nfl));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_face_compress, (size_t)nfl * sizeof(int32_t)));
    const uint32_t nthreads = nfl > nc ? nfl : (uint32_t)nc;
    hipLaunchKernelGGL(pa::structured_faces_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, ctx->stream, sm, nfl,
                       ctx->d_face_pts, ctx->d_face_dir, ctx->d_face_compress, (uint32_t)nc, ctx->d_cell_faces);
    PA_HIP(ctx, hipGetLastError());
    ctx->nfaces_local = nfl; ctx->face_base = pa::sm_face_base(sm); ctx->num_other_faces = pa::sm_num_other_faces(sm);
    ctx->sm = sm; ctx->structured = true;
    return PA_OK;
}

int pa_mesh_set_points(pa_context *ctx, const double *d_points, size_t npoints)
{
    if (!ctx || !d_points) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_points || !ctx->owns_mesh) return PA_ERR_NO_MESH;
    if (npoints != ctx->npoints) return PA_ERR_INVALID_ARG;
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_points, d_points, npoints * 2 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return PA_OK;
}

int pa_mesh_set_faces(pa_context *ctx, const uint32_t *cell_faces, const uint32_t *face_pts,
                      const uint8_t *face_is_dirichlet, size_t nfaces)
{
    if (!ctx || !cell_faces || !face_pts || !face_is_dirichlet || nfaces == 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_points) return PA_ERR_NO_MESH;
    for (size_t i = 0; i < 4 * ctx->ncells; ++i)
        if (cell_faces[i] >= nfaces) return PA_ERR_INVALID_ARG;
    for (size_t i = 0; i < 2 * nfaces; ++i)
        if (face_pts[i] >= ctx->npoints) return PA_ERR_INVALID_ARG;
    PA_HIP(ctx, hipSetDevice(ctx->device));
    release_faces(ctx);
    // compress table, hho.hpp:313-323
    int32_t *comp = (int32_t *)std::malloc(nfaces * sizeof(int32_t));
    if (!comp) return PA_ERR_INVALID_ARG;
    size_t co = 0;
    for (size_t i = 0; i < nfaces; ++i) comp[i] = face_is_dirichlet[i] ? -1 : (int32_t)co++;
    hipError_t e = hipMalloc((void **)&ctx->d_cell_faces, (ctx->ncells ? ctx->ncells : 1) * 4 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_face_pts, nfaces * 2 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_face_dir, nfaces);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_face_compress, nfaces * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_cell_faces, cell_faces, ctx->ncells * 4 * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_face_pts, face_pts, nfaces * 2 * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_face_dir, face_is_dirichlet, nfaces, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_face_compress, comp, nfaces * sizeof(int32_t), hipMemcpyHostToDevice);
    std::free(comp);
    if (e != hipSuccess) { ctx->last_error = std::string("pa_mesh_set_faces: ") + hipGetErrorString(e); return PA_ERR_HIP; }
    ctx->nfaces_local = nfaces; ctx->face_base = 0; ctx->num_other_faces = co;
    return PA_OK;
}

int pa_assembler_query(pa_context *ctx, pa_degree_info di, pa_assembler_info *out)
{
    if (!ctx || !out || di.cell_deg < 0 || di.face_deg < 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    out->ncells_global = ctx->ncells_global; out->cell_base = ctx->cell_base;
    out->nfaces_local = ctx->nfaces_local; out->face_base = ctx->face_base;
    out->num_other_faces = ctx->num_other_faces;
    out->system_size = (uint64_t)pa::P2(di.cell_deg) * ctx->ncells_global + (uint64_t)(di.face_deg + 1) * ctx->num_other_faces;
    return PA_OK;
}

int pa_dirichlet_data_batch(pa_context *ctx, int face_deg, int fn, const double *d_fvals, double *d_g)
{
    if (!ctx || !d_g || face_deg < 0 || face_deg > 3) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (fn < PA_FN_SAMPLED || fn > PA_FN_ONE || (fn == PA_FN_SAMPLED && !d_fvals)) return PA_ERR_INVALID_ARG;
    const uint32_t nf = (uint32_t)ctx->nfaces_local;
    if (nf == 0) return PA_OK;
    const dim3 grid((nf + 255) / 256), block(256);
#define PA_DD_CASE(FD)                                                                                              \
    case FD:                                                                                                        \
        hipLaunchKernelGGL((pa::dirichlet_data_kernel<FD>), grid, block, 0, ctx->stream, ctx->d_tab, ctx->d_points, \
                           ctx->d_face_pts, ctx->d_face_dir, nf, fn, d_fvals, d_g);                                 \
        break;
    switch (face_deg) { PA_DD_CASE(0) PA_DD_CASE(1) PA_DD_CASE(2) PA_DD_CASE(3) }
#undef PA_DD_CASE
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_face_quadrature_points(pa_context *ctx, int face_deg, double *d_xyw)
{
    if (!ctx || !d_xyw || face_deg < 0 || face_deg > 7) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    const uint32_t nf = (uint32_t)ctx->nfaces_local;
    if (nf == 0) return PA_OK;
    hipLaunchKernelGGL(pa::face_qpoints_kernel, dim3((nf + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_tab,
                       ctx->d_points, ctx->d_face_pts, nf, face_deg + 1, d_xyw);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_triplets_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n, const double *d_lc,
                      const double *d_rhs, const double *d_g, int32_t *d_rows, int32_t *d_cols, double *d_vals,
                      int32_t *d_rhs_rows, double *d_rhs_vals)
{
    if (!ctx || !d_lc || !d_rows || !d_cols || !d_vals || !d_rhs_rows || !d_rhs_vals) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (di.cell_deg < 0 || di.face_deg < 0 || di.face_deg > 3 || di.cell_deg > 4) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    pa_assembler_info info;
    pa_assembler_query(ctx, di, &info);
    if (info.system_size >= ((uint64_t)1 << 31)) return PA_ERR_INVALID_ARG;      // Eigen::Triplet stores int indices
    if (n == 0) return PA_OK;
    pa::TripletArgs a;
    a.cell_faces = ctx->d_cell_faces; a.face_dir = ctx->d_face_dir; a.face_compress = ctx->d_face_compress;
    a.g = d_g; a.lc = d_lc; a.rhs = d_rhs; a.first = first; a.n = n;
    a.cell_base = ctx->cell_base; a.ncells_global = ctx->ncells_global;
    a.cbs = pa::P2(di.cell_deg); a.fbs = di.face_deg + 1;
    a.rows = d_rows; a.cols = d_cols; a.vals = d_vals; a.rhs_rows = d_rhs_rows; a.rhs_vals = d_rhs_vals;
    const int msize = a.cbs + 4 * a.fbs;
    const size_t shmem = msize * sizeof(double) + msize * sizeof(int32_t);
    const size_t resident = (size_t)ctx->num_cus * 8;
    const int grid = (int)(n < resident ? n : resident);
    hipLaunchKernelGGL(pa::triplets_kernel, dim3(grid), dim3(256), shmem, ctx->stream, a);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

static int take_local(pa_context *ctx, pa_degree_info di, size_t first, size_t n, const double *d_solution,
                      const double *d_g, int expanded, double *d_out)
{
    if (!ctx || !d_solution || !d_out) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (di.cell_deg < 0 || di.face_deg < 0 || di.face_deg > 3 || di.cell_deg > 4) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    if (n == 0) return PA_OK;
    pa::TakeArgs a;
    a.cell_faces = ctx->d_cell_faces; a.face_compress = ctx->d_face_compress; a.g = d_g; a.solution = d_solution;
    a.first = first; a.n = n; a.cell_base = ctx->cell_base; a.ncells_global = ctx->ncells_global;
    a.face_base = ctx->face_base; a.cbs = pa::P2(di.cell_deg); a.fbs = di.face_deg + 1; a.expanded = expanded;
    a.out = d_out;
    const size_t total = n * (size_t)(a.cbs + 4 * a.fbs);
    hipLaunchKernelGGL(pa::take_local_data_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_take_local_data_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n, const double *d_solution,
                             const double *d_g, double *d_out)
{
    return take_local(ctx, di, first, n, d_solution, d_g, 0, d_out);
}

int pa_obstacle_take_local_data_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n,
                                      const double *d_expanded, double *d_out)
{
    return take_local(ctx, di, first, n, d_expanded, nullptr, 1, d_out);
}

static bool whole_mesh(const pa_context *ctx)
{
    return ctx->d_cell_faces && ctx->cell_base == 0 && ctx->ncells == ctx->ncells_global;
}

int pa_obstacle_tables(pa_context *ctx, const uint8_t *d_in_A, int32_t *d_A_ct, int32_t *d_B_ct, size_t *num_I,
                       size_t *num_A)
{
    if (!ctx || !d_in_A || !d_A_ct || !d_B_ct) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_ptids) return PA_ERR_NO_MESH;
    const uint32_t n = (uint32_t)ctx->ncells;
    const uint32_t nblocks = (n + pa::SCAN_TILE - 1) / pa::SCAN_TILE;
    uint32_t *d_counts = nullptr;
    PA_HIP(ctx, hipMalloc(&d_counts, (nblocks + 1) * sizeof(uint32_t)));
    hipLaunchKernelGGL(pa::active_count_kernel, dim3(nblocks), dim3(pa::SCAN_BLOCK), 0, ctx->stream, d_in_A, n, d_counts);
    hipLaunchKernelGGL(pa::active_block_scan_kernel, dim3(1), dim3(pa::SCAN_BLOCK), 0, ctx->stream, d_counts, nblocks);
    hipLaunchKernelGGL(pa::active_tables_kernel, dim3(nblocks), dim3(pa::SCAN_BLOCK), 0, ctx->stream, d_in_A, n, d_counts,
                       d_A_ct, d_B_ct);
    uint32_t total = 0;
    hipError_t e = hipMemcpyAsync(&total, d_counts + nblocks, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_counts);
    PA_HIP(ctx, e);
    if (num_A) *num_A = total;
    if (num_I) *num_I = n - total;
    return PA_OK;
}

int pa_obstacle_triplets_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n, const double *d_lc,
                               const double *d_rhs, const double *d_g, const double *d_gamma, const uint8_t *d_in_A,
                               const int32_t *d_A_ct, const int32_t *d_B_ct, size_t num_I, int32_t *d_rows,
                               int32_t *d_cols, double *d_vals, int32_t *d_rhs_rows, double *d_rhs_vals)
{
    if (!ctx || !d_lc || !d_gamma || !d_in_A || !d_A_ct || !d_B_ct || !d_rows || !d_cols || !d_vals || !d_rhs_rows ||
        !d_rhs_vals)
        return PA_ERR_INVALID_ARG;
    if (di.cell_deg < 0 || di.face_deg < 0 || di.face_deg > 3 || di.cell_deg > 4) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (!whole_mesh(ctx)) { ctx->last_error = "obstacle assembler needs the whole mesh on the context"; return PA_ERR_INVALID_ARG; }
    if (first > ctx->ncells || n > ctx->ncells - first || num_I > ctx->ncells) return PA_ERR_INVALID_ARG;
    pa_assembler_info info;
    pa_assembler_query(ctx, di, &info);
    if (info.system_size >= ((uint64_t)1 << 31)) return PA_ERR_INVALID_ARG;
    if (n == 0) return PA_OK;
    pa::ObstacleArgs o;
    pa::TripletArgs &a = o.t;
    a.cell_faces = ctx->d_cell_faces; a.face_dir = ctx->d_face_dir; a.face_compress = ctx->d_face_compress;
    a.g = d_g; a.lc = d_lc; a.rhs = d_rhs; a.first = first; a.n = n;
    a.cell_base = 0; a.ncells_global = ctx->ncells_global;
    a.cbs = pa::P2(di.cell_deg); a.fbs = di.face_deg + 1;
    a.rows = d_rows; a.cols = d_cols; a.vals = d_vals; a.rhs_rows = d_rhs_rows; a.rhs_vals = d_rhs_vals;
    o.in_A = d_in_A; o.A_ct = d_A_ct; o.B_ct = d_B_ct; o.gamma = d_gamma; o.num_I = num_I; o.num_other = ctx->num_other_faces;
    const int msize = a.cbs + 4 * a.fbs;
    const size_t shmem = msize * sizeof(double) + 2 * msize * sizeof(int32_t);
    const size_t resident = (size_t)ctx->num_cus * 8;
    const int grid = (int)(n < resident ? n : resident);
    hipLaunchKernelGGL(pa::obstacle_triplets_kernel, dim3(grid), dim3(256), shmem, ctx->stream, o);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_obstacle_expand_solution(pa_context *ctx, pa_degree_info di, const double *d_solution, const double *d_g,
                                const double *d_gamma, const uint8_t *d_in_A, const int32_t *d_A_ct,
                                const int32_t *d_B_ct, size_t num_I, double *d_alpha, double *d_beta)
{
    if (!ctx || !d_solution || !d_gamma || !d_in_A || !d_A_ct || !d_B_ct || !d_alpha || !d_beta) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (di.cell_deg < 0 || di.face_deg < 0 || di.face_deg > 3 || di.cell_deg > 4) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (!whole_mesh(ctx)) { ctx->last_error = "obstacle assembler needs the whole mesh on the context"; return PA_ERR_INVALID_ARG; }
    pa::ExpandArgs a;
    a.in_A = d_in_A; a.face_dir = ctx->d_face_dir; a.A_ct = d_A_ct; a.B_ct = d_B_ct; a.face_compress = ctx->d_face_compress;
    a.solution = d_solution; a.g = d_g; a.gamma = d_gamma;
    a.ncells = ctx->ncells; a.nfaces = ctx->nfaces_local; a.num_I = num_I; a.num_other = ctx->num_other_faces;
    a.cbs = pa::P2(di.cell_deg); a.fbs = di.face_deg + 1; a.alpha = d_alpha; a.beta = d_beta;
    const uint64_t total = a.ncells * a.cbs + a.nfaces * a.fbs;
    hipLaunchKernelGGL(pa::obstacle_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_csr_from_triplets(pa_context *ctx, size_t nslots, const int32_t *d_rows, const int32_t *d_cols, const double *d_vals,
                         size_t nrows, int64_t *d_rowptr, int32_t *d_colind, double *d_values, size_t *nnz)
{
    if (!ctx || !d_rowptr || (nslots && (!d_rows || !d_cols || !d_vals || !d_colind || !d_values))) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (nslots >= ((size_t)1 << 31) || nrows >= ((size_t)1 << 31)) return PA_ERR_INVALID_ARG;
    PA_HIP(ctx, pa::csr_from_triplets(ctx->stream, nslots, d_rows, d_cols, d_vals, nrows, d_rowptr, d_colind, d_values, nnz));
    return PA_OK;
}

int pa_conjugated_gradient(pa_context *ctx, size_t nrows, const int64_t *d_rowptr, const int32_t *d_colind, const double *d_values,
                           const double *d_b, double *d_x, double convergence_threshold, double divergence_threshold,
                           size_t max_iter, int apply_preconditioner, int32_t *exit_reason, size_t *iterations,
                           double *relative_residual)
{
    if (!ctx || !d_rowptr || (nrows && (!d_colind || !d_values || !d_b || !d_x))) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    int reason = 0;
    PA_HIP(ctx, pa::conjugated_gradient(ctx->stream, nrows, d_rowptr, d_colind, d_values, d_b, d_x, convergence_threshold,
                                        divergence_threshold, max_iter, apply_preconditioner, &reason, iterations, relative_residual));
    if (exit_reason) *exit_reason = reason;
    return PA_OK;
}

int pa_conjugated_gradient_rows(pa_context *ctx, const pa_cg_transport *transport, int64_t row_begin, int64_t row_end,
                                const int64_t *d_rowptr, const int32_t *d_colind, const double *d_values, const double *d_b, double *d_x,
                                double convergence_threshold, double divergence_threshold, size_t max_iter, int apply_preconditioner,
                                int32_t *exit_reason, size_t *iterations, double *relative_residual, int32_t *transport_status)
{
    if (!ctx || !d_rowptr || row_end < row_begin) return PA_ERR_INVALID_ARG;
    if (row_end > row_begin && (!d_colind || !d_values || !d_b || !d_x)) return PA_ERR_INVALID_ARG;
    if (transport && (!transport->allreduce_sum || !transport->halo || !transport->neighbour_counts)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    pa::CgTransport tp{};
    if (transport) { tp.user = transport->user; tp.allreduce_sum = transport->allreduce_sum; tp.halo = transport->halo; tp.neighbour_counts = transport->neighbour_counts; }
    int reason = 0, tstat = 0;
    const hipError_t he = pa::conjugated_gradient_rows(ctx->stream, transport ? &tp : nullptr, row_begin, row_end, d_rowptr, d_colind, d_values,
                                                       d_b, d_x, convergence_threshold, divergence_threshold, max_iter, apply_preconditioner,
                                                       &reason, iterations, relative_residual, &tstat);
    if (exit_reason) *exit_reason = reason;
    if (transport_status) *transport_status = tstat;
    if (he != hipSuccess) { ctx->last_error = std::string("pa_conjugated_gradient_rows: ") + hipGetErrorString(he); return PA_ERR_HIP; }
    if (tstat == 3) ctx->last_error = "pa_conjugated_gradient_rows: another rank failed; every rank left the solve at the same reduction";
    return (tstat == 1 || tstat == 3) ? PA_ERR_COMM : (tstat == 2 ? PA_ERR_INVALID_ARG : PA_OK);
}

int pa_copy_to_host(pa_context *ctx, void *host_dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!host_dst || !d_src))) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (bytes) PA_HIP(ctx, hipMemcpyAsync(host_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

int pa_copy_to_device(pa_context *ctx, void *d_dst, const void *host_src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !host_src))) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (bytes) PA_HIP(ctx, hipMemcpyAsync(d_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

int pa_mesh_counts(pa_context *ctx, size_t *npoints, size_t *ncells)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (npoints) *npoints = ctx->npoints;
    if (ncells) *ncells = ctx->ncells;
    return PA_OK;
}

// ---- the hot path -------------------------------------------------------------------------
static int pick_lanes(int cd, int fd, int quad)
{
    const int gmin = min_lanes(cd, fd, quad);
    if (gmin == 0) return 0;
    int lanes = gmin;                                   // fewest lanes per cell = most cells per wavefront
#ifdef PA_TUNING      // profiling / A-B builds only (proton_amd/_build.py, PA_BUILD_TAG): the shipped library reads no knob
    if (const char *env = std::getenv("PA_LANES_PER_CELL")) {
        const int v = std::atoi(env);
        if ((v == 16 || v == 32 || v == 64) && v >= gmin) lanes = v;
    }
#endif
    return lanes;
}

static int select_kernel(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t n,
                         const pa::KernelEntry **entry, int *grid, bool cond = false)
{
    pa_sizes sz;
    const int st = pa_sizes_for(di, quad_kind, &sz);
    if (st != PA_OK) return st;
    if (stab_kind < PA_STAB_NONE || stab_kind > PA_STAB_FANCY) return PA_ERR_INVALID_ARG;
    const int lanes = pick_lanes(di.cell_deg, di.face_deg, quad_kind);
    const pa::KernelEntry *e = lanes ? find_kernel(di.cell_deg, di.face_deg, quad_kind, stab_kind, lanes) : nullptr;
    if (!e || (cond && !e->launch_cond)) return PA_ERR_INVALID_DEGREE;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cond ? e->func_cond : e->func, 64,
                                                     cond ? e->lds_bytes_cond : e->lds_bytes) != hipSuccess || per_cu < 1)
        per_cu = 1;
    // A persistent grid of exactly waves_per_simd x 4 blocks per CU: when the compiler needs fewer registers than
    // the launch bound allows the hardware could hold more, and more was measured to be slower (msize 9: +24 %)
    const int waves = cond ? e->waves_per_simd_cond : e->waves_per_simd;
    if (per_cu > 4 * waves) per_cu = 4 * waves;
#ifdef PA_TUNING
    if (const char *env = std::getenv("PA_BLOCKS_PER_CU")) {
        const int v = std::atoi(env);
        if (v > 0) per_cu = v;
    }
#endif
    const size_t cpb = 64 / lanes;
    size_t blocks = (n + cpb - 1) / cpb;
    const size_t resident = (size_t)per_cu * (size_t)ctx->num_cus;
    if (blocks > resident) blocks = resident;            // persistent: every block loops over its share of cells
    if (blocks == 0) blocks = 1;
    *entry = e;
    *grid = (int)blocks;
    return PA_OK;
}

// outputs of one pass over cells [first, first + n): the local-operator modes write oper / data / stab / lc, the
// condensed mode (cond) reads rhs (and uF) and writes the packed condensed records (or uT)
struct LocalOpsOut {
    double *oper = nullptr, *data = nullptr, *stab = nullptr, *lc = nullptr;
    int32_t *info = nullptr;
    bool cond = false;
    const double *rhs = nullptr, *uF = nullptr;
    double *cond_out = nullptr, *uT = nullptr;
};

static int run_local_ops(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t first, size_t n,
                         const LocalOpsOut &o)
{
    double *d_oper = o.oper, *d_data = o.data, *d_stab = o.stab, *d_lc = o.lc;
    int32_t *d_info = o.info;
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_points) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    PA_HIP(ctx, hipSetDevice(ctx->device));
    const pa::KernelEntry *e = nullptr;
    int grid = 0;
    const int st = select_kernel(ctx, di, quad_kind, stab_kind, n, &e, &grid, o.cond);
    if (st != PA_OK) return st;
    if (n == 0) return PA_OK;
    uint32_t ablate = 0;
#ifdef PA_TUNING      // stage ablation produces garbage operators on purpose: never in the shipped library
    if (const char *env = std::getenv("PA_ABLATE")) ablate = (uint32_t)std::strtoul(env, nullptr, 0);
#endif
    const bool split = !o.cond && (d_data != nullptr || d_stab != nullptr);
    // msize <= 9 and nothing but lc (and info) asked for: the thread-per-cell kernel of hho_small.hpp -- one launch, no record
    if (e->launch_small && !o.cond && !split && d_oper == nullptr && d_lc != nullptr && ablate == 0) {
        pa::SmallOpsArgs sa;
        sa.tab = ctx->d_tab; sa.points = ctx->d_points; sa.ptids = ctx->d_ptids;
        sa.first = first; sa.n = n; sa.lc = d_lc; sa.info = d_info;
        PA_HIP(ctx, e->launch_small(sa, ctx->stream));
        return PA_OK;
    }
    // Kernels that take the per-cell head from the pre-pass run in pieces of at most `piece` cells: pre-pass of a
    // piece into the context's record buffer, then the cooperative kernel over the same cells (same stream).
    size_t piece = n;
    if (e->launch_pre && e->self_pre) {
        // the cooperative kernel forms the records itself: one ring of 64 records per block of the (persistent) grid, one launch
        const size_t need = (size_t)grid * 64 * (size_t)e->pre_doubles;
        if (ctx->pre_capacity < need) {
            if (ctx->d_pre) { PA_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_pre); ctx->d_pre = nullptr; ctx->pre_capacity = 0; }
            PA_HIP(ctx, hipMalloc((void **)&ctx->d_pre, need * sizeof(double)));
            ctx->pre_capacity = need;
        }
    } else if (e->launch_pre) {
        size_t cap_bytes = ctx->pre_cap_bytes;                            // (pa_context_set_record_cap; default 4 GiB)
        const size_t per_cell = (size_t)e->pre_doubles * sizeof(double);
        size_t max_cells = (cap_bytes / per_cell) & ~(size_t)4095;
        // Pieces of at most PA_PIECE_CELLS cells by default: the records of a piece (134 MB at k = 2, 255 MB at k = 3) are then
        // still in the Infinity Cache when the cooperative kernel reads them, and the next piece's overwrite them there --
        // measured on 1024 x 1024 cells against one piece: 0.53 -> 0.46 ms at k = 1, 1.35 -> 1.27 ms at k = 2, 2.70 -> 2.44 ms at
        // k = 3, 11.1 -> 9.8 ms on 2048 x 2048 at k = 3 (tools/slab_timing.py; 96 Ki ... 256 Ki cells per piece within 2 %)
        // (not in the condensed mode, which writes 720 B per cell instead of 3.9 KB and is not short of HBM bandwidth: there the
        // extra launches and tails cost 2-8 %)
        if (!o.cond && max_cells > PA_PIECE_CELLS) max_cells = PA_PIECE_CELLS;
        if (max_cells < 4096) max_cells = 4096;
        if (piece > max_cells) {                                          // equal pieces, whole multiples of 4096 cells
            const size_t npieces = (n + max_cells - 1) / max_cells;
            piece = (((n + npieces - 1) / npieces) + 4095) & ~(size_t)4095;
        }
        const size_t need = ((piece + 7) / 8) * 8 * (size_t)e->pre_doubles;      // whole tiles of 8 records
        if (ctx->pre_capacity < need) {
            if (ctx->d_pre) { PA_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_pre); ctx->d_pre = nullptr; ctx->pre_capacity = 0; }
            PA_HIP(ctx, hipMalloc((void **)&ctx->d_pre, need * sizeof(double)));
            ctx->pre_capacity = need;
        }
    }
    pa_sizes sz;
    (void)pa_sizes_for(di, quad_kind, &sz);
    const size_t mm = (size_t)sz.msize * (size_t)sz.msize, opn = (size_t)sz.oper_rows * (size_t)sz.msize;
    for (size_t off = 0; off < n; off += piece) {
        const size_t m = n - off < piece ? n - off : piece;
        int g = grid;
        if (m != n) {
            const pa::KernelEntry *e2 = nullptr;
            const int st2 = select_kernel(ctx, di, quad_kind, stab_kind, m, &e2, &g, o.cond);
            if (st2 != PA_OK) return st2;
        }
        pa::LocalOpsArgs a;
        a.tab = ctx->d_tab; a.points = ctx->d_points; a.ptids = ctx->d_ptids;
        a.first = first + off; a.n = m;
        a.pre = nullptr;
        a.pre_ring = nullptr;
        if (e->launch_pre && e->self_pre) {
            a.pre_ring = ctx->d_pre;
        } else if (e->launch_pre) {
            pa::PreArgs pa_;
            pa_.tab = ctx->d_tab; pa_.points = ctx->d_points; pa_.ptids = ctx->d_ptids;
            pa_.first = first + off; pa_.n = m; pa_.pre = ctx->d_pre;
            PA_HIP(ctx, e->launch_pre(pa_, ctx->stream));
            a.pre = ctx->d_pre;
        }
        a.oper = d_oper ? d_oper + off * opn : nullptr;
        a.data = d_data ? d_data + off * mm : nullptr;
        a.stab = d_stab ? d_stab + off * mm : nullptr;
        a.lc = d_lc ? d_lc + off * mm : nullptr;
        a.info = d_info ? d_info + off : nullptr;
        const size_t ncond = (size_t)(4 * sz.fbs) * (size_t)(4 * sz.fbs + 1) / 2 + (size_t)(4 * sz.fbs);
        a.rhs = o.rhs ? o.rhs + off * (size_t)sz.cbs : nullptr;
        a.cond = o.cond_out ? o.cond_out + off * ncond : nullptr;
        a.uF = o.uF ? o.uF + off * (size_t)(4 * sz.fbs) : nullptr;
        a.uT = o.uT ? o.uT + off * (size_t)sz.cbs : nullptr;
        a.ablate = ablate;
        a.dbg = nullptr;
#ifdef PA_STAGE_CLOCK
        // diagnostic build: per-stage shader clocks of the cooperative kernel, averaged over blocks, to stderr
        static long long *d_dbg = nullptr;
        const size_t ndbg = (size_t)g * PA_NSTAGE;
        if (!d_dbg) (void)hipMalloc((void **)&d_dbg, (size_t)(1 << 20) * sizeof(long long));
        (void)hipMemsetAsync(d_dbg, 0, ndbg * sizeof(long long), ctx->stream);
        a.dbg = d_dbg;
#endif
        PA_HIP(ctx, (o.cond ? e->launch_cond : split ? e->launch_split : e->launch)(a, g, ctx->stream));
#ifdef PA_STAGE_CLOCK
        {
            std::vector<long long> h(ndbg);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipMemcpy(h.data(), d_dbg, ndbg * sizeof(long long), hipMemcpyDeviceToHost);
            double sum[PA_NSTAGE] = {0};
            for (int b = 0; b < g; ++b) for (int i = 0; i < PA_NSTAGE; ++i) sum[i] += (double)h[(size_t)b * PA_NSTAGE + i];
            const double iters = (double)m / (64 / e->lanes_per_cell);
            std::fprintf(stderr, "PA_STAGE_CLOCK %s grid %d: clocks per wave pass:", e->name, g);
            for (int i = 0; i < PA_NSTAGE; ++i) std::fprintf(stderr, " s%d=%.0f", i, sum[i] / iters);
            std::fprintf(stderr, "\n");
        }
#endif
    }
    return PA_OK;
}

int pa_local_ops_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t first, size_t n,
                       double *d_oper, double *d_data, double *d_stab, double *d_lc, int32_t *d_info)
{
    LocalOpsOut o;
    o.oper = d_oper; o.data = d_data; o.stab = d_stab; o.lc = d_lc; o.info = d_info;
    return run_local_ops(ctx, di, quad_kind, stab_kind, first, n, o);
}

int pa_condensed_ops_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t first, size_t n,
                           const double *d_rhs, double *d_cond, int32_t *d_info)
{
    if (!d_cond) return PA_ERR_INVALID_ARG;
    if (stab_kind == PA_STAB_NONE) return PA_ERR_INVALID_ARG;      // A_TT = data_TT is singular (constants)
    LocalOpsOut o;
    o.cond = true; o.rhs = d_rhs; o.cond_out = d_cond; o.info = d_info;
    return run_local_ops(ctx, di, quad_kind, stab_kind, first, n, o);
}

int pa_condensed_recover_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t first, size_t n,
                               const double *d_rhs, const double *d_uF, double *d_uT, int32_t *d_info)
{
    if (!d_uF || !d_uT) return PA_ERR_INVALID_ARG;
    if (stab_kind == PA_STAB_NONE) return PA_ERR_INVALID_ARG;
    LocalOpsOut o;
    o.cond = true; o.rhs = d_rhs; o.uF = d_uF; o.uT = d_uT; o.info = d_info;
    return run_local_ops(ctx, di, quad_kind, stab_kind, first, n, o);
}

static int launch_info(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t n, bool cond, pa_launch_info *out)
{
    if (!ctx || !out) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const pa::KernelEntry *e = nullptr;
    int grid = 0;
    const int st = select_kernel(ctx, di, quad_kind, stab_kind, n, &e, &grid, cond);
    if (st != PA_OK) return st;
    out->lanes_per_cell = e->lanes_per_cell;
    out->cells_per_block = 64 / e->lanes_per_cell;
    out->block_threads = 64;
    out->lds_bytes_per_block = cond ? e->lds_bytes_cond : e->lds_bytes;
    out->grid_blocks = grid;
    out->kernel_name = cond ? e->name_cond : e->name;
    return PA_OK;
}

int pa_local_ops_launch_info(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t n, pa_launch_info *out)
{
    return launch_info(ctx, di, quad_kind, stab_kind, n, false, out);
}

int pa_condensed_launch_info(pa_context *ctx, pa_degree_info di, int quad_kind, int stab_kind, size_t n, pa_launch_info *out)
{
    return launch_info(ctx, di, quad_kind, stab_kind, n, true, out);
}

}  // extern "C"

// ---- right-hand sides, quadrature points ---------------------------------------------------
template <int QUAD>
static int launch_rhs(pa_context *ctx, int degree, int qdeg, int nqp, int fn, const double *d_fvals, size_t first,
                      size_t n, double *d_rhs, const int8_t *d_cell_loc = nullptr, int where = 0)
{
    const int block = 256;
    const int grid = (int)((n + block - 1) / block);
#define PA_RHS_CASE(D)                                                                                     \
    case D:                                                                                                \
        hipLaunchKernelGGL((pa::cell_rhs_kernel<D, QUAD>), dim3(grid), dim3(block), 0, ctx->stream, ctx->d_tab, \
                           ctx->d_points, ctx->d_ptids, first, n, qdeg, nqp, fn, d_fvals, d_rhs, d_cell_loc, where); \
        break;
    switch (degree) {
        PA_RHS_CASE(0) PA_RHS_CASE(1) PA_RHS_CASE(2) PA_RHS_CASE(3) PA_RHS_CASE(4)
    default: return PA_ERR_INVALID_DEGREE;
    }
#undef PA_RHS_CASE
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

template <int QUAD>
static int launch_project(pa_context *ctx, int degree, int qdeg, int nqp, int fn, const double *d_fvals, size_t first,
                          size_t n, double *d_out, int stride, int32_t *d_info)
{
    const int block = 256;
    const int grid = (int)((n + block - 1) / block);
#define PA_PROJ_CASE(D)                                                                                    \
    case D:                                                                                                \
        hipLaunchKernelGGL((pa::cell_project_kernel<D, QUAD>), dim3(grid), dim3(block), 0, ctx->stream, ctx->d_tab, \
                           ctx->d_points, ctx->d_ptids, first, n, qdeg, nqp, fn, d_fvals, d_out, stride, d_info); \
        break;
    switch (degree) {
        PA_PROJ_CASE(0) PA_PROJ_CASE(1) PA_PROJ_CASE(2) PA_PROJ_CASE(3) PA_PROJ_CASE(4)
    default: return PA_ERR_INVALID_DEGREE;
    }
#undef PA_PROJ_CASE
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

extern "C" {

static int rhs_quadrature(pa_context *ctx, int qdeg, int quad_kind, int *nqp)
{
    if (quad_kind == PA_QUAD_TENSOR) {
        if (pa::gauss_nodes(qdeg) > 8) return PA_ERR_QUADRATURE;      // tables: closed forms to 5 nodes, golub_welsch's rules to 8
    } else if (quad_kind == PA_QUAD_FAN) {
        if (qdeg > 8) return PA_ERR_QUADRATURE;
    } else return PA_ERR_INVALID_ARG;
    *nqp = pa::cell_qp_count(&ctx->host_tab, quad_kind, qdeg);
    return PA_OK;
}

int pa_cell_rhs_batch(pa_context *ctx, int degree, int dinc, int quad_kind, int fn, const double *d_fvals,
                      size_t first, size_t n, double *d_rhs)
{
    if (!ctx || !d_rhs || degree < 0 || dinc < 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_points) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    if (fn < PA_FN_SAMPLED || fn > PA_FN_ONE || (fn == PA_FN_SAMPLED && !d_fvals)) return PA_ERR_INVALID_ARG;
    const int qdeg = 2 * (degree + dinc);                          // utils.hpp:165
    int nqp = 0;
    const int st = rhs_quadrature(ctx, qdeg, quad_kind, &nqp);
    if (st != PA_OK) return st;
    if (n == 0) return PA_OK;
    return quad_kind == PA_QUAD_TENSOR ? launch_rhs<pa::QUAD_TENSOR>(ctx, degree, qdeg, nqp, fn, d_fvals, first, n, d_rhs)
                                       : launch_rhs<pa::QUAD_FAN>(ctx, degree, qdeg, nqp, fn, d_fvals, first, n, d_rhs);
}

int pa_project_function_batch(pa_context *ctx, pa_degree_info di, int quad_kind, int dinc, int fn,
                              const double *d_cell_fvals, const double *d_face_fvals, size_t first, size_t n,
                              double *d_out, int32_t *d_info)
{
    if (!ctx || !d_out || dinc < 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (di.cell_deg < 0 || di.cell_deg > 4 || di.face_deg < 0 || di.face_deg > 3) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_points) return PA_ERR_NO_MESH;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    if (fn < PA_FN_SAMPLED || fn > PA_FN_ONE || (fn == PA_FN_SAMPLED && (!d_cell_fvals || !d_face_fvals))) return PA_ERR_INVALID_ARG;
    const int qdeg = 2 * (di.cell_deg + dinc);                      // utils.hpp:123,165
    int nqp = 0;
    int st = rhs_quadrature(ctx, qdeg, quad_kind, &nqp);
    if (st != PA_OK) return st;
    const int nfq = di.face_deg + dinc + 1;                         // integrate(msh, fc, 2*(facdeg+di))
    if (nfq > 8) return PA_ERR_QUADRATURE;
    if (n == 0) return PA_OK;
    const int cbs = pa::P2(di.cell_deg), msize = cbs + 4 * (di.face_deg + 1);
    st = quad_kind == PA_QUAD_TENSOR
             ? launch_project<pa::QUAD_TENSOR>(ctx, di.cell_deg, qdeg, nqp, fn, d_cell_fvals, first, n, d_out, msize, d_info)
             : launch_project<pa::QUAD_FAN>(ctx, di.cell_deg, qdeg, nqp, fn, d_cell_fvals, first, n, d_out, msize, d_info);
    if (st != PA_OK) return st;
    const int grid = (int)((4 * n + 255) / 256);
#define PA_FPROJ_CASE(D)                                                                                   \
    case D:                                                                                                \
        hipLaunchKernelGGL((pa::face_project_kernel<D>), dim3(grid), dim3(256), 0, ctx->stream, ctx->d_tab, ctx->d_points, \
                           ctx->d_face_pts, ctx->d_cell_faces, first, n, nfq, fn, d_face_fvals, d_out, msize, cbs); \
        break;
    switch (di.face_deg) { PA_FPROJ_CASE(0) PA_FPROJ_CASE(1) PA_FPROJ_CASE(2) PA_FPROJ_CASE(3) }
#undef PA_FPROJ_CASE
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_energy_form_batch(pa_context *ctx, pa_degree_info di, size_t n, const double *d_lc, const double *d_u,
                         const double *d_v, double *d_out)
{
    if (!ctx || !d_lc || !d_u || !d_out) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (di.cell_deg < 0 || di.cell_deg > 4 || di.face_deg < 0 || di.face_deg > 3) return PA_ERR_INVALID_DEGREE;
    if (n == 0) return PA_OK;
    const int msize = pa::P2(di.cell_deg) + 4 * (di.face_deg + 1);
    const size_t resident = (size_t)ctx->num_cus * 32;
    hipLaunchKernelGGL(pa::energy_form_kernel, dim3((unsigned)(n < resident ? n : resident)), dim3(64), 0, ctx->stream, n, msize,
                       d_lc, d_u, d_v, d_out);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_cell_quadrature_points(pa_context *ctx, int degree, int quad_kind, size_t first, size_t n, double *d_xyw,
                              int32_t *nqp_out)
{
    if (!ctx || degree < 0) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    int nqp = 0;
    const int st = rhs_quadrature(ctx, degree, quad_kind, &nqp);
    if (st != PA_OK) return st;
    if (nqp_out) *nqp_out = nqp;
    if (!d_xyw) return PA_OK;                                       // size query
    if (!ctx->d_points) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    if (n == 0 || nqp == 0) return PA_OK;
    const int block = 256;
    const int grid = (int)((n + block - 1) / block);
    if (quad_kind == PA_QUAD_TENSOR)
        hipLaunchKernelGGL((pa::cell_qpoints_kernel<pa::QUAD_TENSOR>), dim3(grid), dim3(block), 0, ctx->stream, ctx->d_tab,
                           ctx->d_points, ctx->d_ptids, first, n, degree, nqp, d_xyw);
    else
        hipLaunchKernelGGL((pa::cell_qpoints_kernel<pa::QUAD_FAN>), dim3(grid), dim3(block), 0, ctx->stream, ctx->d_tab,
                           ctx->d_points, ctx->d_ptids, first, n, degree, nqp, d_xyw);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

// ---- static condensation -------------------------------------------------------------------
static int condense(pa_context *ctx, pa_degree_info di, size_t n, const double *d_lc, const double *d_rhs, double *d_S,
                    double *d_g, double *d_rec, int32_t *d_info, int packed);

int pa_static_condensation_batch(pa_context *ctx, pa_degree_info di, size_t n, const double *d_lc, const double *d_rhs,
                                 double *d_S, double *d_g, double *d_rec, int32_t *d_info)
{
    return condense(ctx, di, n, d_lc, d_rhs, d_S, d_g, d_rec, d_info, 0);
}

int pa_static_condensation_packed_batch(pa_context *ctx, pa_degree_info di, size_t n, const double *d_lc,
                                        const double *d_rhs, double *d_Sp, double *d_g, int32_t *d_info)
{
    return condense(ctx, di, n, d_lc, d_rhs, d_Sp, d_g, nullptr, d_info, 1);
}

static int condense(pa_context *ctx, pa_degree_info di, size_t n, const double *d_lc, const double *d_rhs, double *d_S,
                    double *d_g, double *d_rec, int32_t *d_info, int packed)
{
    if (!ctx || !d_lc) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    pa_sizes sz;
    const int st = pa_sizes_for(di, PA_QUAD_TENSOR, &sz);
    if (st != PA_OK && st != PA_ERR_QUADRATURE) return st;
    if (n == 0) return PA_OK;
#define PA_SC_CASE(CD, FD)                                                                                    \
    if (di.cell_deg == CD && di.face_deg == FD) {                                                             \
        constexpr int G_ = (pa::P2(CD) <= 16 && 4 * (FD + 1) + 1 <= 16) ? 16 : 32;                            \
        const size_t blocks = (n + 64 / G_ - 1) / (64 / G_), resident = (size_t)ctx->num_cus * 16;             \
        const int grid = (int)(blocks < resident ? blocks : resident);                                        \
        hipLaunchKernelGGL((pa::static_condensation_kernel<pa::P2(CD), 4 * (FD + 1), G_>), dim3(grid), dim3(64), 0, \
                           ctx->stream, n, d_lc, d_rhs, d_S, d_g, d_rec, d_info, packed);                     \
        PA_HIP(ctx, hipGetLastError());                                                                       \
        return PA_OK;                                                                                         \
    }
    PA_SC_CASE(1, 0) PA_SC_CASE(0, 0) PA_SC_CASE(2, 1) PA_SC_CASE(1, 1) PA_SC_CASE(0, 1) PA_SC_CASE(3, 2)
    PA_SC_CASE(2, 2) PA_SC_CASE(1, 2) PA_SC_CASE(4, 3) PA_SC_CASE(3, 3) PA_SC_CASE(2, 3)
#undef PA_SC_CASE
    return PA_ERR_INVALID_DEGREE;
}

// ---- condensed (face-only) system -----------------------------------------------------------------
static bool cond_degree_ok(pa_degree_info di) { return di.cell_deg >= 0 && di.cell_deg <= 4 && di.face_deg >= 0 && di.face_deg <= 3; }

static pa::CondMesh cond_mesh(const pa_context *ctx)
{
    pa::CondMesh m;
    m.cell_faces = ctx->d_cell_faces; m.face_compress = ctx->d_face_compress; m.adj = ctx->d_adj;
    m.sm = ctx->sm; m.structured = ctx->structured;
    return m;
}

// first compressed id at or after global face `gid` of the generator mesh (the compress table is monotone)
static int32_t sm_first_compress_from(const pa::StructuredMesh &sm, uint32_t gid)
{
    const uint32_t nfaces = sm.Ny * pa::sm_face_row(sm) + sm.Nx;
    for (uint32_t f = gid; f < nfaces; ++f) {
        uint32_t lo, hi; bool d; int32_t comp;
        pa::sm_face_decode(sm, f, lo, hi, d, comp);
        if (!d) return comp;
    }
    return (int32_t)pa::sm_num_other_faces(sm);
}

// symbolic phase, cached per mesh: adjacency, owned faces, their column faces
static int cond_prepare(pa_context *ctx)
{
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (ctx->cond_ready) return PA_OK;
    PA_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t owned_range, nown; int32_t p0;
    if (ctx->structured) {
        const pa::StructuredMesh &sm = ctx->sm;
        owned_range = (sm.row1 - sm.row0) * pa::sm_face_row(sm);
        p0 = sm_first_compress_from(sm, sm.row0 * pa::sm_face_row(sm));
        const int32_t p1 = sm_first_compress_from(sm, sm.row1 * pa::sm_face_row(sm));
        nown = (uint32_t)(p1 - p0);
    } else {
        owned_range = (uint32_t)ctx->nfaces_local; p0 = 0; nown = (uint32_t)ctx->num_other_faces;
    }
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_adj, (ctx->nfaces_local ? ctx->nfaces_local : 1) * 2 * sizeof(int32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_cfaces, ((size_t)nown + 1) * sizeof(pa::CondFace)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_cfaces_lean, ((size_t)nown + 1) * sizeof(pa::CondFaceLean)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_ncols, ((size_t)nown + 1) * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_prefix, ((size_t)nown + 1) * sizeof(uint32_t)));
    PA_HIP(ctx, pa::cond_build_tables(ctx->stream, cond_mesh(ctx), (uint32_t)ctx->nfaces_local, (uint32_t)ctx->ncells, owned_range, p0,
                                      nown, ctx->d_adj, ctx->d_cfaces, ctx->d_cfaces_lean, ctx->d_ncols, ctx->d_prefix));
    uint32_t total = 0;
    PA_HIP(ctx, hipMemcpy(&total, ctx->d_prefix + nown, sizeof(uint32_t), hipMemcpyDeviceToHost));
    ctx->cond_nown = nown; ctx->cond_owned_range = owned_range; ctx->cond_p0 = p0; ctx->cond_total_cols = total;
    ctx->cond_ready = true;
    return PA_OK;
}

// the part of pa_condensed_info a slab's closed forms determine
static void cond_partition_fill(const pa::StructuredMesh &sm, uint64_t fbs, pa_condensed_info *out)
{
    const int32_t p0 = sm_first_compress_from(sm, sm.row0 * pa::sm_face_row(sm));
    const int32_t p1 = sm_first_compress_from(sm, sm.row1 * pa::sm_face_row(sm));
    out->num_other_faces = pa::sm_num_other_faces(sm);
    out->system_size = fbs * out->num_other_faces;
    out->nf = (int32_t)(4 * fbs);
    out->cond_doubles = (int32_t)(4 * fbs * (4 * fbs + 1) / 2 + 4 * fbs);
    out->row_begin = (uint64_t)p0 * fbs;
    out->row_end = (uint64_t)p1 * fbs;
    out->nnz_owned = 0;
    out->halo_cells = sm.row1 < sm.Ny ? sm.Nx : 0;
    out->halo_doubles = (int32_t)(fbs * (4 * fbs + 1));
    out->has_below = sm.row0 > 0 ? 1 : 0;
}

int pa_condensed_partition_info(size_t Nx, size_t Ny, size_t row_begin, size_t row_end, pa_degree_info di, pa_condensed_info *out)
{
    if (!out || !cond_degree_ok(di) || Nx == 0 || Ny == 0 || row_begin >= row_end || row_end > Ny) return PA_ERR_INVALID_ARG;
    if ((Nx + 1) * (Ny + 1) >= ((size_t)1 << 32)) return PA_ERR_INVALID_ARG;
    const pa::StructuredMesh sm = {(uint32_t)Nx, (uint32_t)Ny, (uint32_t)row_begin, (uint32_t)row_end};
    cond_partition_fill(sm, (uint64_t)di.face_deg + 1, out);
    return PA_OK;
}

int pa_condensed_query(pa_context *ctx, pa_degree_info di, pa_condensed_info *out)
{
    if (!ctx || !out || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const int st = cond_prepare(ctx);
    if (st != PA_OK) return st;
    const uint64_t fbs = (uint64_t)di.face_deg + 1;
    out->num_other_faces = ctx->num_other_faces;
    out->system_size = fbs * ctx->num_other_faces;
    out->nf = (int32_t)(4 * fbs);
    out->cond_doubles = (int32_t)(4 * fbs * (4 * fbs + 1) / 2 + 4 * fbs);
    out->row_begin = (uint64_t)ctx->cond_p0 * fbs;
    out->row_end = ((uint64_t)ctx->cond_p0 + ctx->cond_nown) * fbs;
    out->nnz_owned = ctx->cond_total_cols * fbs * fbs;
    out->halo_cells = (ctx->structured && ctx->sm.row1 < ctx->sm.Ny) ? ctx->sm.Nx : 0;
    out->halo_doubles = (int32_t)(fbs * (4 * fbs + 1));
    out->has_below = (ctx->structured && ctx->sm.row0 > 0) ? 1 : 0;
    return PA_OK;
}

int pa_condensed_triplets_batch(pa_context *ctx, pa_degree_info di, size_t first, size_t n, const double *d_cond, const double *d_g,
                                int32_t *d_rows, int32_t *d_cols, double *d_vals, int32_t *d_rhs_rows, double *d_rhs_vals)
{
    if (!ctx || !d_cond || !d_rows || !d_cols || !d_vals || !d_rhs_rows || !d_rhs_vals) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!cond_degree_ok(di)) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    if ((uint64_t)(di.face_deg + 1) * ctx->num_other_faces >= ((uint64_t)1 << 31)) return PA_ERR_INVALID_ARG;      // int triplet indices
    PA_HIP(ctx, hipSetDevice(ctx->device));
    PA_HIP(ctx, pa::cond_triplets(ctx->stream, cond_mesh(ctx), ctx->num_cus, first, n, di.face_deg + 1, d_cond, d_g, d_rows, d_cols,
                                  d_vals, d_rhs_rows, d_rhs_vals));
    return PA_OK;
}

int pa_condensed_csr_pattern(pa_context *ctx, pa_degree_info di, int64_t *d_rowptr, int32_t *d_colind)
{
    if (!ctx || !d_rowptr || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const int st = cond_prepare(ctx);
    if (st != PA_OK) return st;
    if ((uint64_t)(di.face_deg + 1) * ctx->num_other_faces >= ((uint64_t)1 << 31)) return PA_ERR_INVALID_ARG;      // int32 column ids
    PA_HIP(ctx, pa::cond_pattern(ctx->stream, ctx->cond_nown, di.face_deg + 1, ctx->d_cfaces, ctx->d_prefix, d_rowptr, d_colind));
    return PA_OK;
}

int pa_condensed_csr_fill(pa_context *ctx, pa_degree_info di, const double *d_cond, const double *d_g, const double *d_halo_below,
                          double *d_values, double *d_rhs)
{
    if (!ctx || !d_cond || !d_values || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const int st = cond_prepare(ctx);
    if (st != PA_OK) return st;
    if (ctx->structured && ctx->sm.row0 > 0 && !d_halo_below) {
        ctx->last_error = "pa_condensed_csr_fill: this slab has a slab below: d_halo_below (pa_condensed_halo_pack of that slab) is required";
        return PA_ERR_INVALID_ARG;
    }
    PA_HIP(ctx, hipSetDevice(ctx->device));
    PA_HIP(ctx, pa::cond_fill(ctx->stream, cond_mesh(ctx), ctx->cond_nown, di.face_deg + 1, ctx->d_cfaces_lean, ctx->d_prefix, d_cond, d_g,
                              d_halo_below, d_values, d_rhs));
    return PA_OK;
}

// ---- the assembler's own system (cell + face unknowns) directly in CSR: assembler_csr.hip ---------------------------
static int asm_prepare(pa_context *ctx)
{
    const int st = cond_prepare(ctx);
    if (st != PA_OK) return st;
    if (ctx->structured && (ctx->sm.row0 != 0 || ctx->sm.row1 != ctx->sm.Ny)) {
        ctx->last_error = "pa_assembler_csr_*: whole-mesh contexts only (a slab assembles the face-only system: pa_condensed_*)";
        return PA_ERR_INVALID_ARG;
    }
    if (ctx->asm_ready) return PA_OK;
    const uint32_t nc = (uint32_t)ctx->ncells, nown = ctx->cond_nown;
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_asm_nfc, ((size_t)nc + 1) * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_asm_cprefix, ((size_t)nc + 1) * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_asm_nfcell, ((size_t)nown + 1) * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_asm_fprefix, ((size_t)nown + 1) * sizeof(uint32_t)));
    PA_HIP(ctx, pa::asm_build_tables(ctx->stream, cond_mesh(ctx), nc, nown, ctx->d_cfaces_lean, ctx->d_asm_nfc, ctx->d_asm_cprefix,
                                     ctx->d_asm_nfcell, ctx->d_asm_fprefix));
    uint32_t a = 0, b = 0;
    PA_HIP(ctx, hipMemcpy(&a, ctx->d_asm_cprefix + nc, sizeof(uint32_t), hipMemcpyDeviceToHost));
    PA_HIP(ctx, hipMemcpy(&b, ctx->d_asm_fprefix + nown, sizeof(uint32_t), hipMemcpyDeviceToHost));
    ctx->asm_cell_faces_total = a; ctx->asm_face_cells_total = b;
    ctx->asm_ready = true;
    return PA_OK;
}

static void asm_sizes(const pa_context *ctx, pa_degree_info di, uint64_t *cell_nnz, uint64_t *nnz, uint64_t *nrows)
{
    const uint64_t cbs = (uint64_t)(di.cell_deg + 2) * (di.cell_deg + 1) / 2, fbs = (uint64_t)di.face_deg + 1;
    *cell_nnz = cbs * (ctx->ncells * cbs + ctx->asm_cell_faces_total * fbs);
    *nnz = *cell_nnz + fbs * (ctx->asm_face_cells_total * cbs + ctx->cond_total_cols * fbs);
    *nrows = cbs * ctx->ncells + fbs * ctx->num_other_faces;
}

int pa_assembler_csr_query(pa_context *ctx, pa_degree_info di, pa_assembler_csr_info *out)
{
    if (!ctx || !out || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const int st = asm_prepare(ctx);
    if (st != PA_OK) return st;
    uint64_t cell_nnz;
    asm_sizes(ctx, di, &cell_nnz, &out->nnz, &out->nrows);
    return PA_OK;
}

int pa_assembler_csr_pattern(pa_context *ctx, pa_degree_info di, int64_t *d_rowptr, int32_t *d_colind)
{
    if (!ctx || !d_rowptr || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const int st = asm_prepare(ctx);
    if (st != PA_OK) return st;
    uint64_t cell_nnz, nnz, nrows;
    asm_sizes(ctx, di, &cell_nnz, &nnz, &nrows);
    if (nrows >= ((uint64_t)1 << 31)) return PA_ERR_INVALID_ARG;                    // int32 column ids
    PA_HIP(ctx, pa::asm_pattern(ctx->stream, cond_mesh(ctx), (di.cell_deg + 2) * (di.cell_deg + 1) / 2, di.face_deg + 1, (uint32_t)ctx->ncells,
                                ctx->cond_nown, cell_nnz, ctx->d_cfaces, ctx->d_prefix, ctx->d_asm_cprefix, ctx->d_asm_fprefix, d_rowptr,
                                d_colind));
    return PA_OK;
}

int pa_assembler_csr_fill(pa_context *ctx, pa_degree_info di, const double *d_lc, const double *d_rhs, const double *d_g,
                          double *d_values, double *d_RHS)
{
    if (!ctx || !d_lc || !d_values || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    const int st = asm_prepare(ctx);
    if (st != PA_OK) return st;
    uint64_t cell_nnz, nnz, nrows;
    asm_sizes(ctx, di, &cell_nnz, &nnz, &nrows);
    PA_HIP(ctx, pa::asm_fill(ctx->stream, cond_mesh(ctx), (di.cell_deg + 2) * (di.cell_deg + 1) / 2, di.face_deg + 1, (uint32_t)ctx->ncells,
                             ctx->cond_nown, cell_nnz, ctx->d_cfaces_lean, ctx->d_prefix, ctx->d_asm_cprefix, ctx->d_asm_fprefix, d_lc, d_rhs,
                             d_g, d_values, d_RHS));
    return PA_OK;
}

int pa_condensed_halo_pack(pa_context *ctx, pa_degree_info di, const double *d_cond, const double *d_g, double *d_halo)
{
    if (!ctx || !d_cond || !d_halo || !cond_degree_ok(di)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (!ctx->structured || ctx->sm.row1 >= ctx->sm.Ny) return PA_OK;          // nothing above this slab
    PA_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t Nx = ctx->sm.Nx;
    PA_HIP(ctx, pa::cond_halo_pack(ctx->stream, cond_mesh(ctx), (uint32_t)ctx->ncells - Nx, Nx, di.face_deg + 1, d_cond, d_g, d_halo));
    return PA_OK;
}

int pa_condensed_take_faces(pa_context *ctx, pa_degree_info di, size_t first, size_t n, const double *d_solution, const double *d_g,
                            double *d_uF)
{
    if (!ctx || !d_solution || !d_uF) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!cond_degree_ok(di)) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (first > ctx->ncells || n > ctx->ncells - first) return PA_ERR_INVALID_ARG;
    PA_HIP(ctx, hipSetDevice(ctx->device));
    PA_HIP(ctx, pa::cond_take_faces(ctx->stream, cond_mesh(ctx), first, n, di.face_deg + 1, d_solution, d_g, d_uF));
    return PA_OK;
}

int pa_condensed_expand_solution(pa_context *ctx, pa_degree_info di, const double *d_uT, const double *d_xF, double *d_full)
{
    if (!ctx || !d_uT || !d_full) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!cond_degree_ok(di)) return PA_ERR_INVALID_DEGREE;
    if (!ctx->d_cell_faces) return PA_ERR_NO_MESH;
    PA_HIP(ctx, hipSetDevice(ctx->device));
    PA_HIP(ctx, pa::cond_expand(ctx->stream, ctx->ncells, ctx->cell_base, ctx->ncells_global, pa::P2(di.cell_deg),
                                (size_t)(di.face_deg + 1) * ctx->num_other_faces, d_uT, d_xF, d_full));
    return PA_OK;
}

// ---- cutHHO -----------------------------------------------------------------------------------
static int cut_preprocess_impl(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                               const pa_level_set *ls, int refsteps, bool displace, size_t row_begin, size_t row_end);

int pa_cut_preprocess(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                      const pa_level_set *ls, int refsteps)
{
    return cut_preprocess_impl(ctx, Nx, Ny, min_x, max_x, min_y, max_y, ls, refsteps, true, 0, Ny);
}

int pa_cut_preprocess_rows(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                           const pa_level_set *ls, int refsteps, size_t row_begin, size_t row_end)
{
    return cut_preprocess_impl(ctx, Nx, Ny, min_x, max_x, min_y, max_y, ls, refsteps, true, row_begin, row_end);
}

int pa_cut_preprocess_agglomeration(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                                    const pa_level_set *ls, int refsteps)
{
    return cut_preprocess_impl(ctx, Nx, Ny, min_x, max_x, min_y, max_y, ls, refsteps, false, 0, Ny);
}

int pa_cut_agglo_query(pa_context *ctx, int8_t *agglo_set, int32_t *neighbors)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (ctx->ncells != ctx->cut->ncells()) {
        ctx->last_error = "pa_cut_agglo_query: whole-mesh contexts only";
        return PA_ERR_INVALID_ARG;
    }
    if (agglo_set) {
        std::vector<int8_t> a;
        pa::classify_agglomeration(*ctx->cut, a);
        std::memcpy(agglo_set, a.data(), a.size());
    }
    if (neighbors) {
        std::vector<int32_t> nb;
        pa::structured_neighbors(ctx->cut->sm, nb);
        std::memcpy(neighbors, nb.data(), nb.size() * sizeof(int32_t));
    }
    return PA_OK;
}

static int cut_preprocess_impl(pa_context *ctx, size_t Nx, size_t Ny, double min_x, double max_x, double min_y, double max_y,
                               const pa_level_set *ls, int refsteps, bool displace, size_t row_begin, size_t row_end)
{
    if (!ctx || !ls || refsteps < 0 || refsteps > 10 || (ls->kind != 0 && ls->kind != 1)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    int st = pa_mesh_generate(ctx, Nx, Ny, min_x, max_x, min_y, max_y, row_begin, row_end);
    if (st != PA_OK) return st;
    const bool whole = row_begin == 0 && row_end == Ny;
    pa::CutMeshHost *cm = new (std::nothrow) pa::CutMeshHost();
    if (!cm) return PA_ERR_INVALID_ARG;
    const pa::LevelSet L = {ls->kind, ls->radius, ls->alpha, ls->beta, ls->cut_y};
    try {
        // the host preprocessing always sees the WHOLE mesh (a node is displaced by looking at its neighbours; the tags of a
        // slab's faces and nodes are those of the whole mesh): every rank of a row partition runs the same deterministic pass
        pa::cut_preprocess(*cm, (uint32_t)Nx, (uint32_t)Ny, min_x, max_x, min_y, max_y, L, refsteps, displace);
    } catch (const std::exception &e) {
        ctx->last_error = std::string("cutHHO preprocessing: ") + e.what();
        delete cm;
        return PA_ERR_INVALID_ARG;
    }
    const size_t nc_all = cm->ncells(), nc = ctx->ncells, base = ctx->cell_base;
    if (!whole) {
        // a slab keeps its own cut cells (global ids on the host, ids relative to the slab on the device) and their polylines
        std::vector<uint32_t> mine;
        std::vector<pa::P2d> ifc;
        for (size_t r = 0; r < cm->cut_cells.size(); ++r) {
            const uint32_t c = cm->cut_cells[r];
            if (c < base || c >= base + nc) continue;
            mine.push_back(c);
            ifc.insert(ifc.end(), cm->iface.begin() + r * cm->nif, cm->iface.begin() + (r + 1) * cm->nif);
        }
        cm->cut_cells.swap(mine);
        cm->iface.swap(ifc);
        cm->cut_index.assign(nc_all, -1);
        for (size_t r = 0; r < cm->cut_cells.size(); ++r) cm->cut_index[cm->cut_cells[r]] = (int32_t)r;
    }
    release_cut(ctx);
    ctx->cut = cm;
    const size_t ncut = cm->cut_cells.size();
    std::vector<uint32_t> local_ids(ncut);
    for (size_t r = 0; r < ncut; ++r) local_ids[r] = cm->cut_cells[r] - (uint32_t)base;
    // displaced coordinates of the slab's node rows row_begin .. row_end
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_points, cm->pts.data() + 2 * row_begin * (Nx + 1), ctx->npoints * 2 * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_cut_cells, (ncut ? ncut : 1) * sizeof(uint32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_cell_loc, nc));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_cut_index, nc * sizeof(int32_t)));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_cut_cells, local_ids.data(), ncut * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_cell_loc, cm->cell_loc.data() + base, nc, hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_cut_index, cm->cut_index.data() + base, nc * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (!whole) {                // the interface_assembler's tables number the whole mesh: whole-mesh contexts only
        PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PA_OK;
    }
    // interface_assembler tables (cuthho_square.cpp:1142-1178): cut cells / cut faces own two blocks, so the
    // first block of an element = its plain (compressed) index + the number of cut elements before it
    const size_t nf = cm->nfaces();
    std::vector<int32_t> cell_table(nc), face_table(nf);
    std::vector<uint32_t> cut_faces;
    for (uint32_t cc : cm->cut_cells) {                     // every cut face belongs to a cut cell
        uint32_t fcs[4];
        cm->cell_face_ids(cc, fcs);
        for (int i = 0; i < 4; ++i)
            if (cm->face_loc[fcs[i]] == pa::LOC_CUT) cut_faces.push_back(fcs[i]);
    }
    std::sort(cut_faces.begin(), cut_faces.end());
    cut_faces.erase(std::unique(cut_faces.begin(), cut_faces.end()), cut_faces.end());
    pa::parallel_ranges(nc, [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; ++c)
            cell_table[c] = (int32_t)(c + (std::lower_bound(cm->cut_cells.begin(), cm->cut_cells.end(), (uint32_t)c) - cm->cut_cells.begin()));
    });
    ctx->if_num_all_cells = nc + ncut;
    pa::parallel_ranges(nf, [&](size_t f0, size_t f1) {
        for (uint32_t f = (uint32_t)f0; f < f1; ++f) {
            uint32_t lo, hi; bool dirichlet; int32_t comp;
            pa::sm_face_decode(cm->sm, f, lo, hi, dirichlet, comp);
            face_table[f] = dirichlet ? -1 : comp + (int32_t)(std::lower_bound(cut_faces.begin(), cut_faces.end(), f) - cut_faces.begin());
        }
    });
    ctx->if_num_other_faces = pa::sm_num_other_faces(cm->sm) + cut_faces.size();      // cut faces are never on the boundary
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_face_loc, nf));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_if_cell_table, nc * sizeof(int32_t)));
    PA_HIP(ctx, hipMalloc((void **)&ctx->d_if_face_table, nf * sizeof(int32_t)));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_face_loc, cm->face_loc.data(), nf, hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_if_cell_table, cell_table.data(), nc * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipMemcpyAsync(ctx->d_if_face_table, face_table.data(), nf * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

int pa_cut_query(pa_context *ctx, size_t *ncut, int8_t *cell_location, int32_t *cut_index)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (ncut) *ncut = ctx->cut->cut_cells.size();
    // (the cells of the context: of a slab of pa_cut_preprocess_rows, its own rows)
    if (cell_location) std::memcpy(cell_location, ctx->cut->cell_loc.data() + ctx->cell_base, ctx->ncells);
    if (cut_index) std::memcpy(cut_index, ctx->cut->cut_index.data() + ctx->cell_base, ctx->ncells * sizeof(int32_t));
    return PA_OK;
}

// host list building + upload of the cut quadrature of one side: once per (face degree, side)
static int ensure_cut_lists(pa_context *ctx, int face_deg, int where)
{
    if (ctx->cl[where].face_deg == face_deg && ctx->cl[where].where == where) return PA_OK;
    pa::CutLists L;
    try {
        pa::build_cut_lists(*ctx->cut, ctx->host_tab, face_deg, where, L);
    } catch (const std::invalid_argument &ex) {
        ctx->last_error = ex.what();
        return PA_ERR_QUADRATURE;
    } catch (const std::exception &ex) {
        ctx->last_error = ex.what();
        return PA_ERR_INVALID_ARG;
    }
    release_cut_lists(ctx, where);
    auto &c = ctx->cl[where];
    hipError_t e = upload_vec(L.cell_off, &c.co, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.il_off, &c.io, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.ir_off, &c.ro, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.cell_xyw, &c.cx, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.il_xyw, &c.ix, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.ir_xyw, &c.rx, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.fl_xyw, &c.fl, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.fs_xyw, &c.fs, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.fl_cnt, &c.flc, ctx->stream);
    if (e == hipSuccess) e = upload_vec(L.fs_cnt, &c.fsc, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // the host vectors go out of scope
    if (e != hipSuccess) { ctx->last_error = std::string("cut lists upload: ") + hipGetErrorString(e); return PA_ERR_HIP; }
    c.face_deg = face_deg; c.where = where;
    return PA_OK;
}

static int cut_local_ops(pa_context *ctx, int face_deg, const pa_level_set *ls, int where, int rhs_fn, int bcs_fn,
                         const double *d_rhs_vals, const double *d_bcs_vals,
                         double *d_oper, double *d_data, double *d_stab, double *d_lc, double *d_rhs, int32_t *d_info);

int pa_cut_local_ops_batch(pa_context *ctx, int face_deg, const pa_level_set *ls, int where, int rhs_fn, int bcs_fn,
                           double *d_oper, double *d_data, double *d_stab, double *d_lc, double *d_rhs, int32_t *d_info)
{
    if (rhs_fn <= PA_FN_SAMPLED || rhs_fn > PA_FN_ONE || bcs_fn <= PA_FN_SAMPLED || bcs_fn > PA_FN_ONE) return PA_ERR_INVALID_ARG;
    return cut_local_ops(ctx, face_deg, ls, where, rhs_fn, bcs_fn, nullptr, nullptr, d_oper, d_data, d_stab, d_lc, d_rhs, d_info);
}

int pa_cut_rhs_sampled_batch(pa_context *ctx, int face_deg, const pa_level_set *ls, int where, const double *d_rhs_vals,
                             const double *d_bcs_vals, double *d_rhs)
{
    if (!d_rhs_vals || !d_bcs_vals || !d_rhs) return PA_ERR_INVALID_ARG;
    return cut_local_ops(ctx, face_deg, ls, where, PA_FN_SAMPLED, PA_FN_SAMPLED, d_rhs_vals, d_bcs_vals, nullptr, nullptr, nullptr,
                         nullptr, d_rhs, nullptr);
}

int pa_cut_quadrature_points(pa_context *ctx, int face_deg, int where, int which, uint32_t *h_offsets, double *h_xyw,
                             size_t *count)
{
    if (!ctx || (where != PA_LOC_NEGATIVE && where != PA_LOC_POSITIVE) || which < 0 || which > 2) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (face_deg < 0) return PA_ERR_INVALID_DEGREE;
    if (face_deg > 2) return PA_ERR_QUADRATURE;
    pa::CutLists L;
    try {
        pa::build_cut_lists(*ctx->cut, ctx->host_tab, face_deg, where, L);
    } catch (const std::invalid_argument &ex) {
        ctx->last_error = ex.what();
        return PA_ERR_QUADRATURE;
    } catch (const std::exception &ex) {
        ctx->last_error = ex.what();
        return PA_ERR_INVALID_ARG;
    }
    const std::vector<uint32_t> &off = which == 0 ? L.cell_off : which == 1 ? L.il_off : L.ir_off;
    const std::vector<double> &xyw = which == 0 ? L.cell_xyw : which == 1 ? L.il_xyw : L.ir_xyw;
    if (count) *count = xyw.size() / 3;
    if (h_offsets) std::memcpy(h_offsets, off.data(), off.size() * sizeof(uint32_t));
    if (h_xyw) std::memcpy(h_xyw, xyw.data(), xyw.size() * sizeof(double));
    return PA_OK;
}

int pa_cut_query_tags(pa_context *ctx, int8_t *node_location, int8_t *face_location, double *points)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    const pa::CutMeshHost &cm = *ctx->cut;
    if (node_location) std::memcpy(node_location, cm.node_loc.data(), cm.npoints());
    if (face_location) std::memcpy(face_location, cm.face_loc.data(), cm.nfaces());
    if (points) std::memcpy(points, cm.pts.data(), cm.pts.size() * sizeof(double));
    return PA_OK;
}

static int cut_local_ops(pa_context *ctx, int face_deg, const pa_level_set *ls, int where, int rhs_fn, int bcs_fn,
                         const double *d_rhs_vals, const double *d_bcs_vals,
                         double *d_oper, double *d_data, double *d_stab, double *d_lc, double *d_rhs, int32_t *d_info)
{
    if (!ctx || !ls || (where != PA_LOC_NEGATIVE && where != PA_LOC_POSITIVE)) return PA_ERR_INVALID_ARG;
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (face_deg < 0) return PA_ERR_INVALID_DEGREE;
    if (face_deg > 2) return PA_ERR_QUADRATURE;            // 2*recdeg = 8 selects the empty rules[8]
    const size_t ncut = ctx->cut->cut_cells.size();
    if (ncut == 0) return PA_OK;
    int st = ensure_cut_lists(ctx, face_deg, where);
    if (st != PA_OK) return st;
    hipError_t e = hipSuccess;
    {
        const auto &c = ctx->cl[where];
        pa::CutArgs a;
        a.tab = ctx->d_tab; a.points = ctx->d_points; a.ptids = ctx->d_ptids; a.cut_cells = ctx->d_cut_cells;
        a.ncut = (uint32_t)ncut;
        a.cell_off = c.co; a.il_off = c.io; a.ir_off = c.ro;
        a.cell_xyw = c.cx; a.il_xyw = c.ix; a.ir_xyw = c.rx; a.fl_xyw = c.fl; a.fs_xyw = c.fs; a.fl_cnt = c.flc; a.fs_cnt = c.fsc;
        a.ls = pa::LevelSet{ls->kind, ls->radius, ls->alpha, ls->beta, ls->cut_y};
        a.rhs_fn = rhs_fn; a.bcs_fn = bcs_fn; a.rhs_vals = d_rhs_vals; a.bcs_vals = d_bcs_vals;
        a.eta = 5.0;                                                             // cell_eta, cuthho_square.cpp:301-306
        a.oper = d_oper; a.data = d_data; a.stab = d_stab; a.lc = d_lc; a.rhs = d_rhs; a.info = d_info;
        a.dbg = nullptr;
#ifdef PA_TUNING
        static long long *d_cut_dbg = nullptr;
        if (std::getenv("PA_CUT_CLOCK")) {
            if (!d_cut_dbg) (void)hipMalloc((void **)&d_cut_dbg, 16 * sizeof(long long));
            a.dbg = d_cut_dbg;
        }
#endif
        // one wavefront per cut cell and a long serial chain per cell: as many blocks as the chip holds (2 per SIMD),
        // so that a few thousand cut cells take ONE cell's latency, not two or three
        size_t cap_blocks = (size_t)ctx->num_cus * 8;
#ifdef PA_TUNING
        if (const char *env = std::getenv("PA_CUT_BLOCKS_PER_CU")) { const int v = std::atoi(env); if (v > 0) cap_blocks = (size_t)ctx->num_cus * v; }
#endif
        const int grid = (int)(ncut < cap_blocks ? ncut : cap_blocks);
        // With pa_context_set_cut_overlap the kernel goes to the side stream, after everything enqueued on the
        // context's stream so far (the previous merge reads the buffers it writes); pa_cut_merge joins it.
        hipStream_t st_ = ctx->stream;
        if (ctx->cut_overlap && ctx->side) {
            e = hipEventRecord(ctx->ev_main, ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->side, ctx->ev_main, 0);
            st_ = ctx->side;
        }
        bool cut_dd = true;           // stages A-E in double-double (cut_device.hpp); the all-double form is an A/B of tuning builds
#ifdef PA_TUNING
        if (const char *env = std::getenv("PA_CUT_DOUBLE")) cut_dd = std::atoi(env) == 0;
#endif
        if (e == hipSuccess) {
            if (cut_dd) {
                switch (face_deg) {
                case 0: hipLaunchKernelGGL((pa::cut_local_ops_kernel<0, true>), dim3(grid), dim3(64), 0, st_, a); break;
                case 1: hipLaunchKernelGGL((pa::cut_local_ops_kernel<1, true>), dim3(grid), dim3(64), 0, st_, a); break;
                default: hipLaunchKernelGGL((pa::cut_local_ops_kernel<2, true>), dim3(grid), dim3(64), 0, st_, a); break;
                }
            } else {
                switch (face_deg) {
                case 0: hipLaunchKernelGGL((pa::cut_local_ops_kernel<0, false>), dim3(grid), dim3(64), 0, st_, a); break;
                case 1: hipLaunchKernelGGL((pa::cut_local_ops_kernel<1, false>), dim3(grid), dim3(64), 0, st_, a); break;
                default: hipLaunchKernelGGL((pa::cut_local_ops_kernel<2, false>), dim3(grid), dim3(64), 0, st_, a); break;
                }
            }
            e = hipGetLastError();
        }
#ifdef PA_TUNING
        if (a.dbg != nullptr && e == hipSuccess) {
            long long h[16] = {0};
            (void)hipStreamSynchronize(st_);
            (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "PA_CUT_CLOCK block 0 (last cell it worked on), clocks per stage: A %lld stiff %lld B %lld C %lld D %lld solve %lld E %lld F %lld H %lld\n",
                         h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[7], h[9] - h[8]);
        }
#endif
        if (e == hipSuccess && st_ != ctx->stream) {
            e = hipEventRecord(ctx->ev_side, ctx->side);
            ctx->side_pending = true;
        }
    }
    if (e != hipSuccess) { ctx->last_error = std::string("pa_cut_local_ops_batch: ") + hipGetErrorString(e); return PA_ERR_HIP; }
    return PA_OK;
}

int pa_cut_uncut_rhs_batch(pa_context *ctx, int degree, int where, int fn, double *d_rhs)
{
    if (!ctx || !d_rhs || degree < 0 || (where != PA_LOC_NEGATIVE && where != PA_LOC_POSITIVE)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut || !ctx->d_points || !ctx->d_cell_loc) return PA_ERR_NO_MESH;
    if (fn <= PA_FN_SAMPLED || fn > PA_FN_ONE) return PA_ERR_INVALID_ARG;
    const int qdeg = 2 * degree;                                   // utils.hpp:165 with di = 0 (cuthho_square.cpp:631)
    int nqp = 0;
    const int st = rhs_quadrature(ctx, qdeg, PA_QUAD_FAN, &nqp);
    if (st != PA_OK) return st;
    const size_t n = ctx->ncells;
    if (n == 0) return PA_OK;
    return launch_rhs<pa::QUAD_FAN>(ctx, degree, qdeg, nqp, fn, nullptr, 0, n, d_rhs, ctx->d_cell_loc, where);
}

int pa_cut_merge(pa_context *ctx, int face_deg, int where, const double *d_cut_lc, const double *d_cut_rhs, double *d_lc,
                 double *d_rhs)
{
    if (!ctx || face_deg < 0 || face_deg > 2 || (where != PA_LOC_NEGATIVE && where != PA_LOC_POSITIVE)) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    const int cbs = pa::P2(face_deg + 1), ms = cbs + 4 * (face_deg + 1);
    const uint32_t nc = (uint32_t)ctx->ncells;
    const uint32_t ncut = (uint32_t)ctx->cut->cut_cells.size();
    if (ctx->side_pending) {                              // the cut cells' kernel ran on the side stream
        PA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        ctx->side_pending = false;
    }
    if (d_rhs != nullptr && nc) {
        const size_t total = (size_t)nc * (size_t)cbs;
        hipLaunchKernelGGL(pa::cut_zero_rhs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, total, (uint32_t)cbs,
                           ctx->d_cell_loc, where, d_rhs);
    }
    if (ncut)
        hipLaunchKernelGGL(pa::cut_merge_cells_kernel, dim3(ncut), dim3(64), 0, ctx->stream, ncut, ctx->d_cut_cells, ms * ms, cbs, d_cut_lc,
                           d_cut_rhs, d_lc, d_rhs);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_cut_merge_condensed(pa_context *ctx, int face_deg, const double *d_cut_Sp, const double *d_cut_g, double *d_cond)
{
    if (!ctx || !d_cond || face_deg < 0 || face_deg > 2) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    const uint32_t ncut = (uint32_t)ctx->cut->cut_cells.size();
    if (ncut == 0) return PA_OK;
    if (!d_cut_Sp || !d_cut_g) return PA_ERR_INVALID_ARG;
    if (ctx->side_pending) {                              // the cut cells' kernel ran on the side stream
        PA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        ctx->side_pending = false;
    }
    const int nf = 4 * (face_deg + 1), ntri = nf * (nf + 1) / 2;
    hipLaunchKernelGGL(pa::cut_merge_condensed_kernel, dim3(ncut), dim3(64), 0, ctx->stream, ncut, ctx->d_cut_cells, ntri, nf, d_cut_Sp, d_cut_g,
                       d_cond);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

// ---- two-sided interface problem -------------------------------------------------------------
static int interface_checks(pa_context *ctx, int face_deg)
{
    if (!ctx) return PA_ERR_INVALID_ARG;
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (face_deg < 0) return PA_ERR_INVALID_DEGREE;
    if (face_deg > 2) return PA_ERR_QUADRATURE;            // 2*recdeg = 8 selects the empty rules[8]
    return PA_OK;
}

int pa_cut_interface_ops_batch(pa_context *ctx, int face_deg, const pa_level_set *ls, const pa_interface_params *parms,
                               int rhs_fn, double *d_oper, double *d_data, double *d_lc, double *d_rhs, int32_t *d_info)
{
    int st = interface_checks(ctx, face_deg);
    if (st != PA_OK) return st;
    if (!ls || !parms || rhs_fn <= PA_FN_SAMPLED || rhs_fn > PA_FN_ONE) return PA_ERR_INVALID_ARG;
    const size_t ncut = ctx->cut->cut_cells.size();
    if (ncut == 0) return PA_OK;
    for (int side = 0; side < 2; ++side) {
        st = ensure_cut_lists(ctx, face_deg, side);
        if (st != PA_OK) return st;
    }
    const int cbs = pa::P2(face_deg + 1), nfd = 4 * (face_deg + 1), ms = cbs + nfd, m2 = 2 * ms;
    double *data = d_data, *stab_n = nullptr, *stab_p = nullptr;      // [data | stab_n | stab_p] when lc is requested
    if (d_lc) {
        // (the scratch lives in the context: a hipMalloc / hipFree pair and a stream synchronisation per call cost more than the
        // kernels of a 512 x 512 mesh's cut cells; everything that touches it is ordered on the context's stream)
        const size_t need = (d_data ? 0 : ncut * (size_t)m2 * m2) + 2 * ncut * (size_t)ms * ms;
        if (ctx->if_scratch_cap < need) {
            if (ctx->d_if_scratch) { PA_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_if_scratch); ctx->d_if_scratch = nullptr; ctx->if_scratch_cap = 0; }
            PA_HIP(ctx, hipMalloc((void **)&ctx->d_if_scratch, need * sizeof(double)));
            ctx->if_scratch_cap = need;
        }
        double *p = ctx->d_if_scratch;
        if (!d_data) { data = p; p += ncut * (size_t)m2 * m2; }
        stab_n = p; stab_p = p + ncut * (size_t)ms * ms;
        // make_hho_cut_stabilization of both sides through the fictitious-domain kernel (stabilization only: its stages A-E are skipped)
        st = pa_cut_local_ops_batch(ctx, face_deg, ls, PA_LOC_NEGATIVE, PA_FN_ONE, PA_FN_ONE, nullptr, nullptr, stab_n, nullptr, nullptr, nullptr);
        if (st == PA_OK)
            st = pa_cut_local_ops_batch(ctx, face_deg, ls, PA_LOC_POSITIVE, PA_FN_ONE, PA_FN_ONE, nullptr, nullptr, stab_p, nullptr, nullptr, nullptr);
        if (st != PA_OK) return st;
    }
    pa::CutInterfaceArgs a;
    a.points = ctx->d_points; a.ptids = ctx->d_ptids; a.cut_cells = ctx->d_cut_cells; a.ncut = (uint32_t)ncut;
    for (int side = 0; side < 2; ++side) {
        a.cell_off[side] = ctx->cl[side].co; a.cell_xyw[side] = ctx->cl[side].cx;
        a.fl_xyw[side] = ctx->cl[side].fl; a.fl_cnt[side] = ctx->cl[side].flc;
    }
    a.il_off = ctx->cl[0].io; a.il_xyw = ctx->cl[0].ix;     // integrate_interface(.., IN_NEGATIVE_SIDE) (:437)
    a.ls = pa::LevelSet{ls->kind, ls->radius, ls->alpha, ls->beta, ls->cut_y};
    a.rhs_fn = rhs_fn; a.kappa[0] = parms->kappa_1; a.kappa[1] = parms->kappa_2; a.eta = parms->eta;
    a.oper = d_oper; a.data = data; a.rhs = d_rhs; a.info = d_info;
    // (46 KB of LDS per block at k = 2: three blocks per compute unit are resident)
    const int grid = (int)(ncut < (size_t)ctx->num_cus * 3 ? ncut : (size_t)ctx->num_cus * 3);
    switch (face_deg) {
    case 0: hipLaunchKernelGGL((pa::cut_interface_kernel<0>), dim3(grid), dim3(64), 0, ctx->stream, a); break;
    case 1: hipLaunchKernelGGL((pa::cut_interface_kernel<1>), dim3(grid), dim3(64), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((pa::cut_interface_kernel<2>), dim3(grid), dim3(64), 0, ctx->stream, a); break;
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && d_lc) {
        hipLaunchKernelGGL(pa::cut_interface_lc_kernel, dim3(grid), dim3(256), 0, ctx->stream, (uint32_t)ncut, cbs, nfd,
                           parms->kappa_1, parms->kappa_2, data, stab_n, stab_p, d_lc);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { ctx->last_error = std::string("pa_cut_interface_ops_batch: ") + hipGetErrorString(e); return PA_ERR_HIP; }
    return PA_OK;
}

int pa_cut_interface_uncut_batch(pa_context *ctx, int face_deg, const pa_interface_params *parms, int rhs_fn, double *d_lc,
                                 double *d_rhs, int32_t *d_info)
{
    int st = interface_checks(ctx, face_deg);
    if (st != PA_OK) return st;
    if (!parms || (d_rhs && (rhs_fn <= PA_FN_SAMPLED || rhs_fn > PA_FN_ONE))) return PA_ERR_INVALID_ARG;
    const pa_degree_info di = {face_deg + 1, face_deg, face_deg + 1};
    const size_t n = ctx->ncells;
    const int ms = pa::P2(face_deg + 1) + 4 * (face_deg + 1), mm = ms * ms;
    if (d_lc) {
        if (parms->kappa_1 == 1.0 && parms->kappa_2 == 1.0) {
            st = pa_local_ops_batch(ctx, di, PA_QUAD_FAN, PA_STAB_NAIVE, 0, n, nullptr, nullptr, nullptr, d_lc, d_info);
            if (st != PA_OK) return st;
        } else {
            double *scratch = nullptr;
            PA_HIP(ctx, hipMalloc((void **)&scratch, 2 * n * (size_t)mm * sizeof(double)));
            st = pa_local_ops_batch(ctx, di, PA_QUAD_FAN, PA_STAB_NAIVE, 0, n, nullptr, scratch, scratch + n * (size_t)mm, nullptr, d_info);
            hipError_t e = hipSuccess;
            if (st == PA_OK) {
                const size_t total = n * (size_t)mm;
                hipLaunchKernelGGL(pa::cut_interface_uncut_lc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, n, mm,
                                   ctx->d_cell_loc, parms->kappa_1, parms->kappa_2, scratch, scratch + n * (size_t)mm, d_lc);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            }
            (void)hipFree(scratch);
            if (st != PA_OK) return st;
            PA_HIP(ctx, e);
        }
    }
    if (d_rhs) {
        st = pa_cell_rhs_batch(ctx, face_deg + 1, 0, PA_QUAD_FAN, rhs_fn, nullptr, 0, n, d_rhs);
        if (st != PA_OK) return st;
    }
    return PA_OK;
}

int pa_interface_assembler_query(pa_context *ctx, int face_deg, pa_interface_info *out)
{
    if (!ctx || !out || face_deg < 0 || face_deg > 3) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (!ctx->d_if_cell_table) {
        ctx->last_error = "the interface_assembler's numbering covers the whole mesh: not available on a slab of pa_cut_preprocess_rows";
        return PA_ERR_INVALID_ARG;
    }
    out->num_all_cells = ctx->if_num_all_cells;
    out->num_other_faces = ctx->if_num_other_faces;
    out->system_size = (uint64_t)pa::P2(face_deg + 1) * ctx->if_num_all_cells + (uint64_t)(face_deg + 1) * ctx->if_num_other_faces;
    out->ncut = ctx->cut->cut_cells.size();
    return PA_OK;
}

int pa_interface_triplets_batch(pa_context *ctx, int face_deg, const double *d_lc, const double *d_rhs, const double *d_g,
                                const double *d_lc_cut, const double *d_rhs_cut, int32_t *d_rows, int32_t *d_cols,
                                double *d_vals, int32_t *d_rows_cut, int32_t *d_cols_cut, double *d_vals_cut,
                                int32_t *d_rhs_rows, double *d_rhs_vals, int32_t *d_rhs_rows_cut, double *d_rhs_vals_cut)
{
    if (!ctx || !d_lc || !d_rows || !d_cols || !d_vals || !d_rhs_rows || !d_rhs_vals) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (face_deg < 0 || face_deg > 3) return PA_ERR_INVALID_DEGREE;
    if (!ctx->cut || !ctx->d_cell_faces) return PA_ERR_NO_MESH;
    if (!ctx->d_if_cell_table) {
        ctx->last_error = "the interface_assembler's numbering covers the whole mesh: not available on a slab of pa_cut_preprocess_rows";
        return PA_ERR_INVALID_ARG;
    }
    const size_t ncut = ctx->cut->cut_cells.size();
    if (ncut && (!d_lc_cut || !d_rows_cut || !d_cols_cut || !d_vals_cut || !d_rhs_rows_cut || !d_rhs_vals_cut)) return PA_ERR_INVALID_ARG;
    pa_interface_info info;
    pa_interface_assembler_query(ctx, face_deg, &info);
    if (info.system_size >= ((uint64_t)1 << 31)) return PA_ERR_INVALID_ARG;      // Eigen::Triplet stores int indices
    pa::InterfaceTripletArgs a;
    a.cell_faces = ctx->d_cell_faces; a.cell_loc = ctx->d_cell_loc; a.face_loc = ctx->d_face_loc; a.cut_index = ctx->d_cut_index;
    a.cell_table = ctx->d_if_cell_table; a.face_table = ctx->d_if_face_table; a.g = d_g;
    a.lc = d_lc; a.rhs = d_rhs; a.lc_cut = d_lc_cut; a.rhs_cut = d_rhs_cut;
    a.ncells = ctx->ncells; a.num_all_cells = ctx->if_num_all_cells;
    a.cbs = pa::P2(face_deg + 1); a.fbs = face_deg + 1;
    a.rows = d_rows; a.cols = d_cols; a.vals = d_vals; a.rows_cut = d_rows_cut; a.cols_cut = d_cols_cut; a.vals_cut = d_vals_cut;
    a.rhs_rows = d_rhs_rows; a.rhs_vals = d_rhs_vals; a.rhs_rows_cut = d_rhs_rows_cut; a.rhs_vals_cut = d_rhs_vals_cut;
    const int m2 = 2 * (a.cbs + 4 * a.fbs);
    const size_t shmem = m2 * sizeof(double) + m2 * sizeof(int32_t);
    const size_t resident = (size_t)ctx->num_cus * 8;
    const int grid = (int)(ctx->ncells < resident ? ctx->ncells : resident);
    hipLaunchKernelGGL(pa::interface_triplets_kernel, dim3(grid), dim3(256), shmem, ctx->stream, a);
    PA_HIP(ctx, hipGetLastError());
    return PA_OK;
}

int pa_interface_cell_offsets(pa_context *ctx, int face_deg, int64_t *d_offsets)
{
    if (!ctx || !d_offsets || face_deg < 0 || face_deg > 3) return PA_ERR_INVALID_ARG;
    (void)hipSetDevice(ctx->device);
    if (!ctx->cut) return PA_ERR_NO_MESH;
    if (!ctx->d_if_cell_table) {
        ctx->last_error = "the interface_assembler's numbering covers the whole mesh: not available on a slab of pa_cut_preprocess_rows";
        return PA_ERR_INVALID_ARG;
    }
    const pa::CutMeshHost &cm = *ctx->cut;
    const size_t nc = cm.ncells();
    const int64_t cbs = pa::P2(face_deg + 1);
    std::vector<int64_t> off(2 * nc);
    int64_t blocks = 0;
    for (size_t c = 0; c < nc; ++c) {                      // :1368-1379
        const bool cut = cm.cell_loc[c] == pa::LOC_CUT;
        off[2 * c] = blocks * cbs;
        off[2 * c + 1] = cut ? (blocks + 1) * cbs : blocks * cbs;
        blocks += cut ? 2 : 1;
    }
    PA_HIP(ctx, hipMemcpyAsync(d_offsets, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PA_OK;
}

}  // extern "C"
