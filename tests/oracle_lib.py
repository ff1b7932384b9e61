"""ctypes binding of the CPU oracle (oracle/libhho_oracle.so).

Test infrastructure only: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

QUAD_TENSOR, QUAD_FAN = 0, 1
STAB_NONE, STAB_NAIVE, STAB_FANCY = 0, 1, 2


class Degrees(C.Structure):
    _fields_ = [("cell_deg", C.c_int), ("face_deg", C.c_int), ("rec_deg", C.c_int)]

    @property
    def rbs(self):
        return (self.rec_deg + 2) * (self.rec_deg + 1) // 2

    @property
    def cbs(self):
        return (self.cell_deg + 2) * (self.cell_deg + 1) // 2

    @property
    def fbs(self):
        return self.face_deg + 1

    @property
    def msize(self):
        return self.cbs + 4 * self.fbs


class MeshParams(C.Structure):
    _fields_ = [("Nx", C.c_size_t), ("Ny", C.c_size_t),
                ("min_x", C.c_double), ("max_x", C.c_double),
                ("min_y", C.c_double), ("max_y", C.c_double)]


SCALAR_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_void_p)

CUT_NEG, CUT_POS, CUT_ON_INTERFACE = 0, 1, 2


class CutParams(C.Structure):
    _fields_ = [("kappa_1", C.c_double), ("kappa_2", C.c_double), ("eta", C.c_double)]


class CutLevelSet(C.Structure):
    _fields_ = [("kind", C.c_int), ("radius", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("cut_y", C.c_double)]


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libhho_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, n) for n in ("hho_oracle.c", "hho_oracle.h", "cuthho_oracle.c", "cuthho_oracle.h", "Makefile")]
    if not os.path.exists(path) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        build()
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    u64p = C.POINTER(C.c_uint64)
    L.hho_degree_info1.restype = Degrees
    L.hho_degree_info1.argtypes = [C.c_int]
    L.hho_degree_info2.restype = Degrees
    L.hho_degree_info2.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.hho_iexp_pow.restype = C.c_double
    L.hho_iexp_pow.argtypes = [C.c_double, C.c_size_t]
    L.hho_gauss_legendre.argtypes = [C.c_int, dp, dp]
    L.hho_triangle_quadrature.argtypes = [dp, dp, dp, C.c_int, dp, dp, dp]
    L.hho_cell_barycenter.argtypes = [dp, dp]
    L.hho_cell_barycenter.restype = None
    L.hho_cell_diameter.argtypes = [dp]
    L.hho_cell_diameter.restype = C.c_double
    L.hho_cell_measure.argtypes = [dp]
    L.hho_cell_measure.restype = C.c_double
    L.hho_cell_normals.argtypes = [dp, dp]
    L.hho_cell_normals.restype = None
    L.hho_cell_quadrature.argtypes = [dp, C.c_int, C.c_int, dp, dp, dp]
    L.hho_face_quadrature.argtypes = [dp, dp, C.c_int, dp, dp, dp]
    L.hho_cell_basis_eval.argtypes = [dp, C.c_double, C.c_int, C.c_double, C.c_double, dp]
    L.hho_cell_basis_eval.restype = None
    L.hho_cell_basis_grad.argtypes = [dp, C.c_double, C.c_int, C.c_double, C.c_double, dp, dp]
    L.hho_cell_basis_grad.restype = None
    L.hho_face_basis_eval.argtypes = [dp, dp, C.c_int, C.c_double, C.c_double, dp]
    L.hho_face_basis_eval.restype = None
    L.hho_cell_face_points.argtypes = [dp, u64p, C.c_int, dp, dp]
    L.hho_cell_face_points.restype = None
    L.hho_make_laplacian.argtypes = [dp, u64p, Degrees, C.c_int, dp, dp]
    L.hho_make_naive_stabilization.argtypes = [dp, u64p, Degrees, dp]
    L.hho_make_fancy_stabilization.argtypes = [dp, u64p, Degrees, C.c_int, dp, dp]
    L.hho_cell_mass_matrix.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]
    L.hho_face_mass_matrix.argtypes = [dp, dp, C.c_int, C.c_int, dp]
    L.hho_cell_rhs.argtypes = [dp, C.c_int, C.c_int, C.c_int, SCALAR_FN, C.c_void_p, dp]
    L.hho_face_rhs.argtypes = [dp, dp, C.c_int, C.c_int, SCALAR_FN, C.c_void_p, dp]
    L.hho_project_function.argtypes = [dp, u64p, Degrees, C.c_int, SCALAR_FN, C.c_void_p, C.c_int, dp]
    L.hho_llt_factor.argtypes = [dp, C.c_int]
    L.hho_llt_solve_inplace.argtypes = [dp, C.c_int, dp, C.c_int]
    L.hho_llt_solve_inplace.restype = None
    L.hho_static_condensation.argtypes = [dp, dp, C.c_int, C.c_int, dp, dp, dp]
    mpp = C.POINTER(MeshParams)
    for name in ("hho_mesh_num_points", "hho_mesh_num_cells", "hho_mesh_num_faces"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [mpp]
    L.hho_mesh_generate.argtypes = [mpp, dp, u64p]
    L.hho_mesh_generate.restype = None
    L.hho_mesh_generate_faces.argtypes = [mpp, u64p, C.POINTER(C.c_uint8)]
    L.hho_mesh_generate_faces.restype = None
    L.hho_mesh_face_id.argtypes = [mpp, C.c_size_t, C.c_size_t, C.c_int]
    L.hho_mesh_face_id.restype = C.c_size_t
    L.hho_mesh_face_is_boundary.argtypes = [mpp, C.c_size_t, C.c_size_t, C.c_int]
    L.hho_local_ops_batch.argtypes = [dp, u64p, C.c_size_t, C.c_size_t, Degrees, C.c_int, C.c_int,
                                      SCALAR_FN, C.c_void_p, C.c_int, dp, dp, dp, dp, dp]
    L.hho_max_threads.restype = C.c_int
    L.hho_local_ops_batch_mt.argtypes = [dp, u64p, C.c_size_t, C.c_size_t, Degrees, C.c_int, C.c_int,
                                         SCALAR_FN, C.c_void_p, C.c_int, dp, dp, C.c_int]
    L.hho_matrix_assembly_timed.argtypes = [mpp, C.c_size_t, C.c_size_t, Degrees, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, dp, C.POINTER(C.c_size_t), dp]
    i32p = C.POINTER(C.c_int32)
    L.hho_set_from_triplets.restype = C.c_size_t
    L.hho_set_from_triplets.argtypes = [C.c_size_t, i32p, i32p, dp, C.c_size_t, C.POINTER(C.c_int64), i32p, dp, C.c_int]
    i64p, i32p, u8p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.hho_assembler_compress_table.restype = C.c_size_t
    L.hho_assembler_compress_table.argtypes = [u8p, C.c_size_t, i64p]
    L.hho_assembler_system_size.restype = C.c_size_t
    L.hho_assembler_system_size.argtypes = [Degrees, C.c_size_t, C.c_size_t]
    L.hho_dirichlet_face_data.argtypes = [dp, dp, C.c_int, SCALAR_FN, C.c_void_p, dp]
    L.hho_assembler_assemble_cell.argtypes = [Degrees, C.c_size_t, C.c_size_t, u64p, u8p, i64p, dp, dp, dp,
                                              i32p, i32p, dp, C.POINTER(C.c_size_t), i64p, dp]
    L.hho_obstacle_tables.restype = None
    L.hho_obstacle_tables.argtypes = [u8p, C.c_size_t, i64p, i64p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.hho_obstacle_assemble_cell.argtypes = [Degrees, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, u64p, u8p, i64p,
                                             u8p, i64p, i64p, dp, dp, dp, dp,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, C.POINTER(C.c_size_t), i64p, dp]
    L.hho_obstacle_expand_solution.restype = None
    L.hho_obstacle_expand_solution.argtypes = [Degrees, C.c_size_t, C.c_size_t, u8p, i64p, u8p, dp, dp, dp, dp, dp]
    L.hho_obstacle_take_local_data.restype = None
    L.hho_obstacle_take_local_data.argtypes = [Degrees, C.c_size_t, C.c_size_t, u64p, dp, dp]
    L.hho_builtin_fn.restype = SCALAR_FN
    L.hho_builtin_fn.argtypes = [C.c_int]
    # ---- cuthho_oracle.h
    lsp = C.POINTER(CutLevelSet)
    L.cut_ls_eval.restype = C.c_double
    L.cut_ls_eval.argtypes = [lsp, C.c_double, C.c_double]
    L.cut_mesh_create.restype = C.c_void_p
    L.cut_mesh_create.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double]
    L.cut_mesh_free.argtypes = [C.c_void_p]
    L.cut_mesh_free.restype = None
    L.cut_mesh_preprocess.argtypes = [C.c_void_p, lsp, C.c_int]
    L.cut_mesh_preprocess_agglomeration.argtypes = [C.c_void_p, lsp, C.c_int]
    L.cut_mesh_agglo_set.restype = None
    L.cut_mesh_agglo_set.argtypes = [C.c_void_p, C.POINTER(C.c_int8)]
    L.cut_mesh_neighbors.restype = None
    L.cut_mesh_neighbors.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    for name in ("cut_mesh_num_points", "cut_mesh_num_cells", "cut_mesh_num_faces", "cut_mesh_interface_points"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [C.c_void_p]
    for name, rt in (("cut_mesh_points", dp), ("cut_mesh_cell_ptids", u64p), ("cut_mesh_faces", u64p),
                     ("cut_mesh_face_boundary", u8p), ("cut_mesh_node_location", C.POINTER(C.c_int8)),
                     ("cut_mesh_face_location", C.POINTER(C.c_int8)), ("cut_mesh_face_intersection", dp),
                     ("cut_mesh_cell_location", C.POINTER(C.c_int8))):
        getattr(L, name).restype = rt
        getattr(L, name).argtypes = [C.c_void_p]
    L.cut_mesh_cell_interface.restype = dp
    L.cut_mesh_cell_interface.argtypes = [C.c_void_p, C.c_size_t]
    L.cut_mesh_cell_face.restype = C.c_size_t
    L.cut_mesh_cell_face.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.cut_cell_quadrature.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, dp, dp, dp, C.c_int]
    L.cut_face_quadrature.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, dp, dp, dp, C.c_int]
    L.cut_interface_quadrature.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, dp, dp, dp, C.c_int]
    L.cut_cell_measure.restype = C.c_double
    L.cut_cell_measure.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.cut_make_hho_laplacian.argtypes = [C.c_void_p, lsp, C.c_size_t, Degrees, C.c_int, dp, dp, C.POINTER(C.c_int)]
    L.cut_make_hho_cut_stabilization.argtypes = [C.c_void_p, C.c_size_t, Degrees, C.c_int, dp]
    L.cut_make_hho_laplacian_interface.argtypes = [C.c_void_p, lsp, C.c_size_t, Degrees, C.POINTER(CutParams), dp, dp]
    L.cut_make_rhs_side.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, SCALAR_FN, C.c_void_p, dp]
    L.cut_interface_tables.restype = None
    L.cut_interface_tables.argtypes = [C.c_void_p, i64p, i64p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cut_interface_assemble.argtypes = [C.c_void_p, Degrees, C.c_size_t, i64p, i64p, C.c_size_t, dp, dp, dp,
                                         C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, C.POINTER(C.c_size_t), i64p, dp]
    L.cut_interface_cell_offset.restype = C.c_size_t
    L.cut_interface_cell_offset.argtypes = [C.c_void_p, Degrees, C.c_size_t, i64p, C.c_int]
    L.cut_make_rhs.argtypes = [C.c_void_p, lsp, C.c_size_t, C.c_int, C.c_int, SCALAR_FN, SCALAR_FN, C.c_void_p, dp]
    # ---- cut_truth.c (binary128 evaluation of the cut operators)
    L.cut_truth_laplacian.argtypes = [C.c_void_p, lsp, C.c_size_t, Degrees, C.c_int, dp, dp]
    L.cut_truth_stabilization.argtypes = [C.c_void_p, C.c_size_t, Degrees, C.c_int, dp]
    L.cut_truth_rhs.argtypes = [C.c_void_p, lsp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    L.cut_truth_laplacian_interface.argtypes = [C.c_void_p, lsp, C.c_size_t, Degrees, C.POINTER(CutParams), dp, dp]
    L.cut_truth_rhs_side.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, dp]
    L.cut_truth_last_interface_cond.restype = C.c_double
    L.cut_truth_last_interface_cond.argtypes = []
    L.cut_truth_round_inputs.restype = None
    L.cut_truth_round_inputs.argtypes = [C.c_int]
    L.cut_truth_laplacian_cond.restype = C.c_double
    L.cut_truth_laplacian_cond.argtypes = [C.c_void_p, lsp, C.c_size_t, Degrees, C.c_int]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def degrees(cd, fd):
    fb = C.c_int(0)
    d = lib().hho_degree_info2(cd, fd, C.byref(fb))
    return d


def _prep(pts, ids):
    pts = np.ascontiguousarray(np.asarray(pts, dtype=np.float64).reshape(8))
    ids = np.ascontiguousarray(np.asarray(ids, dtype=np.uint64).reshape(4))
    return pts, ids


def make_laplacian(pts, ids, di, quad=QUAD_TENSOR):
    """-> (status, oper[(rbs-1) x msize], data[msize x msize]) as numpy (row,col) arrays."""
    pts, ids = _prep(pts, ids)
    nr, ms = di.rbs - 1, di.msize
    oper = np.zeros((ms, nr))          # column-major storage == transposed C array
    data = np.zeros((ms, ms))
    st = lib().hho_make_laplacian(_dp(pts), _u64p(ids), di, quad, _dp(oper), _dp(data))
    return st, oper.T.copy(), data.T.copy()


def make_naive_stabilization(pts, ids, di):
    pts, ids = _prep(pts, ids)
    ms = di.msize
    stab = np.zeros((ms, ms))
    st = lib().hho_make_naive_stabilization(_dp(pts), _u64p(ids), di, _dp(stab))
    return st, stab.T.copy()


def make_fancy_stabilization(pts, ids, di, oper, quad=QUAD_TENSOR):
    pts, ids = _prep(pts, ids)
    ms = di.msize
    stab = np.zeros((ms, ms))
    R = np.ascontiguousarray(np.asarray(oper, dtype=np.float64).T)   # to column-major
    st = lib().hho_make_fancy_stabilization(_dp(pts), _u64p(ids), di, quad, _dp(R), _dp(stab))
    return st, stab.T.copy()


def cell_rhs(pts, quad, degree, di, fn):
    pts = np.ascontiguousarray(np.asarray(pts, dtype=np.float64).reshape(8))
    n = (degree + 2) * (degree + 1) // 2
    out = np.zeros(n)
    cb = SCALAR_FN(lambda x, y, u: fn(x, y))
    st = lib().hho_cell_rhs(_dp(pts), quad, degree, di, cb, None, _dp(out))
    return st, out


def project_function(pts, ids, di, fn, quad=QUAD_TENSOR, dinc=0):
    pts, ids = _prep(pts, ids)
    out = np.zeros(di.msize)
    cb = SCALAR_FN(lambda x, y, u: fn(x, y))
    st = lib().hho_project_function(_dp(pts), _u64p(ids), di, quad, cb, None, dinc, _dp(out))
    return st, out


def static_condensation(lc, f, cbs):
    ms = lc.shape[0]
    nf = ms - cbs
    lcc = np.ascontiguousarray(lc.T)
    f = np.ascontiguousarray(np.asarray(f, dtype=np.float64))
    S = np.zeros((nf, nf))
    g = np.zeros(nf)
    rec = np.zeros((nf + 1, cbs))
    st = lib().hho_static_condensation(_dp(lcc), _dp(f), cbs, nf, _dp(S), _dp(g), _dp(rec))
    return st, S.T.copy(), g, rec.T.copy()


def make_mesh(Nx, Ny, lo=(0.0, 0.0), hi=(1.0, 1.0)):
    """-> (params, points[np,2], cell_ptids[nc,4] uint64)"""
    mp = MeshParams(Nx, Ny, lo[0], hi[0], lo[1], hi[1])
    L = lib()
    npts, nc = L.hho_mesh_num_points(C.byref(mp)), L.hho_mesh_num_cells(C.byref(mp))
    points = np.zeros((npts, 2))
    ptids = np.zeros((nc, 4), dtype=np.uint64)
    L.hho_mesh_generate(C.byref(mp), _dp(points), _u64p(ptids))
    return mp, points, ptids


def mesh_faces(mp):
    L = lib()
    nf = L.hho_mesh_num_faces(C.byref(mp))
    faces = np.zeros((nf, 2), dtype=np.uint64)
    bnd = np.zeros(nf, dtype=np.uint8)
    L.hho_mesh_generate_faces(C.byref(mp), _u64p(faces), bnd.ctypes.data_as(C.POINTER(C.c_uint8)))
    return faces, bnd


def local_ops_batch(points, ptids, di, quad, stab, first=0, n=None, fn=None, rhs_di=0,
                    want=("lc",)):
    """Run the oracle's batched per-cell loop; returns dict of cell-major numpy arrays
    with matrices in (row, col) orientation."""
    L = lib()
    points = np.ascontiguousarray(points, dtype=np.float64)
    ptids = np.ascontiguousarray(ptids, dtype=np.uint64)
    if n is None:
        n = ptids.shape[0] - first
    ms, nr, cbs = di.msize, di.rbs - 1, di.cbs
    bufs = {}
    ptr = {}
    for key, shape in (("oper", (n, ms, nr)), ("data", (n, ms, ms)), ("stab", (n, ms, ms)),
                       ("lc", (n, ms, ms))):
        if key in want:
            bufs[key] = np.zeros(shape)
            ptr[key] = _dp(bufs[key])
        else:
            ptr[key] = None
    if fn is None:
        cb = C.cast(None, SCALAR_FN)
    elif isinstance(fn, int):
        cb = L.hho_builtin_fn(fn)          # C function: no Python callback per quadrature point
    else:
        cb = SCALAR_FN(lambda x, y, u: fn(x, y))
    rhs = np.zeros((n, cbs)) if fn is not None else None
    st = L.hho_local_ops_batch(_dp(points), _u64p(ptids), first, n, di, quad, stab, cb, None, rhs_di,
                               ptr["oper"], ptr["data"], ptr["stab"], ptr["lc"],
                               _dp(rhs) if rhs is not None else None)
    out = {k: np.ascontiguousarray(np.swapaxes(v, 1, 2)) for k, v in bufs.items()}
    if rhs is not None:
        out["rhs"] = rhs
    return st, out


class Assembler:
    """assembler<Mesh> (hho.hpp:252-463) on the generator mesh, through the oracle's C restatement."""

    def __init__(self, mp, points, ptids, di, bf_id=None):
        L = lib()
        self.mp, self.points, self.ptids, self.di = mp, points, ptids, di
        self.faces, self.bnd = mesh_faces(mp)
        self.nf, self.nc = self.faces.shape[0], ptids.shape[0]
        self.is_dir = np.ascontiguousarray(self.bnd.astype(np.uint8))        # all boundary faces are Dirichlet
        self.compress = np.zeros(self.nf, dtype=np.int64)
        self.num_other = L.hho_assembler_compress_table(self.is_dir.ctypes.data_as(C.POINTER(C.c_uint8)), self.nf,
                                                        self.compress.ctypes.data_as(C.POINTER(C.c_int64)))
        self.system_size = L.hho_assembler_system_size(di, self.nc, self.num_other)
        Nx = mp.Nx
        self.cell_faces = np.zeros((self.nc, 4), dtype=np.uint64)
        for c in range(self.nc):
            for lf in range(4):
                self.cell_faces[c, lf] = L.hho_mesh_face_id(C.byref(mp), c % Nx, c // Nx, lf)
        self.g = np.zeros((self.nf, di.fbs))
        if bf_id is not None:
            fn = L.hho_builtin_fn(bf_id)
            for f in np.nonzero(self.is_dir)[0]:
                p0 = np.ascontiguousarray(points[int(self.faces[f, 0])])
                p1 = np.ascontiguousarray(points[int(self.faces[f, 1])])
                assert L.hho_dirichlet_face_data(_dp(p0), _dp(p1), di.face_deg, fn, None, _dp(self.g[f])) == 0

    def assemble_cell(self, c, lhs_rowcol, rhs):
        """-> (rows, cols, vals) in push order, rhs_rows[msize], rhs_vals[msize]"""
        L = lib()
        di = self.di
        ms, cbs, fbs = di.msize, di.cbs, di.fbs
        lhs = np.ascontiguousarray(lhs_rowcol.T)
        rhs = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64))
        fids = np.ascontiguousarray(self.cell_faces[c])
        fdir = np.ascontiguousarray(self.is_dir[fids.astype(np.int64)])
        dd = np.zeros(ms)
        for lf in range(4):
            dd[cbs + lf * fbs: cbs + (lf + 1) * fbs] = self.g[int(fids[lf])]
        tr = np.zeros(ms * ms, dtype=np.int32)
        tc = np.zeros(ms * ms, dtype=np.int32)
        tv = np.zeros(ms * ms)
        nt = C.c_size_t(0)
        rr = np.zeros(ms, dtype=np.int64)
        rv = np.zeros(ms)
        st = L.hho_assembler_assemble_cell(di, c, self.nc, _u64p(fids), fdir.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           self.compress.ctypes.data_as(C.POINTER(C.c_int64)), _dp(lhs), _dp(rhs), _dp(dd),
                                           tr.ctypes.data_as(C.POINTER(C.c_int32)), tc.ctypes.data_as(C.POINTER(C.c_int32)),
                                           _dp(tv), C.byref(nt), rr.ctypes.data_as(C.POINTER(C.c_int64)), _dp(rv))
        assert st == 0
        n = nt.value
        return tr[:n], tc[:n], tv[:n], rr, rv


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class ObstacleAssembler(Assembler):
    """obstacle_assembler<Mesh> (hho.hpp:471-751) through the oracle's C restatement."""

    def __init__(self, mp, points, ptids, di, in_A, bf_id=None):
        super().__init__(mp, points, ptids, di, bf_id)
        L = lib()
        self.in_A = np.ascontiguousarray(np.asarray(in_A).astype(np.uint8))
        assert self.in_A.shape == (self.nc,)
        self.A_ct = np.zeros(self.nc, dtype=np.int64)
        self.B_ct = np.zeros(self.nc, dtype=np.int64)
        ni, na = C.c_size_t(0), C.c_size_t(0)
        L.hho_obstacle_tables(_u8p(self.in_A), self.nc, _i64p(self.A_ct), _i64p(self.B_ct), C.byref(ni), C.byref(na))
        self.num_I, self.num_A = ni.value, na.value
        self.system_size = di.cbs * (self.num_I + self.num_A) + di.fbs * self.num_other      # hho.hpp:586

    def assemble_cell(self, c, lhs_rowcol, rhs, gamma):
        L = lib()
        di = self.di
        ms, cbs, fbs = di.msize, di.cbs, di.fbs
        lhs = np.ascontiguousarray(lhs_rowcol.T)
        rhs = np.ascontiguousarray(np.asarray(rhs, dtype=np.float64))
        fids = np.ascontiguousarray(self.cell_faces[c])
        fdir = np.ascontiguousarray(self.is_dir[fids.astype(np.int64)])
        dd = np.zeros(ms)
        for lf in range(4):
            dd[cbs + lf * fbs: cbs + (lf + 1) * fbs] = self.g[int(fids[lf])]
        tr = np.zeros(ms * ms + 1, dtype=np.int32)
        tc = np.zeros(ms * ms + 1, dtype=np.int32)
        tv = np.zeros(ms * ms + 1)
        nt = C.c_size_t(0)
        rr = np.zeros(ms, dtype=np.int64)
        rv = np.zeros(ms)
        st = L.hho_obstacle_assemble_cell(di, c, self.nc, self.num_I, self.num_other, _u64p(fids), _u8p(fdir),
                                          _i64p(self.compress), _u8p(self.in_A), _i64p(self.A_ct), _i64p(self.B_ct),
                                          _dp(lhs), _dp(rhs), _dp(gamma), _dp(dd),
                                          tr.ctypes.data_as(C.POINTER(C.c_int32)), tc.ctypes.data_as(C.POINTER(C.c_int32)),
                                          _dp(tv), C.byref(nt), _i64p(rr), _dp(rv))
        assert st == 0
        n = nt.value
        return tr[:n], tc[:n], tv[:n], rr, rv

    def expand_solution(self, solution, gamma):
        """-> (alpha, beta)  hho.hpp:698-744"""
        di = self.di
        alpha = np.zeros(self.nc * di.cbs + self.nf * di.fbs)
        beta = np.zeros(self.nc * di.cbs)
        sol = np.ascontiguousarray(solution, dtype=np.float64)
        lib().hho_obstacle_expand_solution(di, self.nc, self.nf, _u8p(self.is_dir), _i64p(self.compress), _u8p(self.in_A),
                                           _dp(sol), _dp(self.g), _dp(gamma), _dp(alpha), _dp(beta))
        return alpha, beta

    def take_local_data(self, c, expanded):
        out = np.zeros(self.di.msize)
        fids = np.ascontiguousarray(self.cell_faces[c])
        lib().hho_obstacle_take_local_data(self.di, c, self.nc, _u64p(fids), _dp(expanded), _dp(out))
        return out


class CutMesh:
    """cuthho_poly_mesh + the preprocessing of cuthho_square.cpp:2036-2052 (oracle side)."""

    def __init__(self, N, radius=0.35, center=(0.5, 0.5), refsteps=4, agglomeration=False, line_y=None):
        L = lib()
        self.L = L
        self.N = N
        # circle_level_set (cuthho_square.cpp:56-89) or, line_y given, line_level_set y - line_y (:91-124)
        self.ls = CutLevelSet(0, radius, center[0], center[1], 0.0) if line_y is None else CutLevelSet(1, 0.0, 0.0, 0.0, line_y)
        self.h = L.cut_mesh_create(N, N, 0.0, 1.0, 0.0, 1.0)
        st = (L.cut_mesh_preprocess_agglomeration if agglomeration else L.cut_mesh_preprocess)(self.h, C.byref(self.ls), refsteps)
        if st != 0:
            raise RuntimeError("cut_mesh_preprocess failed: %d" % st)
        self.np_, self.nc, self.nf = (L.cut_mesh_num_points(self.h), L.cut_mesh_num_cells(self.h), L.cut_mesh_num_faces(self.h))
        as_np = np.ctypeslib.as_array
        self.points = as_np(L.cut_mesh_points(self.h), shape=(self.np_, 2)).copy()
        self.ptids = as_np(L.cut_mesh_cell_ptids(self.h), shape=(self.nc, 4)).copy()
        self.faces = as_np(L.cut_mesh_faces(self.h), shape=(self.nf, 2)).copy()
        self.bnd = as_np(L.cut_mesh_face_boundary(self.h), shape=(self.nf,)).copy()
        self.node_loc = as_np(L.cut_mesh_node_location(self.h), shape=(self.np_,)).copy()
        self.face_loc = as_np(L.cut_mesh_face_location(self.h), shape=(self.nf,)).copy()
        self.face_ip = as_np(L.cut_mesh_face_intersection(self.h), shape=(self.nf, 2)).copy()
        self.cell_loc = as_np(L.cut_mesh_cell_location(self.h), shape=(self.nc,)).copy()
        self.niface = L.cut_mesh_interface_points(self.h)
        self.cell_faces = np.array([[L.cut_mesh_cell_face(self.h, c, lf) for lf in range(4)] for c in range(self.nc)],
                                   dtype=np.uint64)

    def __del__(self):
        try:
            self.L.cut_mesh_free(self.h)
        except Exception:
            pass

    def interface(self, c):
        p = self.L.cut_mesh_cell_interface(self.h, c)
        return np.ctypeslib.as_array(p, shape=(self.niface, 2)).copy() if p else None

    def cell_quadrature(self, c, degree, where=CUT_NEG):
        qx, qy, qw = np.zeros(1024), np.zeros(1024), np.zeros(1024)
        n = self.L.cut_cell_quadrature(self.h, c, degree, where, _dp(qx), _dp(qy), _dp(qw), 1024)
        assert n >= 0, n
        return qx[:n], qy[:n], qw[:n]

    def face_quadrature(self, c, lf, degree, where=CUT_NEG):
        qx, qy, qw = np.zeros(8), np.zeros(8), np.zeros(8)
        n = self.L.cut_face_quadrature(self.h, c, lf, degree, where, _dp(qx), _dp(qy), _dp(qw), 8)
        assert n >= 0, n
        return qx[:n], qy[:n], qw[:n]

    def interface_quadrature(self, c, degree, where=CUT_NEG):
        qx, qy, qw = np.zeros(1024), np.zeros(1024), np.zeros(1024)
        n = self.L.cut_interface_quadrature(self.h, c, degree, where, _dp(qx), _dp(qy), _dp(qw), 1024)
        assert n >= 0, n
        return qx[:n], qy[:n], qw[:n]

    def laplacian(self, c, di, where=CUT_NEG):
        ms, rbs = di.msize, di.rbs
        oper = np.zeros((ms, rbs))
        data = np.zeros((ms, ms))
        rows = C.c_int(0)
        st = self.L.cut_make_hho_laplacian(self.h, C.byref(self.ls), c, di, where, _dp(oper), _dp(data), C.byref(rows))
        r = rows.value
        op = np.ascontiguousarray(oper.reshape(-1)[: ms * r].reshape(ms, r).T)
        return st, op, data.T.copy()

    def cut_stabilization(self, c, di, where=CUT_NEG):
        ms = di.msize
        stab = np.zeros((ms, ms))
        st = self.L.cut_make_hho_cut_stabilization(self.h, c, di, where, _dp(stab))
        return st, stab.T.copy()

    def rhs(self, c, degree, where=CUT_NEG, f_id=1, bcs_id=2):
        cbs = (degree + 2) * (degree + 1) // 2
        out = np.zeros(cbs)
        st = self.L.cut_make_rhs(self.h, C.byref(self.ls), c, degree, where, self.L.hho_builtin_fn(f_id),
                                 self.L.hho_builtin_fn(bcs_id), None, _dp(out))
        return st, out

    # ---- the same operators in binary128 from the same double quadrature lists (oracle/cut_truth.c): what sliver cells
    # are judged against.  Same return shapes as laplacian / cut_stabilization / rhs / laplacian_interface / rhs_side.
    def truth_laplacian(self, c, di, where=CUT_NEG):
        ms, rbs = di.msize, di.rbs
        oper = np.zeros((ms, rbs))
        data = np.zeros((ms, ms))
        st = self.L.cut_truth_laplacian(self.h, C.byref(self.ls), c, di, where, _dp(oper), _dp(data))
        return st, oper.T.copy(), data.T.copy()

    def truth_stabilization(self, c, di, where=CUT_NEG):
        ms = di.msize
        stab = np.zeros((ms, ms))
        st = self.L.cut_truth_stabilization(self.h, c, di, where, _dp(stab))
        return st, stab.T.copy()

    def truth_rhs(self, c, degree, where=CUT_NEG, f_id=1, bcs_id=2):
        out = np.zeros((degree + 2) * (degree + 1) // 2)
        st = self.L.cut_truth_rhs(self.h, C.byref(self.ls), c, degree, where, f_id, bcs_id, _dp(out))
        return st, out

    def truth_laplacian_interface(self, c, di, kappa_1=1.0, kappa_2=1.0, eta=5.0):
        ms2, rb2 = 2 * di.msize, 2 * di.rbs
        oper = np.zeros((ms2, rb2))
        data = np.zeros((ms2, ms2))
        prm = CutParams(kappa_1, kappa_2, eta)
        st = self.L.cut_truth_laplacian_interface(self.h, C.byref(self.ls), c, di, C.byref(prm), _dp(oper), _dp(data))
        self.last_interface_cond = float(self.L.cut_truth_last_interface_cond())
        return st, oper.T.copy(), data.T.copy()

    def truth_rhs_side(self, c, degree, where, f_id=1):
        out = np.zeros((degree + 2) * (degree + 1) // 2)
        st = self.L.cut_truth_rhs_side(self.h, c, degree, where, f_id, _dp(out))
        return st, out

    def truth_cond(self, c, di, where=CUT_NEG):
        return float(self.L.cut_truth_laplacian_cond(self.h, C.byref(self.ls), c, di, where))

    def agglo_set(self):
        a = np.zeros(self.nc, dtype=np.int8)
        self.L.cut_mesh_agglo_set(self.h, a.ctypes.data_as(C.POINTER(C.c_int8)))
        return a

    def neighbors(self):
        nb = np.zeros((self.nc, 8), dtype=np.int32)
        self.L.cut_mesh_neighbors(self.h, nb.ctypes.data_as(C.POINTER(C.c_int32)))
        return nb

    # ---- two-sided interface problem (cuthho_square -i) ----
    def laplacian_interface(self, c, di, kappa_1=1.0, kappa_2=1.0, eta=5.0):
        """make_hho_laplacian_interface -> (status, oper 2rbs x 2msize, data 2msize x 2msize), row-major numpy."""
        ms2, rb2 = 2 * di.msize, 2 * di.rbs
        oper = np.zeros((ms2, rb2))
        data = np.zeros((ms2, ms2))
        prm = CutParams(kappa_1, kappa_2, eta)
        st = self.L.cut_make_hho_laplacian_interface(self.h, C.byref(self.ls), c, di, C.byref(prm), _dp(oper), _dp(data))
        return st, oper.T.copy(), data.T.copy()

    def rhs_side(self, c, degree, where, f_id=1):
        out = np.zeros((degree + 2) * (degree + 1) // 2)
        st = self.L.cut_make_rhs_side(self.h, c, degree, where, self.L.hho_builtin_fn(f_id), None, _dp(out))
        return st, out

    def interface_tables(self):
        ct = np.zeros(self.nc, dtype=np.int64)
        ft = np.zeros(self.nf, dtype=np.int64)
        a, b = C.c_size_t(0), C.c_size_t(0)
        self.L.cut_interface_tables(self.h, _i64p(ct), _i64p(ft), C.byref(a), C.byref(b))
        return ct, ft, a.value, b.value

    def interface_assemble(self, di, c, cell_table, face_table, num_all_cells, lhs_rowcol, rhs, dirichlet_data):
        n = lhs_rowcol.shape[0]
        lhs = np.ascontiguousarray(lhs_rowcol.T)
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        dd = np.ascontiguousarray(dirichlet_data, dtype=np.float64)
        tr = np.zeros(n * n, dtype=np.int32); tc = np.zeros(n * n, dtype=np.int32); tv = np.zeros(n * n)
        nt = C.c_size_t(0)
        rr = np.zeros(n, dtype=np.int64); rv = np.zeros(n)
        st = self.L.cut_interface_assemble(self.h, di, c, _i64p(cell_table), _i64p(face_table), num_all_cells, _dp(lhs), _dp(rhs),
                                           _dp(dd), tr.ctypes.data_as(C.POINTER(C.c_int32)), tc.ctypes.data_as(C.POINTER(C.c_int32)),
                                           _dp(tv), C.byref(nt), _i64p(rr), _dp(rv))
        assert st == 0, st
        k = nt.value
        return tr[:k], tc[:k], tv[:k], rr, rv


def max_threads():
    """OpenMP threads the oracle's multi-threaded loops may use on this host"""
    return int(lib().hho_max_threads())


def matrix_assembly_timed(N, di, quad, stab, rows, rhs_fn=1, bcs_fn=2, rhs_di=0, lo=(0.0, 0.0), hi=(1.0, 1.0), nthreads=1):
    """The reference's "Matrix assembly" span (cuthho_square.cpp:881-905) on a bounded sample of the N x N generator
    mesh: the strip of its cell rows [rows[0], rows[1]) as a mesh of its own (N x (rows[1] - rows[0]) cells of the same
    size and shape, assembled completely -- so that the assembler's tables, the right-hand side and finalize cost what
    they cost per cell on the full mesh) -> dict(seconds_ops, seconds_assembly, cells, nnz, checksum)"""
    nrows = rows[1] - rows[0]
    hy = (hi[1] - lo[1]) / N
    mp = MeshParams(N, nrows, lo[0], hi[0], lo[1] + rows[0] * hy, lo[1] + rows[1] * hy)
    sec = (C.c_double * 2)()
    nnz, cs = C.c_size_t(0), C.c_double(0.0)
    st = lib().hho_matrix_assembly_timed(C.byref(mp), 0, nrows, di, quad, stab, rhs_fn, bcs_fn, rhs_di, nthreads, sec,
                                         C.byref(nnz), C.byref(cs))
    if st not in (0,):
        raise RuntimeError("hho_matrix_assembly_timed: status %d" % st)
    return {"seconds_ops": sec[0], "seconds_assembly": sec[1], "cells": nrows * N, "nnz": nnz.value,
            "checksum": cs.value}


def set_from_triplets(rows, cols, vals, nrows, nthreads=1):
    rows = np.ascontiguousarray(rows, dtype=np.int32).ravel()
    cols = np.ascontiguousarray(cols, dtype=np.int32).ravel()
    vals = np.ascontiguousarray(vals, dtype=np.float64).ravel()
    rowptr = np.zeros(nrows + 1, dtype=np.int64)
    colind = np.zeros(max(rows.size, 1), dtype=np.int32)
    values = np.zeros(max(rows.size, 1))
    i32p = C.POINTER(C.c_int32)
    nnz = lib().hho_set_from_triplets(rows.size, rows.ctypes.data_as(i32p), cols.ctypes.data_as(i32p), _dp(vals), nrows,
                                      rowptr.ctypes.data_as(C.POINTER(C.c_int64)), colind.ctypes.data_as(i32p), _dp(values), nthreads)
    return rowptr, colind[:nnz], values[:nnz]
