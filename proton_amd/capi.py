"""ctypes binding of libproton_amd.so (the C ABI in include/proton_amd.h).

Plumbing only: the product is the shared library.  There is NO CPU fallback -- if the
library is missing or the GPU is absent the calls fail loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PA_LIB") or os.path.join(HERE, "lib", "libproton_amd.so")   # PA_LIB: A/B tuning builds

QUAD_TENSOR, QUAD_FAN = 0, 1
STAB_NONE, STAB_NAIVE, STAB_FANCY = 0, 1, 2
FN_SAMPLED, FN_SIN_SIN_RHS, FN_SIN_SIN_SOL, FN_OBSTACLE_RHS, FN_OBSTACLE_SOL, FN_ONE = range(6)

STATUS = {0: "PA_OK", 1: "PA_ERR_INVALID_ARG", 2: "PA_ERR_INVALID_DEGREE", 3: "PA_ERR_QUADRATURE",
          4: "PA_ERR_HIP", 5: "PA_ERR_NO_MESH", 6: "PA_ERR_NOT_SPD", 7: "PA_ERR_COMM"}

# every symbol include/proton_amd.h declares
EXPORTS = [
    "pa_abi_version", "pa_degree_info_equal", "pa_degree_info_make", "pa_sizes_for",
    "pa_context_create", "pa_context_destroy", "pa_context_synchronize", "pa_context_set_cut_overlap", "pa_last_error",
    "pa_context_trim", "pa_context_set_record_cap",
    "pa_malloc", "pa_free", "pa_memcpy_h2d", "pa_memcpy_d2h", "pa_memset",
    "pa_mesh_upload", "pa_mesh_attach_device", "pa_mesh_generate", "pa_mesh_counts",
    "pa_local_ops_batch", "pa_cell_rhs_batch", "pa_cell_quadrature_points",
    "pa_static_condensation_batch", "pa_static_condensation_packed_batch", "pa_local_ops_launch_info",
    "pa_mesh_set_faces", "pa_assembler_query", "pa_dirichlet_data_batch", "pa_face_quadrature_points",
    "pa_triplets_batch", "pa_csr_from_triplets", "pa_conjugated_gradient", "pa_take_local_data_batch", "pa_project_function_batch", "pa_energy_form_batch",
    "pa_obstacle_tables", "pa_obstacle_triplets_batch", "pa_obstacle_expand_solution",
    "pa_obstacle_take_local_data_batch",
    "pa_cut_preprocess", "pa_cut_query", "pa_cut_local_ops_batch", "pa_cut_merge",
    "pa_cut_preprocess_agglomeration", "pa_cut_agglo_query", "pa_cut_query_tags", "pa_cut_quadrature_points", "pa_cut_rhs_sampled_batch",
    "pa_cut_interface_ops_batch", "pa_cut_interface_uncut_batch", "pa_interface_assembler_query",
    "pa_interface_triplets_batch", "pa_interface_cell_offsets",
    "pa_condensed_ops_batch", "pa_condensed_recover_batch", "pa_condensed_query", "pa_condensed_triplets_batch",
    "pa_assembler_csr_query", "pa_assembler_csr_pattern", "pa_assembler_csr_fill",
    "pa_condensed_csr_pattern", "pa_condensed_csr_fill", "pa_condensed_halo_pack", "pa_condensed_take_faces",
    "pa_condensed_expand_solution", "pa_condensed_launch_info", "pa_condensed_partition_info",
    "pa_comm_unique_id", "pa_comm_create", "pa_comm_destroy", "pa_comm_info", "pa_comm_last_error",
    "pa_comm_halo_exchange_start", "pa_comm_allgather_start", "pa_comm_allreduce_sum_start", "pa_comm_wait",
    "pa_comm_neighbour_exchange_start", "pa_conjugated_gradient_rows", "pa_comm_cg_transport", "pa_copy_to_host", "pa_copy_to_device", "pa_cut_uncut_rhs_batch",
    "pa_mesh_set_points", "pa_cut_preprocess_rows", "pa_cut_merge_condensed",
]


# pa_cg_transport: the three callbacks of pa_conjugated_gradient_rows
CG_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
CG_HALO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                      C.c_void_p)
CG_COUNTS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class CgTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allreduce_sum", CG_ALLREDUCE), ("halo", CG_HALO), ("neighbour_counts", CG_COUNTS)]


class DegreeInfo(C.Structure):
    _fields_ = [("cell_deg", C.c_int32), ("face_deg", C.c_int32), ("rec_deg", C.c_int32)]


class Sizes(C.Structure):
    _fields_ = [("rbs", C.c_int32), ("cbs", C.c_int32), ("fbs", C.c_int32), ("msize", C.c_int32),
                ("oper_rows", C.c_int32), ("cell_qps", C.c_int32), ("face_qps", C.c_int32)]


class InterfaceParams(C.Structure):
    _fields_ = [("kappa_1", C.c_double), ("kappa_2", C.c_double), ("eta", C.c_double)]


class InterfaceInfo(C.Structure):
    _fields_ = [("num_all_cells", C.c_uint64), ("num_other_faces", C.c_uint64), ("system_size", C.c_uint64), ("ncut", C.c_uint64)]


class LaunchInfo(C.Structure):
    _fields_ = [("lanes_per_cell", C.c_int32), ("cells_per_block", C.c_int32), ("block_threads", C.c_int32),
                ("lds_bytes_per_block", C.c_int32), ("grid_blocks", C.c_int32), ("kernel_name", C.c_char_p)]


class AssemblerInfo(C.Structure):
    _fields_ = [("system_size", C.c_uint64), ("ncells_global", C.c_uint64), ("cell_base", C.c_uint64),
                ("nfaces_local", C.c_uint64), ("face_base", C.c_uint64), ("num_other_faces", C.c_uint64)]


class AssemblerCsrInfo(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("nnz", C.c_uint64)]


class CondensedInfo(C.Structure):
    _fields_ = [("system_size", C.c_uint64), ("num_other_faces", C.c_uint64), ("nf", C.c_int32), ("cond_doubles", C.c_int32),
                ("row_begin", C.c_uint64), ("row_end", C.c_uint64), ("nnz_owned", C.c_uint64), ("halo_cells", C.c_uint64),
                ("halo_doubles", C.c_int32), ("has_below", C.c_int32)]


class LevelSet(C.Structure):
    _fields_ = [("kind", C.c_int32), ("radius", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("cut_y", C.c_double)]


LOC_NEGATIVE, LOC_POSITIVE, LOC_ON_INTERFACE = 0, 1, 2


class ProtonAmdError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: %s%s" % (where, STATUS.get(status, status), (" (" + detail + ")") if detail else ""))


_lib = None


def lib():
    """Load the shared library (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("PA_NO_TORCH_PRELOAD") != "1":
        # torch bundles its own HIP/HSA runtime; two runtimes in one process cannot both open the
        # device.  Import torch first so libproton_amd.so binds to the runtime torch already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("proton_amd: %s is missing -- run `python -m proton_amd._build` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, dp = C.c_void_p, C.c_size_t, C.c_void_p
    L.pa_abi_version.restype = C.c_int
    L.pa_degree_info_equal.restype = DegreeInfo
    L.pa_degree_info_equal.argtypes = [C.c_int]
    L.pa_degree_info_make.restype = DegreeInfo
    L.pa_degree_info_make.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.pa_sizes_for.argtypes = [DegreeInfo, C.c_int, C.POINTER(Sizes)]
    L.pa_context_create.argtypes = [C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.pa_context_destroy.argtypes = [vp]
    L.pa_context_synchronize.argtypes = [vp]
    L.pa_context_set_cut_overlap.argtypes = [vp, C.c_int]
    L.pa_context_trim.argtypes = [vp]
    L.pa_context_set_record_cap.argtypes = [vp, sz]
    L.pa_last_error.argtypes = [vp]
    L.pa_last_error.restype = C.c_char_p
    L.pa_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.pa_free.argtypes = [vp, vp]
    L.pa_memcpy_h2d.argtypes = [vp, vp, vp, sz]
    L.pa_memcpy_d2h.argtypes = [vp, vp, vp, sz]
    L.pa_memset.argtypes = [vp, vp, C.c_int, sz]
    L.pa_mesh_upload.argtypes = [vp, vp, sz, vp, sz]
    L.pa_mesh_attach_device.argtypes = [vp, dp, sz, dp, sz]
    L.pa_mesh_generate.argtypes = [vp, sz, sz, C.c_double, C.c_double, C.c_double, C.c_double, sz, sz]
    L.pa_mesh_counts.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    L.pa_local_ops_batch.argtypes = [vp, DegreeInfo, C.c_int, C.c_int, sz, sz, dp, dp, dp, dp, dp]
    L.pa_cell_rhs_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp, sz, sz, dp]
    L.pa_cell_quadrature_points.argtypes = [vp, C.c_int, C.c_int, sz, sz, dp, C.POINTER(C.c_int32)]
    L.pa_static_condensation_batch.argtypes = [vp, DegreeInfo, sz, dp, dp, dp, dp, dp, dp]
    L.pa_static_condensation_packed_batch.argtypes = [vp, DegreeInfo, sz, dp, dp, dp, dp, dp]
    L.pa_local_ops_launch_info.argtypes = [vp, DegreeInfo, C.c_int, C.c_int, sz, C.POINTER(LaunchInfo)]
    L.pa_mesh_set_faces.argtypes = [vp, vp, vp, vp, sz]
    L.pa_assembler_query.argtypes = [vp, DegreeInfo, C.POINTER(AssemblerInfo)]
    L.pa_dirichlet_data_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.pa_face_quadrature_points.argtypes = [vp, C.c_int, dp]
    L.pa_triplets_batch.argtypes = [vp, DegreeInfo, sz, sz, dp, dp, dp, dp, dp, dp, dp, dp]
    L.pa_csr_from_triplets.argtypes = [vp, sz, dp, dp, dp, sz, dp, dp, dp, C.POINTER(sz)]
    L.pa_conjugated_gradient.argtypes = [vp, sz, dp, dp, dp, dp, dp, C.c_double, C.c_double, sz, C.c_int,
                                         C.POINTER(C.c_int32), C.POINTER(sz), C.POINTER(C.c_double)]
    L.pa_conjugated_gradient_rows.argtypes = [vp, C.POINTER(CgTransport), C.c_int64, C.c_int64, dp, dp, dp, dp, dp, C.c_double, C.c_double, sz,
                                              C.c_int, C.POINTER(C.c_int32), C.POINTER(sz), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.pa_comm_cg_transport.argtypes = [vp, C.POINTER(CgTransport)]
    L.pa_copy_to_host.argtypes = [vp, vp, vp, sz]
    L.pa_copy_to_device.argtypes = [vp, vp, vp, sz]
    L.pa_comm_neighbour_exchange_start.argtypes = [vp, dp, sz, dp, sz, dp, sz, dp, sz]
    L.pa_take_local_data_batch.argtypes = [vp, DegreeInfo, sz, sz, dp, dp, dp]
    L.pa_project_function_batch.argtypes = [vp, DegreeInfo, C.c_int, C.c_int, C.c_int, dp, dp, sz, sz, dp, dp]
    L.pa_energy_form_batch.argtypes = [vp, DegreeInfo, sz, dp, dp, dp, dp]
    L.pa_obstacle_tables.argtypes = [vp, vp, vp, vp, C.POINTER(sz), C.POINTER(sz)]
    L.pa_obstacle_triplets_batch.argtypes = [vp, DegreeInfo, sz, sz, dp, dp, dp, dp, vp, vp, vp, sz, dp, dp, dp, dp, dp]
    L.pa_obstacle_expand_solution.argtypes = [vp, DegreeInfo, dp, dp, dp, vp, vp, vp, sz, dp, dp]
    L.pa_obstacle_take_local_data_batch.argtypes = [vp, DegreeInfo, sz, sz, dp, dp]
    L.pa_cut_preprocess.argtypes = [vp, sz, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(LevelSet), C.c_int]
    L.pa_cut_preprocess_rows.argtypes = [vp, sz, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(LevelSet), C.c_int, sz, sz]
    L.pa_cut_merge_condensed.argtypes = [vp, C.c_int, dp, dp, dp]
    L.pa_mesh_set_points.argtypes = [vp, dp, sz]
    L.pa_cut_query.argtypes = [vp, C.POINTER(sz), vp, vp]
    L.pa_cut_local_ops_batch.argtypes = [vp, C.c_int, C.POINTER(LevelSet), C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp]
    L.pa_cut_merge.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp]
    L.pa_cut_uncut_rhs_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    L.pa_cut_query_tags.argtypes = [vp, vp, vp, vp]
    L.pa_cut_preprocess_agglomeration.argtypes = [vp, sz, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(LevelSet), C.c_int]
    L.pa_cut_agglo_query.argtypes = [vp, vp, vp]
    L.pa_cut_quadrature_points.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(sz)]
    L.pa_cut_rhs_sampled_batch.argtypes = [vp, C.c_int, C.POINTER(LevelSet), C.c_int, dp, dp, dp]
    L.pa_cut_interface_ops_batch.argtypes = [vp, C.c_int, C.POINTER(LevelSet), C.POINTER(InterfaceParams), C.c_int, dp, dp, dp, dp, dp]
    L.pa_cut_interface_uncut_batch.argtypes = [vp, C.c_int, C.POINTER(InterfaceParams), C.c_int, dp, dp, dp]
    L.pa_interface_assembler_query.argtypes = [vp, C.c_int, C.POINTER(InterfaceInfo)]
    L.pa_interface_triplets_batch.argtypes = [vp, C.c_int] + [dp] * 15
    L.pa_interface_cell_offsets.argtypes = [vp, C.c_int, dp]
    L.pa_condensed_ops_batch.argtypes = [vp, DegreeInfo, C.c_int, C.c_int, sz, sz, dp, dp, dp]
    L.pa_condensed_recover_batch.argtypes = [vp, DegreeInfo, C.c_int, C.c_int, sz, sz, dp, dp, dp, dp]
    L.pa_condensed_query.argtypes = [vp, DegreeInfo, C.POINTER(CondensedInfo)]
    L.pa_condensed_triplets_batch.argtypes = [vp, DegreeInfo, sz, sz, dp, dp, dp, dp, dp, dp, dp]
    L.pa_assembler_csr_query.argtypes = [vp, DegreeInfo, C.POINTER(AssemblerCsrInfo)]
    L.pa_assembler_csr_pattern.argtypes = [vp, DegreeInfo, dp, dp]
    L.pa_assembler_csr_fill.argtypes = [vp, DegreeInfo, dp, dp, dp, dp, dp]
    L.pa_condensed_csr_pattern.argtypes = [vp, DegreeInfo, dp, dp]
    L.pa_condensed_csr_fill.argtypes = [vp, DegreeInfo, dp, dp, dp, dp, dp]
    L.pa_condensed_halo_pack.argtypes = [vp, DegreeInfo, dp, dp, dp]
    L.pa_condensed_take_faces.argtypes = [vp, DegreeInfo, sz, sz, dp, dp, dp]
    L.pa_condensed_expand_solution.argtypes = [vp, DegreeInfo, dp, dp, dp]
    L.pa_condensed_launch_info.argtypes = [vp, DegreeInfo, C.c_int, C.c_int, sz, C.POINTER(LaunchInfo)]
    L.pa_condensed_partition_info.argtypes = [sz, sz, sz, sz, DegreeInfo, C.POINTER(CondensedInfo)]
    L.pa_comm_unique_id.argtypes = [vp, sz]
    L.pa_comm_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.pa_comm_destroy.argtypes = [vp]
    L.pa_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pa_comm_last_error.argtypes = [vp]
    L.pa_comm_last_error.restype = C.c_char_p
    L.pa_comm_halo_exchange_start.argtypes = [vp, dp, sz, dp, sz]
    L.pa_comm_allgather_start.argtypes = [vp, dp, dp, sz]
    L.pa_comm_allreduce_sum_start.argtypes = [vp, dp, sz]
    L.pa_comm_wait.argtypes = [vp]
    _lib = L
    return L


def degree_info(cd, fd):
    fb = C.c_int(0)
    d = lib().pa_degree_info_make(cd, fd, C.byref(fb))
    return d, bool(fb.value)


def sizes_for(di, quad):
    s = Sizes()
    st = lib().pa_sizes_for(di, quad, C.byref(s))
    if st != 0:
        raise ProtonAmdError(st, "pa_sizes_for")
    return s


def condensed_partition_info(Nx, Ny, rows, di):
    """closed-form row partition of the face-only system for the slab `rows` (no context, no device)"""
    out = CondensedInfo()
    st = lib().pa_condensed_partition_info(Nx, Ny, rows[0], rows[1], di, C.byref(out))
    if st != 0:
        raise ProtonAmdError(st, "pa_condensed_partition_info")
    return out


COMM_ID_BYTES = 128


def comm_unique_id():
    """bytes of a fresh RCCL unique id (rank 0; hand it to the other ranks)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    st = lib().pa_comm_unique_id(buf, COMM_ID_BYTES)
    if st != 0:
        raise ProtonAmdError(st, "pa_comm_unique_id", "is librccl loadable?")
    return buf.raw


class Comm:
    """pa_comm: the RCCL communicator of one rank's context (one process per GPU)."""

    def __init__(self, ctx, nranks, rank, unique_id):
        self._L = lib()
        h = C.c_void_p()
        idb = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        st = self._L.pa_comm_create(ctx.h, nranks, rank, idb, C.byref(h))
        if st != 0:
            raise ProtonAmdError(st, "pa_comm_create")
        self.h, self.nranks, self.rank = h, nranks, rank

    def _ck(self, st, where):
        if st != 0:
            raise ProtonAmdError(st, where, self._L.pa_comm_last_error(self.h).decode())

    def halo_exchange_start(self, send_up, send_count, recv_below, recv_count):
        self._ck(self._L.pa_comm_halo_exchange_start(self.h, send_up, send_count, recv_below, recv_count), "pa_comm_halo_exchange_start")

    def allgather_start(self, send, recv, bytes_per_rank):
        self._ck(self._L.pa_comm_allgather_start(self.h, send, recv, bytes_per_rank), "pa_comm_allgather_start")

    def allreduce_sum_start(self, buf, count):
        self._ck(self._L.pa_comm_allreduce_sum_start(self.h, buf, count), "pa_comm_allreduce_sum_start")

    def wait(self):
        self._ck(self._L.pa_comm_wait(self.h), "pa_comm_wait")

    def neighbour_exchange_start(self, send_lo, n_send_lo, send_hi, n_send_hi, recv_lo, n_recv_lo, recv_hi, n_recv_hi):
        self._ck(self._L.pa_comm_neighbour_exchange_start(self.h, send_lo, n_send_lo, send_hi, n_send_hi, recv_lo, n_recv_lo, recv_hi, n_recv_hi),
                 "pa_comm_neighbour_exchange_start")

    def cg_transport(self):
        """the RCCL transport of pa_conjugated_gradient_rows over this communicator"""
        t = CgTransport()
        self._ck(self._L.pa_comm_cg_transport(self.h, C.byref(t)), "pa_comm_cg_transport")
        return t

    def close(self):
        if self.h:
            self._L.pa_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """RAII wrapper of pa_context.  `stream` is a raw hipStream_t handle (int; 0/None = HIP's
    default stream); own_stream=True lets the library create its own non-blocking stream."""

    def __init__(self, device=0, stream=None, own_stream=False):
        self._L = lib()
        h = C.c_void_p()
        st = self._L.pa_context_create(device, C.c_void_p(stream) if stream else None, int(own_stream), C.byref(h))
        if st != 0:
            raise ProtonAmdError(st, "pa_context_create", "is a GPU visible?")
        self.h = h
        self.device = device

    def _ck(self, st, where):
        if st != 0:
            raise ProtonAmdError(st, where, self._L.pa_last_error(self.h).decode())

    def close(self):
        if self.h:
            self._L.pa_context_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._ck(self._L.pa_context_synchronize(self.h), "pa_context_synchronize")

    def trim(self):
        self._ck(self._L.pa_context_trim(self.h), "pa_context_trim")

    def set_record_cap(self, nbytes):
        self._ck(self._L.pa_context_set_record_cap(self.h, nbytes), "pa_context_set_record_cap")

    def set_cut_overlap(self, on):
        self._ck(self._L.pa_context_set_cut_overlap(self.h, 1 if on else 0), "pa_context_set_cut_overlap")

    def mesh_upload(self, points, ptids):
        import numpy as np
        points = np.ascontiguousarray(points, dtype=np.float64)
        ptids = np.ascontiguousarray(ptids, dtype=np.uint32)
        self._ck(self._L.pa_mesh_upload(self.h, points.ctypes.data, points.shape[0], ptids.ctypes.data, ptids.shape[0]),
                 "pa_mesh_upload")

    def mesh_attach_device(self, d_points, npoints, d_ptids, ncells):
        self._ck(self._L.pa_mesh_attach_device(self.h, d_points, npoints, d_ptids, ncells), "pa_mesh_attach_device")

    def mesh_generate(self, Nx, Ny, lo=(0.0, 0.0), hi=(1.0, 1.0), rows=None):
        r0, r1 = rows if rows is not None else (0, Ny)
        self._ck(self._L.pa_mesh_generate(self.h, Nx, Ny, lo[0], hi[0], lo[1], hi[1], r0, r1), "pa_mesh_generate")

    def mesh_counts(self):
        a, b = C.c_size_t(), C.c_size_t()
        self._ck(self._L.pa_mesh_counts(self.h, C.byref(a), C.byref(b)), "pa_mesh_counts")
        return a.value, b.value

    def local_ops(self, di, quad, stab, first, n, oper=None, data=None, stab_out=None, lc=None, info=None):
        """Device pointers (ints) or None.  Asynchronous on the context's stream."""
        self._ck(self._L.pa_local_ops_batch(self.h, di, quad, stab, first, n, oper, data, stab_out, lc, info),
                 "pa_local_ops_batch")

    def cell_rhs(self, degree, dinc, quad, fn, first, n, rhs, fvals=None):
        self._ck(self._L.pa_cell_rhs_batch(self.h, degree, dinc, quad, fn, fvals, first, n, rhs), "pa_cell_rhs_batch")

    def cell_quadrature_points(self, degree, quad, first, n, xyw=None):
        nq = C.c_int32(0)
        self._ck(self._L.pa_cell_quadrature_points(self.h, degree, quad, first, n, xyw, C.byref(nq)),
                 "pa_cell_quadrature_points")
        return nq.value

    def static_condensation_packed(self, di, n, lc, rhs=None, Sp=None, g=None, info=None):
        self._ck(self._L.pa_static_condensation_packed_batch(self.h, di, n, lc, rhs, Sp, g, info),
                 "pa_static_condensation_packed_batch")

    def static_condensation(self, di, n, lc, rhs=None, S=None, g=None, rec=None, info=None):
        self._ck(self._L.pa_static_condensation_batch(self.h, di, n, lc, rhs, S, g, rec, info),
                 "pa_static_condensation_batch")

    def mesh_set_faces(self, cell_faces, face_pts, face_is_dirichlet):
        import numpy as np
        cf = np.ascontiguousarray(cell_faces, dtype=np.uint32)
        fp = np.ascontiguousarray(face_pts, dtype=np.uint32)
        fd = np.ascontiguousarray(face_is_dirichlet, dtype=np.uint8)
        self._ck(self._L.pa_mesh_set_faces(self.h, cf.ctypes.data, fp.ctypes.data, fd.ctypes.data, fd.shape[0]),
                 "pa_mesh_set_faces")

    def assembler_query(self, di):
        info = AssemblerInfo()
        self._ck(self._L.pa_assembler_query(self.h, di, C.byref(info)), "pa_assembler_query")
        return info

    def dirichlet_data(self, face_deg, fn, g, fvals=None):
        self._ck(self._L.pa_dirichlet_data_batch(self.h, face_deg, fn, fvals, g), "pa_dirichlet_data_batch")

    def face_quadrature_points(self, face_deg, xyw):
        self._ck(self._L.pa_face_quadrature_points(self.h, face_deg, xyw), "pa_face_quadrature_points")

    def triplets(self, di, first, n, lc, rhs, g, rows, cols, vals, rhs_rows, rhs_vals):
        self._ck(self._L.pa_triplets_batch(self.h, di, first, n, lc, rhs, g, rows, cols, vals, rhs_rows, rhs_vals),
                 "pa_triplets_batch")

    def csr_from_triplets(self, nslots, rows, cols, vals, nrows, rowptr, colind, values):
        nnz = C.c_size_t(0)
        self._ck(self._L.pa_csr_from_triplets(self.h, nslots, rows, cols, vals, nrows, rowptr, colind, values, C.byref(nnz)),
                 "pa_csr_from_triplets")
        return nnz.value

    def conjugated_gradient(self, nrows, rowptr, colind, values, b, x, tol=1e-9, div=100.0, max_iter=1000, precond=True):
        reason, iters, rr = C.c_int32(0), C.c_size_t(0), C.c_double(0.0)
        self._ck(self._L.pa_conjugated_gradient(self.h, nrows, rowptr, colind, values, b, x, tol, div, max_iter, int(precond),
                                                C.byref(reason), C.byref(iters), C.byref(rr)), "pa_conjugated_gradient")
        return reason.value, iters.value, rr.value

    def copy_to_host(self, host_dst, d_src, nbytes):
        self._ck(self._L.pa_copy_to_host(self.h, host_dst, d_src, nbytes), "pa_copy_to_host")

    def copy_to_device(self, d_dst, host_src, nbytes):
        self._ck(self._L.pa_copy_to_device(self.h, d_dst, host_src, nbytes), "pa_copy_to_device")

    def conjugated_gradient_rows(self, transport, row_begin, row_end, rowptr, colind, values, b, x, tol=1e-9, div=100.0, max_iter=1000,
                                 precond=True):
        """pa_conjugated_gradient_rows; transport: a CgTransport (or None = one rank).  Returns (reason, iterations, rr)."""
        reason, iters, rr, tstat = C.c_int32(0), C.c_size_t(0), C.c_double(0.0), C.c_int32(0)
        tp = C.byref(transport) if transport is not None else None
        self._ck(self._L.pa_conjugated_gradient_rows(self.h, tp, row_begin, row_end, rowptr, colind, values, b, x, tol, div, max_iter,
                                                     int(precond), C.byref(reason), C.byref(iters), C.byref(rr), C.byref(tstat)),
                 "pa_conjugated_gradient_rows (transport status %d)" % tstat.value)
        return reason.value, iters.value, rr.value

    def take_local_data(self, di, first, n, solution, g, out):
        self._ck(self._L.pa_take_local_data_batch(self.h, di, first, n, solution, g, out), "pa_take_local_data_batch")

    def project_function(self, di, quad, dinc, fn, cell_fvals, face_fvals, first, n, out, info):
        self._ck(self._L.pa_project_function_batch(self.h, di, quad, dinc, fn, cell_fvals, face_fvals, first, n, out, info),
                 "pa_project_function_batch")

    def energy_form(self, di, n, lc, u, v, out):
        self._ck(self._L.pa_energy_form_batch(self.h, di, n, lc, u, v, out), "pa_energy_form_batch")

    def obstacle_tables(self, in_A, A_ct, B_ct):
        ni, na = C.c_size_t(0), C.c_size_t(0)
        self._ck(self._L.pa_obstacle_tables(self.h, in_A, A_ct, B_ct, C.byref(ni), C.byref(na)), "pa_obstacle_tables")
        return ni.value, na.value

    def obstacle_triplets(self, di, first, n, lc, rhs, g, gamma, in_A, A_ct, B_ct, num_I, rows, cols, vals, rhs_rows, rhs_vals):
        self._ck(self._L.pa_obstacle_triplets_batch(self.h, di, first, n, lc, rhs, g, gamma, in_A, A_ct, B_ct, num_I,
                                                    rows, cols, vals, rhs_rows, rhs_vals), "pa_obstacle_triplets_batch")

    def obstacle_expand_solution(self, di, solution, g, gamma, in_A, A_ct, B_ct, num_I, alpha, beta):
        self._ck(self._L.pa_obstacle_expand_solution(self.h, di, solution, g, gamma, in_A, A_ct, B_ct, num_I, alpha, beta),
                 "pa_obstacle_expand_solution")

    def obstacle_take_local_data(self, di, first, n, expanded, out):
        self._ck(self._L.pa_obstacle_take_local_data_batch(self.h, di, first, n, expanded, out),
                 "pa_obstacle_take_local_data_batch")

    def cut_quadrature_points(self, face_deg, where, which):
        """-> (offsets[ncut+1] uint32, xyw[count, 3]) host arrays"""
        import numpy as np
        n = C.c_size_t(0)
        self._ck(self._L.pa_cut_quadrature_points(self.h, face_deg, where, which, None, None, C.byref(n)), "pa_cut_quadrature_points")
        ncut = self.cut_query()[0]
        off = np.zeros(ncut + 1, dtype=np.uint32)
        xyw = np.zeros((n.value, 3))
        self._ck(self._L.pa_cut_quadrature_points(self.h, face_deg, where, which, off.ctypes.data, xyw.ctypes.data, C.byref(n)),
                 "pa_cut_quadrature_points")
        return off, xyw

    def cut_rhs_sampled(self, face_deg, ls, where, rhs_vals, bcs_vals, rhs):
        self._ck(self._L.pa_cut_rhs_sampled_batch(self.h, face_deg, C.byref(ls), where, rhs_vals, bcs_vals, rhs), "pa_cut_rhs_sampled_batch")

    def cut_interface_ops(self, face_deg, ls, parms, rhs_fn, oper, data, lc, rhs, info):
        self._ck(self._L.pa_cut_interface_ops_batch(self.h, face_deg, C.byref(ls), C.byref(parms), rhs_fn, oper, data, lc, rhs, info),
                 "pa_cut_interface_ops_batch")

    def cut_interface_uncut(self, face_deg, parms, rhs_fn, lc, rhs, info):
        self._ck(self._L.pa_cut_interface_uncut_batch(self.h, face_deg, C.byref(parms), rhs_fn, lc, rhs, info),
                 "pa_cut_interface_uncut_batch")

    def interface_info(self, face_deg):
        out = InterfaceInfo()
        self._ck(self._L.pa_interface_assembler_query(self.h, face_deg, C.byref(out)), "pa_interface_assembler_query")
        return out

    def interface_triplets(self, face_deg, *ptrs):
        self._ck(self._L.pa_interface_triplets_batch(self.h, face_deg, *ptrs), "pa_interface_triplets_batch")

    def interface_cell_offsets(self, face_deg, out):
        self._ck(self._L.pa_interface_cell_offsets(self.h, face_deg, out), "pa_interface_cell_offsets")

    def cut_preprocess(self, Nx, Ny, ls, refsteps, lo=(0.0, 0.0), hi=(1.0, 1.0), rows=None):
        if rows is None:
            self._ck(self._L.pa_cut_preprocess(self.h, Nx, Ny, lo[0], hi[0], lo[1], hi[1], C.byref(ls), refsteps), "pa_cut_preprocess")
        else:
            self._ck(self._L.pa_cut_preprocess_rows(self.h, Nx, Ny, lo[0], hi[0], lo[1], hi[1], C.byref(ls), refsteps, rows[0], rows[1]),
                     "pa_cut_preprocess_rows")

    def mesh_set_points(self, d_points, npoints):
        self._ck(self._L.pa_mesh_set_points(self.h, d_points, npoints), "pa_mesh_set_points")

    def cut_merge_condensed(self, face_deg, cut_Sp, cut_g, cond):
        self._ck(self._L.pa_cut_merge_condensed(self.h, face_deg, cut_Sp, cut_g, cond), "pa_cut_merge_condensed")

    def cut_preprocess_agglomeration(self, Nx, Ny, ls, refsteps, lo=(0.0, 0.0), hi=(1.0, 1.0)):
        self._ck(self._L.pa_cut_preprocess_agglomeration(self.h, Nx, Ny, lo[0], hi[0], lo[1], hi[1], C.byref(ls), refsteps),
                 "pa_cut_preprocess_agglomeration")

    def cut_agglo_query(self):
        import numpy as np
        nc = self.mesh_counts()[1]
        agglo = np.zeros(nc, dtype=np.int8)
        nb = np.zeros((nc, 8), dtype=np.int32)
        self._ck(self._L.pa_cut_agglo_query(self.h, agglo.ctypes.data, nb.ctypes.data), "pa_cut_agglo_query")
        return agglo, nb

    def cut_query(self):
        import numpy as np
        n = C.c_size_t(0)
        nc = self.mesh_counts()[1]
        loc = np.zeros(nc, dtype=np.int8)
        idx = np.zeros(nc, dtype=np.int32)
        self._ck(self._L.pa_cut_query(self.h, C.byref(n), loc.ctypes.data, idx.ctypes.data), "pa_cut_query")
        return n.value, loc, idx

    def cut_local_ops(self, face_deg, ls, where, rhs_fn, bcs_fn, oper=None, data=None, stab=None, lc=None, rhs=None, info=None):
        self._ck(self._L.pa_cut_local_ops_batch(self.h, face_deg, C.byref(ls), where, rhs_fn, bcs_fn, oper, data, stab, lc, rhs, info),
                 "pa_cut_local_ops_batch")

    def cut_merge(self, face_deg, where, cut_lc, cut_rhs, lc, rhs):
        self._ck(self._L.pa_cut_merge(self.h, face_deg, where, cut_lc, cut_rhs, lc, rhs), "pa_cut_merge")

    def cut_uncut_rhs(self, degree, where, fn, rhs):
        self._ck(self._L.pa_cut_uncut_rhs_batch(self.h, degree, where, fn, rhs), "pa_cut_uncut_rhs_batch")

    # ---- condensed mode --------------------------------------------------------------
    def condensed_ops(self, di, quad, stab, first, n, rhs, cond, info=None):
        self._ck(self._L.pa_condensed_ops_batch(self.h, di, quad, stab, first, n, rhs, cond, info), "pa_condensed_ops_batch")

    def condensed_recover(self, di, quad, stab, first, n, rhs, uF, uT, info=None):
        self._ck(self._L.pa_condensed_recover_batch(self.h, di, quad, stab, first, n, rhs, uF, uT, info), "pa_condensed_recover_batch")

    def condensed_query(self, di):
        out = CondensedInfo()
        self._ck(self._L.pa_condensed_query(self.h, di, C.byref(out)), "pa_condensed_query")
        return out

    def condensed_triplets(self, di, first, n, cond, g, rows, cols, vals, rhs_rows, rhs_vals):
        self._ck(self._L.pa_condensed_triplets_batch(self.h, di, first, n, cond, g, rows, cols, vals, rhs_rows, rhs_vals),
                 "pa_condensed_triplets_batch")

    def assembler_csr_query(self, di):
        out = AssemblerCsrInfo()
        self._ck(self._L.pa_assembler_csr_query(self.h, di, C.byref(out)), "pa_assembler_csr_query")
        return out

    def assembler_csr_pattern(self, di, rowptr, colind):
        self._ck(self._L.pa_assembler_csr_pattern(self.h, di, rowptr, colind), "pa_assembler_csr_pattern")

    def assembler_csr_fill(self, di, lc, rhs, g, values, RHS):
        self._ck(self._L.pa_assembler_csr_fill(self.h, di, lc, rhs, g, values, RHS), "pa_assembler_csr_fill")

    def condensed_csr_pattern(self, di, rowptr, colind):
        self._ck(self._L.pa_condensed_csr_pattern(self.h, di, rowptr, colind), "pa_condensed_csr_pattern")

    def condensed_csr_fill(self, di, cond, g, halo_below, values, rhs):
        self._ck(self._L.pa_condensed_csr_fill(self.h, di, cond, g, halo_below, values, rhs), "pa_condensed_csr_fill")

    def condensed_halo_pack(self, di, cond, g, halo):
        self._ck(self._L.pa_condensed_halo_pack(self.h, di, cond, g, halo), "pa_condensed_halo_pack")

    def condensed_take_faces(self, di, first, n, solution, g, uF):
        self._ck(self._L.pa_condensed_take_faces(self.h, di, first, n, solution, g, uF), "pa_condensed_take_faces")

    def condensed_expand_solution(self, di, uT, xF, full):
        self._ck(self._L.pa_condensed_expand_solution(self.h, di, uT, xF, full), "pa_condensed_expand_solution")

    def launch_info(self, di, quad, stab, n, condensed=False):
        li = LaunchInfo()
        if condensed:
            self._ck(self._L.pa_condensed_launch_info(self.h, di, quad, stab, n, C.byref(li)), "pa_condensed_launch_info")
        else:
            self._ck(self._L.pa_local_ops_launch_info(self.h, di, quad, stab, n, C.byref(li)), "pa_local_ops_launch_info")
        return li
