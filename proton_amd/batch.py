"""Torch-tensor convenience layer over the C ABI (device memory + streams only).

All compute happens inside libproton_amd.so; torch provides allocations and the stream.
"""
import numpy as np
import torch

from . import capi


def _ptr(t):
    return None if t is None else t.data_ptr()


class BatchAssembler:
    """Batched counterpart of the reference's per-cell loop (convergence_test.cpp:202-213):
    make_hho_laplacian + stabilization (+ make_rhs) for a block of cells, on one GPU."""

    def __init__(self, device=0, use_torch_stream=True):
        if not torch.cuda.is_available():
            raise RuntimeError("proton_amd needs a GPU: torch.cuda.is_available() is False (no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        if use_torch_stream:      # enqueue on torch's current stream (0 = the default stream)
            self.ctx = capi.Context(device, torch.cuda.current_stream(self.device).cuda_stream)
        else:
            self.ctx = capi.Context(device, own_stream=True)
        self._keep = []

    # ---- mesh -----------------------------------------------------------------------
    def set_mesh(self, points, ptids):
        points = np.ascontiguousarray(points, dtype=np.float64)
        ptids = np.ascontiguousarray(ptids)
        assert ptids.max() < 2 ** 32
        self.ctx.mesh_upload(points, ptids.astype(np.uint32))

    def generate_mesh(self, Nx, Ny, lo=(0.0, 0.0), hi=(1.0, 1.0), rows=None):
        self.ctx.mesh_generate(Nx, Ny, lo, hi, rows)

    @property
    def ncells(self):
        return self.ctx.mesh_counts()[1]

    # ---- hot path -------------------------------------------------------------------
    def local_ops(self, cd, fd, quad=capi.QUAD_TENSOR, stab=capi.STAB_FANCY, first=0, n=None,
                  want=("lc",), out=None):
        """Returns dict of cell-major torch tensors [n, cols, rows] (column-major matrices:
        entry (i, j) is t[c, j, i]); `out` may carry preallocated tensors to reuse."""
        di, _ = capi.degree_info(cd, fd)
        sz = capi.sizes_for(di, quad)
        if n is None:
            n = self.ncells - first
        out = dict(out or {})
        shapes = {"oper": (n, sz.msize, sz.oper_rows), "data": (n, sz.msize, sz.msize),
                  "stab": (n, sz.msize, sz.msize), "lc": (n, sz.msize, sz.msize)}
        for k in want:
            if k == "info":
                if "info" not in out:
                    out["info"] = torch.empty(n, dtype=torch.int32, device=self.device)
            elif k not in out:
                out[k] = torch.empty(shapes[k], dtype=torch.float64, device=self.device)
        self.ctx.local_ops(di, quad, stab, first, n, _ptr(out.get("oper")), _ptr(out.get("data")),
                           _ptr(out.get("stab")), _ptr(out.get("lc")), _ptr(out.get("info")))
        return out

    def cell_rhs(self, degree, fn, quad=capi.QUAD_TENSOR, dinc=0, first=0, n=None, fvals=None, out=None):
        if n is None:
            n = self.ncells - first
        cbs = (degree + 2) * (degree + 1) // 2
        if out is None:
            out = torch.empty((n, cbs), dtype=torch.float64, device=self.device)
        self.ctx.cell_rhs(degree, dinc, quad, fn, first, n, out.data_ptr(), _ptr(fvals))
        return out

    def quadrature_points(self, degree, quad=capi.QUAD_TENSOR, first=0, n=None):
        if n is None:
            n = self.ncells - first
        nq = self.ctx.cell_quadrature_points(degree, quad, first, n, None)
        out = torch.empty((n, nq, 3), dtype=torch.float64, device=self.device)
        self.ctx.cell_quadrature_points(degree, quad, first, n, out.data_ptr())
        return out

    def static_condensation(self, cd, fd, lc, rhs=None):
        di, _ = capi.degree_info(cd, fd)
        sz = capi.sizes_for(di, capi.QUAD_TENSOR)
        n, nf = lc.shape[0], 4 * sz.fbs
        S = torch.empty((n, nf, nf), dtype=torch.float64, device=self.device)
        g = torch.empty((n, nf), dtype=torch.float64, device=self.device)
        rec = torch.empty((n, nf + 1, sz.cbs), dtype=torch.float64, device=self.device)
        info = torch.empty(n, dtype=torch.int32, device=self.device)
        self.ctx.static_condensation(di, n, lc.data_ptr(), _ptr(rhs), S.data_ptr(), g.data_ptr(), rec.data_ptr(),
                                     info.data_ptr())
        return S, g, rec, info

    # ---- condensed mode (static condensation fused into the local-operator pass) ------
    def condensed_ops(self, cd, fd, quad=capi.QUAD_TENSOR, stab=capi.STAB_FANCY, rhs=None, first=0, n=None, out=None, want_info=False):
        """-> packed records [n, nf(nf+1)/2 + nf]: upper triangle of the Schur complement (column-packed), then g."""
        di, _ = capi.degree_info(cd, fd)
        n = self.ncells - first if n is None else n
        nf = 4 * (di.face_deg + 1)
        if out is None:
            out = torch.empty((n, nf * (nf + 1) // 2 + nf), dtype=torch.float64, device=self.device)
        info = torch.empty(n, dtype=torch.int32, device=self.device) if want_info else None
        self.ctx.condensed_ops(di, quad, stab, first, n, _ptr(rhs), out.data_ptr(), _ptr(info))
        return (out, info) if want_info else out

    def condensed_recover(self, cd, fd, uF, quad=capi.QUAD_TENSOR, stab=capi.STAB_FANCY, rhs=None, first=0, n=None):
        """u_T = A_TT^-1 (f_T - A_TF u_F) for cells [first, first+n) -> [n, cbs]."""
        di, _ = capi.degree_info(cd, fd)
        n = self.ncells - first if n is None else n
        cbs = (di.cell_deg + 1) * (di.cell_deg + 2) // 2
        uT = torch.empty((n, cbs), dtype=torch.float64, device=self.device)
        self.ctx.condensed_recover(di, quad, stab, first, n, _ptr(rhs), uF.data_ptr(), uT.data_ptr(), None)
        return uT

    def condensed_info(self, cd, fd):
        di, _ = capi.degree_info(cd, fd)
        return self.ctx.condensed_query(di)

    def condensed_triplets(self, cd, fd, cond, g=None, first=0):
        di, _ = capi.degree_info(cd, fd)
        n, nf = cond.shape[0], 4 * (di.face_deg + 1)
        rows = torch.empty((n, nf * nf), dtype=torch.int32, device=self.device)
        cols = torch.empty((n, nf * nf), dtype=torch.int32, device=self.device)
        vals = torch.empty((n, nf * nf), dtype=torch.float64, device=self.device)
        rhs_rows = torch.empty((n, nf), dtype=torch.int32, device=self.device)
        rhs_vals = torch.empty((n, nf), dtype=torch.float64, device=self.device)
        self.ctx.condensed_triplets(di, first, n, cond.data_ptr(), _ptr(g), rows.data_ptr(), cols.data_ptr(), vals.data_ptr(),
                                    rhs_rows.data_ptr(), rhs_vals.data_ptr())
        return rows, cols, vals, rhs_rows, rhs_vals

    def condensed_csr_pattern(self, cd, fd):
        """symbolic phase -> (rowptr int64 [rows+1], colind int32 [nnz]) of the rows this context owns"""
        di, _ = capi.degree_info(cd, fd)
        ci = self.ctx.condensed_query(di)
        rowptr = torch.empty(ci.row_end - ci.row_begin + 1, dtype=torch.int64, device=self.device)
        colind = torch.empty(max(ci.nnz_owned, 1), dtype=torch.int32, device=self.device)
        self.ctx.condensed_csr_pattern(di, rowptr.data_ptr(), colind.data_ptr())
        return rowptr, colind[:ci.nnz_owned]

    def condensed_csr_fill(self, cd, fd, cond, g=None, halo_below=None, values=None, rhs=None):
        """numeric phase -> (values [nnz], rhs [rows])"""
        di, _ = capi.degree_info(cd, fd)
        ci = self.ctx.condensed_query(di)
        if values is None:
            values = torch.empty(max(ci.nnz_owned, 1), dtype=torch.float64, device=self.device)
        if rhs is None:
            rhs = torch.empty(max(ci.row_end - ci.row_begin, 1), dtype=torch.float64, device=self.device)
        self.ctx.condensed_csr_fill(di, cond.data_ptr(), _ptr(g), _ptr(halo_below), values.data_ptr(), rhs.data_ptr())
        return values[:ci.nnz_owned], rhs[:ci.row_end - ci.row_begin]

    def condensed_halo_pack(self, cd, fd, cond, g=None, out=None):
        di, _ = capi.degree_info(cd, fd)
        ci = self.ctx.condensed_query(di)
        if out is None:
            out = torch.empty((max(ci.halo_cells, 1), ci.halo_doubles), dtype=torch.float64, device=self.device)
        self.ctx.condensed_halo_pack(di, cond.data_ptr(), _ptr(g), out.data_ptr())
        return out[:ci.halo_cells]

    def condensed_take_faces(self, cd, fd, solution, g=None, first=0, n=None):
        di, _ = capi.degree_info(cd, fd)
        n = self.ncells - first if n is None else n
        uF = torch.empty((n, 4 * (di.face_deg + 1)), dtype=torch.float64, device=self.device)
        self.ctx.condensed_take_faces(di, first, n, solution.data_ptr(), _ptr(g), uF.data_ptr())
        return uF

    def condensed_expand_solution(self, cd, fd, uT, xF, full=None):
        di, _ = capi.degree_info(cd, fd)
        info = self.ctx.assembler_query(di)
        if full is None:
            full = torch.zeros(info.system_size, dtype=torch.float64, device=self.device)
        self.ctx.condensed_expand_solution(di, uT.data_ptr(), _ptr(xF), full.data_ptr())
        return full

    # ---- assembler (hho.hpp:252-463) -------------------------------------------------
    def set_faces(self, cell_faces, face_pts, face_is_dirichlet):
        self.ctx.mesh_set_faces(cell_faces, face_pts, face_is_dirichlet)

    def assembler_info(self, cd, fd):
        di, _ = capi.degree_info(cd, fd)
        return self.ctx.assembler_query(di)

    def dirichlet_data(self, fd, fn, fvals=None):
        info = self.assembler_info(fd, fd)
        g = torch.empty((info.nfaces_local, fd + 1), dtype=torch.float64, device=self.device)
        self.ctx.dirichlet_data(fd, fn, g.data_ptr(), _ptr(fvals))
        return g

    def face_quadrature_points(self, fd):
        info = self.assembler_info(0, 0)
        out = torch.empty((info.nfaces_local, fd + 1, 3), dtype=torch.float64, device=self.device)
        self.ctx.face_quadrature_points(fd, out.data_ptr())
        return out

    def triplets(self, cd, fd, lc, rhs=None, g=None, first=0):
        """assembler::assemble for the cells lc covers -> (rows, cols, vals, rhs_rows, rhs_vals)."""
        di, _ = capi.degree_info(cd, fd)
        n, ms = lc.shape[0], lc.shape[1]
        rows = torch.empty((n, ms * ms), dtype=torch.int32, device=self.device)
        cols = torch.empty((n, ms * ms), dtype=torch.int32, device=self.device)
        vals = torch.empty((n, ms * ms), dtype=torch.float64, device=self.device)
        rhs_rows = torch.empty((n, ms), dtype=torch.int32, device=self.device)
        rhs_vals = torch.empty((n, ms), dtype=torch.float64, device=self.device)
        self.ctx.triplets(di, first, n, lc.data_ptr(), _ptr(rhs), _ptr(g), rows.data_ptr(), cols.data_ptr(),
                          vals.data_ptr(), rhs_rows.data_ptr(), rhs_vals.data_ptr())
        return rows, cols, vals, rhs_rows, rhs_vals

    def assembler_csr_pattern(self, cd, fd):
        """assembler<Mesh>'s own system (cell + face unknowns) directly in CSR, symbolic phase -> (rowptr int64 [nrows+1], colind int32 [nnz])"""
        di, _ = capi.degree_info(cd, fd)
        info = self.ctx.assembler_csr_query(di)
        rowptr = torch.empty(info.nrows + 1, dtype=torch.int64, device=self.device)
        colind = torch.empty(max(info.nnz, 1), dtype=torch.int32, device=self.device)
        self.ctx.assembler_csr_pattern(di, rowptr.data_ptr(), colind.data_ptr())
        return rowptr, colind[:info.nnz]

    def assembler_csr_fill(self, cd, fd, lc, rhs=None, g=None, values=None, RHS=None):
        """numeric phase of the same -> (values [nnz], RHS [nrows])"""
        di, _ = capi.degree_info(cd, fd)
        info = self.ctx.assembler_csr_query(di)
        if values is None:
            values = torch.empty(max(info.nnz, 1), dtype=torch.float64, device=self.device)
        if RHS is None:
            RHS = torch.empty(max(info.nrows, 1), dtype=torch.float64, device=self.device)
        self.ctx.assembler_csr_fill(di, lc.data_ptr(), _ptr(rhs), _ptr(g), values.data_ptr(), RHS.data_ptr())
        return values[:info.nnz], RHS[:info.nrows]

    def csr_from_triplets(self, rows, cols, vals, nrows):
        """setFromTriplets on the device -> (rowptr int64 [nrows+1], colind int32 [nnz], values [nnz])."""
        rows, cols, vals = rows.reshape(-1), cols.reshape(-1), vals.reshape(-1)
        n = rows.numel()
        rowptr = torch.empty(nrows + 1, dtype=torch.int64, device=self.device)
        colind = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        values = torch.empty(max(n, 1), dtype=torch.float64, device=self.device)
        nnz = self.ctx.csr_from_triplets(n, rows.data_ptr(), cols.data_ptr(), vals.data_ptr(), nrows, rowptr.data_ptr(),
                                         colind.data_ptr(), values.data_ptr())
        return rowptr, colind[:nnz], values[:nnz]

    def conjugated_gradient(self, rowptr, colind, values, b, tol=1e-9, div=100.0, max_iter=1000, precond=True):
        """solver_cg.hpp:63-144 on the device -> (x, exit_reason, iterations, relative residual)"""
        x = torch.empty_like(b)
        reason, iters, rr = self.ctx.conjugated_gradient(b.numel(), rowptr.data_ptr(), colind.data_ptr(), values.data_ptr(),
                                                         b.data_ptr(), x.data_ptr(), tol, div, max_iter, precond)
        return x, reason, iters, rr

    def take_local_data(self, cd, fd, solution, g=None, first=0, n=None):
        """assembler::take_local_data (hho.hpp:408-449) for cells [first, first+n) -> n x msize."""
        di, _ = capi.degree_info(cd, fd)
        n = self.ncells - first if n is None else n
        ms = (cd + 1) * (cd + 2) // 2 + 4 * (fd + 1)
        out = torch.empty((n, ms), dtype=torch.float64, device=self.device)
        self.ctx.take_local_data(di, first, n, solution.data_ptr(), _ptr(g), out.data_ptr())
        return out

    def project_function(self, cd, fd, fn, quad=capi.QUAD_TENSOR, dinc=0, cell_fvals=None, face_fvals=None, first=0, n=None):
        """project_function(msh, cl, hdi, f, di) (utils.hpp:199-227) -> n x msize."""
        di, _ = capi.degree_info(cd, fd)
        n = self.ncells - first if n is None else n
        ms = (di.cell_deg + 1) * (di.cell_deg + 2) // 2 + 4 * (di.face_deg + 1)
        out = torch.empty((n, ms), dtype=torch.float64, device=self.device)
        self.ctx.project_function(di, quad, dinc, fn, _ptr(cell_fvals), _ptr(face_fvals), first, n, out.data_ptr(), None)
        return out

    def energy_form(self, cd, fd, lc, u, v=None):
        """per-cell (u - v)^T lc (u - v)."""
        di, _ = capi.degree_info(cd, fd)
        out = torch.empty(lc.shape[0], dtype=torch.float64, device=self.device)
        self.ctx.energy_form(di, lc.shape[0], lc.data_ptr(), u.data_ptr(), _ptr(v), out.data_ptr())
        return out

    # ---- obstacle_assembler (hho.hpp:471-751) -------------------------------------------
    def obstacle_tables(self, in_A):
        """-> (A_ct, B_ct, num_I, num_A) for the uint8 flags in_A (hho.hpp:538-578)."""
        A_ct = torch.empty(self.ncells, dtype=torch.int32, device=self.device)
        B_ct = torch.empty(self.ncells, dtype=torch.int32, device=self.device)
        num_I, num_A = self.ctx.obstacle_tables(in_A.data_ptr(), A_ct.data_ptr(), B_ct.data_ptr())
        return A_ct, B_ct, num_I, num_A

    def obstacle_triplets(self, cd, fd, lc, rhs, g, gamma, in_A, A_ct, B_ct, num_I, first=0):
        """obstacle_assembler::assemble -> (rows, cols, vals) n x (msize^2 + 1), rhs_rows, rhs_vals n x msize."""
        di, _ = capi.degree_info(cd, fd)
        n, ms = lc.shape[0], lc.shape[1]
        rows = torch.empty((n, ms * ms + 1), dtype=torch.int32, device=self.device)
        cols = torch.empty((n, ms * ms + 1), dtype=torch.int32, device=self.device)
        vals = torch.empty((n, ms * ms + 1), dtype=torch.float64, device=self.device)
        rhs_rows = torch.empty((n, ms), dtype=torch.int32, device=self.device)
        rhs_vals = torch.empty((n, ms), dtype=torch.float64, device=self.device)
        self.ctx.obstacle_triplets(di, first, n, lc.data_ptr(), _ptr(rhs), _ptr(g), gamma.data_ptr(), in_A.data_ptr(),
                                   A_ct.data_ptr(), B_ct.data_ptr(), num_I, rows.data_ptr(), cols.data_ptr(),
                                   vals.data_ptr(), rhs_rows.data_ptr(), rhs_vals.data_ptr())
        return rows, cols, vals, rhs_rows, rhs_vals

    def obstacle_expand_solution(self, cd, fd, solution, g, gamma, in_A, A_ct, B_ct, num_I, nfaces):
        """obstacle_assembler::expand_solution -> (alpha, beta)."""
        di, _ = capi.degree_info(cd, fd)
        cbs = (cd + 1) * (cd + 2) // 2
        alpha = torch.empty(self.ncells * cbs + nfaces * (fd + 1), dtype=torch.float64, device=self.device)
        beta = torch.empty(self.ncells * cbs, dtype=torch.float64, device=self.device)
        self.ctx.obstacle_expand_solution(di, solution.data_ptr(), _ptr(g), gamma.data_ptr(), in_A.data_ptr(),
                                          A_ct.data_ptr(), B_ct.data_ptr(), num_I, alpha.data_ptr(), beta.data_ptr())
        return alpha, beta

    def obstacle_take_local_data(self, cd, fd, expanded, first=0, n=None):
        """free take_local_data(msh, cl, di, expanded_solution) (hho.hpp:753-782) -> n x msize."""
        di, _ = capi.degree_info(cd, fd)
        n = self.ncells - first if n is None else n
        ms = (cd + 1) * (cd + 2) // 2 + 4 * (fd + 1)
        out = torch.empty((n, ms), dtype=torch.float64, device=self.device)
        self.ctx.obstacle_take_local_data(di, first, n, expanded.data_ptr(), out.data_ptr())
        return out

    # ---- cutHHO fictitious domain (cuthho_square -f) ------------------------------------
    def cut_preprocess(self, N, radius=0.35, center=(0.5, 0.5), refsteps=4, rows=None, line_y=None):
        """cuthho_square.cpp:2026-2052: mesh, circle level set (or, line_y given, line_level_set y - line_y, :91-124), default (-D)
        preprocessing.  rows = (row_begin, row_end): the context keeps that slab of cell rows (pa_cut_preprocess_rows)."""
        self.level_set = capi.LevelSet(0, radius, center[0], center[1], 0.0) if line_y is None else capi.LevelSet(1, 0.0, 0.0, 0.0, line_y)
        self.ctx.cut_preprocess(N, N, self.level_set, refsteps, rows=rows)
        self.ncut, self.cell_loc, self.cut_index = self.ctx.cut_query()
        return self.ncut

    def cut_local_ops(self, fd, where=capi.LOC_NEGATIVE, rhs_fn=capi.FN_SIN_SIN_RHS, bcs_fn=capi.FN_SIN_SIN_SOL,
                      want=("oper", "data", "stab", "lc", "rhs", "info")):
        cbs = (fd + 3) * (fd + 2) // 2
        ms = cbs + 4 * (fd + 1)
        n = self.ncut
        shapes = {"oper": (n, ms, cbs), "data": (n, ms, ms), "stab": (n, ms, ms), "lc": (n, ms, ms), "rhs": (n, cbs)}
        out = {}
        for k in want:
            out[k] = torch.empty(n, dtype=torch.int32, device=self.device) if k == "info" else \
                torch.empty(shapes[k], dtype=torch.float64, device=self.device)
        self.ctx.cut_local_ops(fd, self.level_set, where, rhs_fn, bcs_fn, _ptr(out.get("oper")), _ptr(out.get("data")),
                               _ptr(out.get("stab")), _ptr(out.get("lc")), _ptr(out.get("rhs")), _ptr(out.get("info")))
        return out

    def fictdom_local_ops(self, fd, where=capi.LOC_NEGATIVE, rhs_fn=capi.FN_SIN_SIN_RHS, bcs_fn=capi.FN_SIN_SIN_SOL,
                          overlap=False):
        """The whole loop body of cuthho_square.cpp:883-900 for every cell: uncut cells through the
        fan-quadrature / naive-stabilization kernel, cut cells through the cut kernel, merged.
        overlap: the cut cells' kernel first, on the context's side stream (pa_context_set_cut_overlap), next to
        the uncut cells' kernels.  -> (lc [n, ms, ms], rhs [n, cbs]) device tensors."""
        cd = fd + 1
        cut = None
        if overlap:
            self.ctx.set_cut_overlap(True)
            if self.ncut:
                cut = self.cut_local_ops(fd, where, rhs_fn, bcs_fn, want=("lc", "rhs"))
        out = self.local_ops(cd, fd, capi.QUAD_FAN, capi.STAB_NAIVE, want=("lc",))
        rhs = self.cell_rhs(cd, rhs_fn, capi.QUAD_FAN)
        if self.ncut:
            if cut is None:
                cut = self.cut_local_ops(fd, where, rhs_fn, bcs_fn, want=("lc", "rhs"))
            self.ctx.cut_merge(fd, where, cut["lc"].data_ptr(), cut["rhs"].data_ptr(), out["lc"].data_ptr(), rhs.data_ptr())
        else:
            self.ctx.cut_merge(fd, where, None, None, None, rhs.data_ptr())
        if overlap:
            self.ctx.set_cut_overlap(False)
        return out["lc"], rhs

    def fictdom_condensed_ops(self, fd, where=capi.LOC_NEGATIVE, rhs_fn=capi.FN_SIN_SIN_RHS, bcs_fn=capi.FN_SIN_SIN_SOL):
        """The same loop body in condensed mode: packed records [n, nf(nf+1)/2 + nf] of every cell of the context -- the fused
        pass with the uncut formulas (fan quadrature, naive stabilization), the cut cells' records from the stand-alone
        condensation of their cut operators (pa_static_condensation_packed_batch + pa_cut_merge_condensed).
        -> (records, rhs [n, cbs])."""
        cd = fd + 1
        di, _ = capi.degree_info(cd, fd)
        nf = 4 * (fd + 1)
        f64 = dict(dtype=torch.float64, device=self.device)
        rhs = torch.empty((self.ncells, (cd + 1) * (cd + 2) // 2), **f64)
        self.ctx.cut_uncut_rhs(cd, where, rhs_fn, rhs.data_ptr())
        cut = None
        if self.ncut:
            cut = self.cut_local_ops(fd, where, rhs_fn, bcs_fn, want=("lc", "rhs"))
            Sp = torch.empty((self.ncut, nf * (nf + 1) // 2), **f64)
            g = torch.empty((self.ncut, nf), **f64)
            self.ctx.static_condensation_packed(di, self.ncut, cut["lc"].data_ptr(), cut["rhs"].data_ptr(), Sp.data_ptr(), g.data_ptr(), None)
        self.ctx.cut_merge(fd, where, None, None if cut is None else cut["rhs"].data_ptr(), None, rhs.data_ptr())
        rec = self.condensed_ops(cd, fd, capi.QUAD_FAN, capi.STAB_NAIVE, rhs=rhs)
        if self.ncut:
            self.ctx.cut_merge_condensed(fd, Sp.data_ptr(), g.data_ptr(), rec.data_ptr())
        return rec, rhs

    # ---- cutHHO two-sided interface problem (cuthho_square -i) ------------------------------
    def interface_local_ops(self, fd, kappa=(1.0, 1.0), eta=5.0, rhs_fn=capi.FN_SIN_SIN_RHS, want_oper=False):
        """-> dict(lc, rhs: all cells, uncut formulas; lc_cut, rhs_cut [, oper_cut, data_cut]: cut cells)."""
        parms = capi.InterfaceParams(kappa[0], kappa[1], eta)
        cbs = (fd + 3) * (fd + 2) // 2
        ms = cbs + 4 * (fd + 1)
        n, nc = self.ncut, self.ncells
        f64 = dict(dtype=torch.float64, device=self.device)
        out = {"lc": torch.empty((nc, ms, ms), **f64), "rhs": torch.empty((nc, cbs), **f64),
               "lc_cut": torch.empty((n, 2 * ms, 2 * ms), **f64), "rhs_cut": torch.empty((n, 2 * cbs), **f64),
               "info_cut": torch.empty(n, dtype=torch.int32, device=self.device)}
        if want_oper:
            out["oper_cut"] = torch.empty((n, 2 * ms, 2 * cbs), **f64)
            out["data_cut"] = torch.empty((n, 2 * ms, 2 * ms), **f64)
        self.ctx.cut_interface_uncut(fd, parms, rhs_fn, out["lc"].data_ptr(), out["rhs"].data_ptr(), None)
        self.ctx.cut_interface_ops(fd, self.level_set, parms, rhs_fn, _ptr(out.get("oper_cut")), _ptr(out.get("data_cut")),
                                   out["lc_cut"].data_ptr(), out["rhs_cut"].data_ptr(), out["info_cut"].data_ptr())
        return out

    def interface_triplets(self, fd, ops, g=None):
        """interface_assembler::assemble / assemble_cut -> dict of device arrays."""
        cbs = (fd + 3) * (fd + 2) // 2
        ms = cbs + 4 * (fd + 1)
        n, nc = self.ncut, self.ncells
        i32 = dict(dtype=torch.int32, device=self.device)
        f64 = dict(dtype=torch.float64, device=self.device)
        t = {"rows": torch.empty((nc, ms * ms), **i32), "cols": torch.empty((nc, ms * ms), **i32), "vals": torch.empty((nc, ms * ms), **f64),
             "rows_cut": torch.empty((n, 4 * ms * ms), **i32), "cols_cut": torch.empty((n, 4 * ms * ms), **i32),
             "vals_cut": torch.empty((n, 4 * ms * ms), **f64),
             "rhs_rows": torch.empty((nc, ms), **i32), "rhs_vals": torch.empty((nc, ms), **f64),
             "rhs_rows_cut": torch.empty((n, 2 * ms), **i32), "rhs_vals_cut": torch.empty((n, 2 * ms), **f64)}
        self.ctx.interface_triplets(fd, ops["lc"].data_ptr(), ops["rhs"].data_ptr(), _ptr(g), ops["lc_cut"].data_ptr(),
                                    ops["rhs_cut"].data_ptr(), t["rows"].data_ptr(), t["cols"].data_ptr(), t["vals"].data_ptr(),
                                    t["rows_cut"].data_ptr(), t["cols_cut"].data_ptr(), t["vals_cut"].data_ptr(),
                                    t["rhs_rows"].data_ptr(), t["rhs_vals"].data_ptr(), t["rhs_rows_cut"].data_ptr(),
                                    t["rhs_vals_cut"].data_ptr())
        return t

    def interface_cell_offsets(self, fd):
        out = torch.empty((self.ncells, 2), dtype=torch.int64, device=self.device)
        self.ctx.interface_cell_offsets(fd, out.data_ptr())
        return out

    def synchronize(self):
        self.ctx.synchronize()


def to_rowcol(t):
    """[n, cols, rows] device tensor of column-major matrices -> numpy [n, rows, cols]."""
    return t.detach().cpu().numpy().swapaxes(1, 2).copy()

