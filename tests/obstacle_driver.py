"""Restatement of the reference's obstacle driver (apps/obstacle/obstacle.cpp:47-227) and of
obstacle_assembler (src/methods/hho_bits/hho.hpp:471-751) in numpy/scipy, parameterised by the
provider of the local operators.  Used to pin the oracle -- and the GPU path -- end to end against
the only numbers the reference commits for this path: apps/obstacle/results/convergence.txt.

The sparse direct solve (Eigen SparseLU in the reference, obstacle.cpp:170-175) is
scipy.sparse.linalg.spsolve here: it is outside the hot path and "unchanged, host-side".
"""
import ctypes as C
import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle_lib as o

R0 = 0.7


def rhs_fun(x, y):                       # obstacle.cpp:67-74
    r = math.sqrt(x * x + y * y)
    if r > R0:
        return -16 * r * r + 8 * R0 * R0
    return -8.0 * (R0 * R0 * (R0 * R0 + 1)) + 8 * R0 * R0 * r * r


def sol_fun(x, y):                       # obstacle.cpp:76-81
    r = math.sqrt(x * x + y * y)
    s = r * r - R0 * R0
    t = max(s, 0.0)
    return t * t


class ObstacleMesh:
    """quad_mesh on [-1,1]^2 (obstacle.cpp:234-238,276) with the generator's numbering."""

    def __init__(self, N):
        self.N = N
        self.mp, self.points, self.ptids = o.make_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
        self.faces, self.bnd = o.mesh_faces(self.mp)
        L = o.lib()
        nc = N * N
        self.cell_faces = np.zeros((nc, 4), dtype=np.int64)
        for j in range(N):
            for i in range(N):
                for lf in range(4):
                    self.cell_faces[j * N + i, lf] = L.hho_mesh_face_id(C.byref(self.mp), i, j, lf)
        self.ncells, self.nfaces = nc, self.faces.shape[0]

    def cell_pts(self, c):
        return self.points[self.ptids[c].astype(np.int64)]


def oracle_local_provider(msh, degree):
    """lc = data + fancy stab and the cell rhs (di = 1) from the CPU oracle."""
    di = o.degrees(0, degree)
    st, out = o.local_ops_batch(msh.points, msh.ptids, di, o.QUAD_TENSOR, o.STAB_FANCY, fn=3, rhs_di=1, want=("lc",))
    assert st == 0
    return out["lc"], out["rhs"]


def dirichlet_projection(msh, face_id, degree):
    """mass.llt().solve(rhs) on a face (hho.hpp:657-659)."""
    L = o.lib()
    fbs = degree + 1
    p0 = np.ascontiguousarray(msh.points[int(msh.faces[face_id, 0])])
    p1 = np.ascontiguousarray(msh.points[int(msh.faces[face_id, 1])])
    mass = np.zeros((fbs, fbs))
    rhs = np.zeros(fbs)
    fn = L.hho_builtin_fn(4)
    assert L.hho_face_mass_matrix(o._dp(p0), o._dp(p1), degree, 0, o._dp(mass)) == 0
    assert L.hho_face_rhs(o._dp(p0), o._dp(p1), degree, 0, fn, None, o._dp(rhs)) == 0
    assert L.hho_llt_factor(o._dp(mass), fbs) == 0
    L.hho_llt_solve_inplace(o._dp(mass), fbs, o._dp(rhs), 1)
    return rhs


def project_solution(msh, c, degree):
    """project_function(msh, cl, hdi, sol_fun, di=1)  (obstacle.cpp:206, utils.hpp:199-227)."""
    di = o.degrees(0, degree)
    pts = np.ascontiguousarray(msh.cell_pts(c).reshape(8))
    ids = np.ascontiguousarray(msh.ptids[c])
    out = np.zeros(di.msize)
    st = o.lib().hho_project_function(o._dp(pts), o._u64p(ids), di, o.QUAD_TENSOR, o.lib().hho_builtin_fn(4), None, 1, o._dp(out))
    assert st == 0
    return out


def run_obstacle(N, degree, local_provider=oracle_local_provider, max_iter=50):
    """-> (sqrt(error), iterations).  Follows run_hho_obstacle line by line."""
    msh = ObstacleMesh(N)
    cbs, fbs = 1, degree + 1
    nc, nf = msh.ncells, msh.nfaces
    is_dir = msh.bnd.astype(bool)
    num_other = int((~is_dir).sum())
    face_ct = np.full(nf, -1, dtype=np.int64)
    face_ct[~is_dir] = np.arange(num_other)

    lc, rhs = local_provider(msh, degree)          # identical in every outer iteration (obstacle.cpp:148-156)
    msize = cbs + 4 * fbs
    assert lc.shape == (nc, msize, msize)

    dir_data = {f: dirichlet_projection(msh, f, degree) for f in np.nonzero(is_dir)[0]}

    alpha = np.zeros(nc + fbs * nf)
    beta = np.ones(nc)
    gamma = np.zeros(nc)                            # obstacle_fun = 0 at the barycenters (:113)
    cpar = 1.0
    it = 0
    while it < max_iter:
        diff = beta + cpar * (alpha[:nc] - gamma)   # :133
        in_A = diff < 0
        num_A, num_I = int(in_A.sum()), int((~in_A).sum())
        A_ct = np.full(nc, -1, dtype=np.int64)
        A_ct[~in_A] = np.arange(num_I)
        B_ct = np.full(nc, -1, dtype=np.int64)
        B_ct[in_A] = np.arange(num_A)
        size = cbs * nc + fbs * num_other
        rows, cols, vals = [], [], []
        RHS = np.zeros(size)
        for c in range(nc):                         # obstacle_assembler::assemble  hho.hpp:609-695
            fcs = msh.cell_faces[c]
            row_idx = [c + i for i in range(cbs)]                                   # :631 (cell_offset + i)
            row_ok = [True] * cbs
            col_idx = [A_ct[c] * cbs + i for i in range(cbs)]                       # :625,632
            col_ok = [not in_A[c]] * cbs
            ddata = np.zeros(msize)
            for lf in range(4):
                f = fcs[lf]
                d = bool(is_dir[f])
                for i in range(fbs):
                    row_idx.append(cbs * nc + face_ct[f] * fbs + i)                 # :644
                    col_idx.append(cbs * num_I + face_ct[f] * fbs + i)              # :645
                    row_ok.append(not d)
                    col_ok.append(not d)
                if d:
                    ddata[cbs + lf * fbs: cbs + (lf + 1) * fbs] = dir_data[f]
            A = lc[c]
            for i in range(msize):
                if not row_ok[i]:
                    continue
                for j in range(msize):
                    if col_ok[j]:
                        rows.append(row_idx[i]); cols.append(col_idx[j]); vals.append(A[i, j])
                    elif j < cbs:
                        RHS[row_idx[i]] -= A[i, j] * gamma[c]                        # :677
                    else:
                        RHS[row_idx[i]] -= A[i, j] * ddata[j]                        # :679
            RHS[c:c + cbs] += rhs[c, :cbs]                                           # :686
            if in_A[c]:
                rows.append(c * cbs); cols.append(num_I * cbs + num_other * fbs + B_ct[c]); vals.append(1.0)   # :688-693
        LHS = sp.csc_matrix((vals, (rows, cols)), shape=(size, size))               # setFromTriplets sums duplicates
        sol = spla.spsolve(LHS, RHS)

        alpha_prev = alpha.copy()                   # expand_solution  hho.hpp:698-744
        alpha[:nc] = gamma
        I_cells = np.nonzero(~in_A)[0]
        alpha[I_cells] = sol[:num_I]
        beta = np.zeros(nc)
        A_cells = np.nonzero(in_A)[0]
        beta[A_cells] = sol[num_I * cbs + num_other * fbs: num_I * cbs + num_other * fbs + num_A]
        for f in range(nf):
            if is_dir[f]:
                alpha[nc * cbs + f * fbs: nc * cbs + (f + 1) * fbs] = dir_data[f]
            else:
                o0 = cbs * num_I + face_ct[f] * fbs
                alpha[nc * cbs + f * fbs: nc * cbs + (f + 1) * fbs] = sol[o0:o0 + fbs]
        if np.linalg.norm(alpha_prev - alpha) < 1e-7:                                # obstacle.cpp:193
            break
        it += 1

    error = 0.0                                     # obstacle.cpp:202-213
    for c in range(nc):
        local = np.zeros(msize)
        local[0] = alpha[c]
        for lf in range(4):
            f = msh.cell_faces[c, lf]
            local[cbs + lf * fbs: cbs + (lf + 1) * fbs] = alpha[cbs * nc + f * fbs: cbs * nc + (f + 1) * fbs]
        d = local - project_solution(msh, c, degree)
        error += d @ (lc[c] @ d)
    return math.sqrt(error), it + 1
