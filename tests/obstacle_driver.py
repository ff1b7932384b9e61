"""Restatement of the reference's obstacle driver (apps/obstacle/obstacle.cpp:47-227) and of
parameterised by the provider of the local operators and of the assembler (default: the oracle's C
restatement of obstacle_assembler, src/methods/hho_bits/hho.hpp:471-751).  Used to pin the oracle -- and the GPU path -- end to end against
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


def project_solution(msh, c, degree):
    """project_function(msh, cl, hdi, sol_fun, di=1)  (obstacle.cpp:206, utils.hpp:199-227)."""
    di = o.degrees(0, degree)
    pts = np.ascontiguousarray(msh.cell_pts(c).reshape(8))
    ids = np.ascontiguousarray(msh.ptids[c])
    out = np.zeros(di.msize)
    st = o.lib().hho_project_function(o._dp(pts), o._u64p(ids), di, o.QUAD_TENSOR, o.lib().hho_builtin_fn(4), None, 1, o._dp(out))
    assert st == 0
    return out


def run_obstacle(N, degree, local_provider=oracle_local_provider, max_iter=50, assembler_factory=None):
    """-> (sqrt(error), iterations).  Follows run_hho_obstacle (obstacle.cpp:47-227); the assembler
    is the oracle's C restatement of obstacle_assembler, or `assembler_factory(msh, di, in_A)` ->
    object with assemble_all(lc, rhs, gamma) -> (rows, cols, vals, RHS), expand_solution, num_I,
    num_A, system_size (the GPU path in tests/test_gpu_obstacle.py)."""
    msh = ObstacleMesh(N)
    di = o.degrees(0, degree)
    cbs, fbs = 1, degree + 1
    nc, nf = msh.ncells, msh.nfaces
    lc, rhs = local_provider(msh, degree)          # identical in every outer iteration (obstacle.cpp:148-156)
    msize = cbs + 4 * fbs
    assert lc.shape == (nc, msize, msize)

    alpha = np.zeros(nc + fbs * nf)
    beta = np.ones(nc)
    gamma = np.zeros(nc)                            # obstacle_fun = 0 at the barycenters (:113)
    cpar = 1.0
    it = 0
    while it < max_iter:
        diff = beta + cpar * (alpha[:nc] - gamma)   # :133
        in_A = diff < 0
        if assembler_factory is not None:
            asm = assembler_factory(msh, di, in_A)
            rows, cols, vals, RHS = asm.assemble_all(lc, rhs, gamma)
        else:
            asm = o.ObstacleAssembler(msh.mp, msh.points, msh.ptids, di, in_A, bf_id=4)
            rows, cols, vals = [], [], []
            RHS = np.zeros(asm.system_size)
            for c in range(nc):                     # obstacle.cpp:148-156
                tr, tc, tv, rr, rv = asm.assemble_cell(c, lc[c], rhs[c], gamma)
                rows.append(tr); cols.append(tc); vals.append(tv)
                ok = rr >= 0
                np.add.at(RHS, rr[ok], rv[ok])
            rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
        size = asm.system_size
        LHS = sp.csc_matrix((vals, (rows, cols)), shape=(size, size))               # setFromTriplets sums duplicates
        sol = spla.spsolve(LHS, RHS)
        alpha_prev = alpha
        alpha, beta = asm.expand_solution(sol, gamma)
        if np.linalg.norm(alpha_prev - alpha) < 1e-7:                                # obstacle.cpp:193
            break
        it += 1

    error = 0.0                                     # obstacle.cpp:202-213
    for c in range(nc):
        local = asm.take_local_data(c, alpha)
        d = local - project_solution(msh, c, degree)
        error += d @ (lc[c] @ d)
    return math.sqrt(error), it + 1
