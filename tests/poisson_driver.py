"""Restatement of the assembly + solve + L2-error part of the reference's convergence driver
(apps/convergence_test/convergence_test.cpp:165-274) on top of a provider of local operators and
triplets; the sparse solve is scipy's (host side, outside the hot path)."""
import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle_lib as o


def oracle_assembly(N, cd, fd):
    """-> (LHS csr, RHS, assembler, di, lc) using only the oracle."""
    mp, points, ptids = o.make_mesh(N, N)
    di = o.degrees(cd, fd)
    st, out = o.local_ops_batch(points, ptids, di, o.QUAD_TENSOR, o.STAB_FANCY, fn=1, want=("lc",))
    assert st == 0
    asm = o.Assembler(mp, points, ptids, di, bf_id=2)
    rows, cols, vals = [], [], []
    RHS = np.zeros(asm.system_size)
    for c in range(asm.nc):
        tr, tc, tv, rr, rv = asm.assemble_cell(c, out["lc"][c], out["rhs"][c])
        rows.append(tr); cols.append(tc); vals.append(tv)
        ok = rr >= 0
        np.add.at(RHS, rr[ok], rv[ok])
    LHS = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                        shape=(asm.system_size, asm.system_size))           # setFromTriplets sums duplicates
    return LHS, RHS, asm, di


def l2_error(asm, di, sol):
    """errors_int of convergence_test.cpp:254-268: sum_q w (u(x_q) - u_h(x_q))^2 at degree 2*celdeg."""
    L = o.lib()
    err = 0.0
    cbs = di.cbs
    qx, qy, qw = np.zeros(64), np.zeros(64), np.zeros(64)
    phi = np.zeros(32)
    bar = np.zeros(2)
    for c in range(asm.nc):
        pts = np.ascontiguousarray(asm.points[asm.ptids[c].astype(np.int64)].reshape(8))
        L.hho_cell_barycenter(o._dp(pts), o._dp(bar))
        h = L.hho_cell_diameter(o._dp(pts))
        nq = L.hho_cell_quadrature(o._dp(pts), o.QUAD_TENSOR, 2 * di.cell_deg, o._dp(qx), o._dp(qy), o._dp(qw))
        dofs = sol[c * cbs:(c + 1) * cbs]
        for q in range(nq):
            L.hho_cell_basis_eval(o._dp(bar), h, di.cell_deg, qx[q], qy[q], o._dp(phi))
            val = float(np.dot(dofs, phi[:cbs]))
            real = math.sin(math.pi * qx[q]) * math.sin(math.pi * qy[q])
            err += qw[q] * (real - val) ** 2
    return math.sqrt(err)


def solve(LHS, RHS):
    return spla.spsolve(LHS.tocsc(), RHS)
