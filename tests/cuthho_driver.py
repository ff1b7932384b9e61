"""Restatement of the fictitious-domain driver (apps/cuthho/cuthho_square.cpp:806-1080, `-f`) on the
oracle: preprocessing, per-cell cut/uncut operators, the generic assembler (hho.hpp:252-463), a
sparse direct solve (scipy instead of Eigen SparseLU) and the energy-norm error of :1030-1049."""
import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle_lib as o


def oracle_cut_provider(msh, di):
    """-> list of (lc, rhs) per cell from the oracle's cut operators."""
    out = []
    for c in range(msh.nc):
        st, oper, data = msh.laplacian(c, di)
        assert st == 0, (c, st)
        st, stab = msh.cut_stabilization(c, di)
        assert st == 0
        st, f = msh.rhs(c, di.cell_deg)
        assert st == 0
        out.append((data + stab, f))
    return out


def run_fictdom(N, k, refsteps=4, provider=oracle_cut_provider):
    msh = o.CutMesh(N, refsteps=refsteps)
    di = o.degrees(k + 1, k)                                   # cuthho_square.cpp:871
    mp = o.MeshParams(N, N, 0.0, 1.0, 0.0, 1.0)
    asm = o.Assembler(mp, msh.points, msh.ptids, di, bf_id=2)   # make_assembler + bcs_fun = sol_fun
    local = provider(msh, di)
    rows, cols, vals = [], [], []
    RHS = np.zeros(asm.system_size)
    for c in range(msh.nc):
        lc, f = local[c]
        tr, tc, tv, rr, rv = asm.assemble_cell(c, lc, f)
        rows.append(tr); cols.append(tc); vals.append(tv)
        ok = rr >= 0
        np.add.at(RHS, rr[ok], rv[ok])
    LHS = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                        shape=(asm.system_size, asm.system_size))
    sol = spla.spsolve(LHS, RHS)

    # energy-norm error, cuthho_square.cpp:1030-1049
    L = o.lib()
    cbs, rd = di.cbs, di.rec_deg
    H1 = 0.0
    gx, gy, bar = np.zeros(32), np.zeros(32), np.zeros(2)
    for c in range(msh.nc):
        if msh.cell_loc[c] == o.CUT_POS:
            continue
        pts = np.ascontiguousarray(msh.points[msh.ptids[c].astype(np.int64)].reshape(8))
        L.hho_cell_barycenter(o._dp(pts), o._dp(bar))
        h = L.hho_cell_diameter(o._dp(pts))
        dofs = sol[c * cbs:(c + 1) * cbs]
        qx, qy, qw = msh.cell_quadrature(c, 2 * di.cell_deg, o.CUT_NEG)
        for q in range(len(qw)):
            L.hho_cell_basis_grad(o._dp(bar), h, rd, qx[q], qy[q], o._dp(gx), o._dp(gy))
            g0 = float(np.dot(dofs[1:], gx[1:cbs]))
            g1 = float(np.dot(dofs[1:], gy[1:cbs]))
            s0 = math.pi * math.cos(math.pi * qx[q]) * math.sin(math.pi * qy[q])
            s1 = math.pi * math.sin(math.pi * qx[q]) * math.cos(math.pi * qy[q])
            H1 += qw[q] * ((s0 - g0) ** 2 + (s1 - g1) ** 2)
    return math.sqrt(H1), msh
