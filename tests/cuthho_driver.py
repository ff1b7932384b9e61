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


def oracle_interface_provider(msh, di, kappa=(1.0, 1.0)):
    """-> list of (lc, rhs) per cell as run_cuthho_interface builds them (cuthho_square.cpp:1664-1716)."""
    cbs, nfd = di.cbs, 4 * di.fbs
    out = []
    for c in range(msh.nc):
        if msh.cell_loc[c] != o.CUT_ON_INTERFACE:
            k = kappa[0] if msh.cell_loc[c] == o.CUT_NEG else kappa[1]                  # :1672-1675
            st, oper, data = msh.laplacian(c, di)                                       # uncut: make_hho_laplacian
            assert st == 0
            st, stab = msh.cut_stabilization(c, di)                                     # uncut: naive stabilization (:1678)
            assert st == 0
            pts = msh.points[msh.ptids[c].astype(np.int64)]
            st, f = o.cell_rhs(pts, o.QUAD_FAN, di.cell_deg, 0, lambda x, y: 2 * math.pi ** 2 * math.sin(math.pi * x) * math.sin(math.pi * y))
            assert st == 0
            out.append((k * data + stab, f))
            continue
        st, oper, lc = msh.laplacian_interface(c, di, kappa[0], kappa[1])               # :1691-1692
        assert st == 0
        st, sn = msh.cut_stabilization(c, di, o.CUT_NEG)
        assert st == 0
        st, sp_ = msh.cut_stabilization(c, di, o.CUT_POS)
        assert st == 0
        sn, sp_ = kappa[0] * sn, kappa[1] * sp_                                         # :1694-1695
        cn, fn_ = slice(0, cbs), slice(2 * cbs, 2 * cbs + nfd)                          # :1697-1700
        cp, fp = slice(cbs, 2 * cbs), slice(2 * cbs + nfd, 2 * cbs + 2 * nfd)           # :1702-1705
        sc, sf = slice(0, cbs), slice(cbs, cbs + nfd)
        for (a, b), blk in (((cn, cn), (sc, sc)), ((cn, fn_), (sc, sf)), ((fn_, cn), (sf, sc)), ((fn_, fn_), (sf, sf))):
            lc[a, b] += sn[blk]
        for (a, b), blk in (((cp, cp), (sc, sc)), ((cp, fp), (sc, sf)), ((fp, cp), (sf, sc)), ((fp, fp), (sf, sf))):
            lc[a, b] += sp_[blk]
        st, f_n = msh.rhs_side(c, di.cell_deg, o.CUT_NEG)                               # :1710-1711
        assert st == 0
        st, f_p = msh.rhs_side(c, di.cell_deg, o.CUT_POS)
        assert st == 0
        out.append((lc, np.concatenate([f_n, f_p])))
    return out


def run_interface(N, k, refsteps=4, provider=oracle_interface_provider):
    """cuthho_square -k K -M N -N N -r R -i : -> (energy-norm error, mesh)   (cuthho_square.cpp:1625-1846)"""
    msh = o.CutMesh(N, refsteps=refsteps)
    di = o.degrees(k + 1, k)                                                            # :1662
    mp = o.MeshParams(N, N, 0.0, 1.0, 0.0, 1.0)
    plain = o.Assembler(mp, msh.points, msh.ptids, di, bf_id=2)                         # Dirichlet data of bcs_fun per face
    ct, ft, num_all_cells, num_other = msh.interface_tables()
    size = di.cbs * num_all_cells + di.fbs * num_other                                  # :1185
    local = provider(msh, di)
    rows, cols, vals = [], [], []
    RHS = np.zeros(size)
    cbs, fbs = di.cbs, di.fbs
    for c in range(msh.nc):
        lc, f = local[c]
        dd = np.zeros(lc.shape[0])
        if msh.cell_loc[c] != o.CUT_ON_INTERFACE:
            for lf in range(4):
                dd[cbs + lf * fbs: cbs + (lf + 1) * fbs] = plain.g[int(msh.cell_faces[c, lf])]
        tr, tc, tv, rr, rv = msh.interface_assemble(di, c, ct, ft, num_all_cells, lc, f, dd)
        rows.append(tr); cols.append(tc); vals.append(tv)
        ok = rr >= 0
        np.add.at(RHS, rr[ok], rv[ok])
    LHS = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(size, size))
    sol = spla.spsolve(LHS, RHS)             # the reference runs Jacobi-PCG here (:1737-1743)

    L = o.lib()                              # energy-norm error :1762-1833
    rd = di.rec_deg
    H1 = 0.0
    gx, gy, bar = np.zeros(32), np.zeros(32), np.zeros(2)
    for c in range(msh.nc):
        pts = np.ascontiguousarray(msh.points[msh.ptids[c].astype(np.int64)].reshape(8))
        L.hho_cell_barycenter(o._dp(pts), o._dp(bar))
        h = L.hho_cell_diameter(o._dp(pts))
        sides = (o.CUT_NEG, o.CUT_POS) if msh.cell_loc[c] == o.CUT_ON_INTERFACE else (int(msh.cell_loc[c]),)
        for where in sides:
            o0 = L.cut_interface_cell_offset(msh.h, di, c, o._i64p(ct), where)
            dofs = sol[o0:o0 + cbs]
            qx, qy, qw = msh.cell_quadrature(c, 2 * di.cell_deg, where)
            for q in range(len(qw)):
                L.hho_cell_basis_grad(o._dp(bar), h, rd, qx[q], qy[q], o._dp(gx), o._dp(gy))
                g0 = float(np.dot(dofs[1:], gx[1:cbs]))
                g1 = float(np.dot(dofs[1:], gy[1:cbs]))
                s0 = math.pi * math.cos(math.pi * qx[q]) * math.sin(math.pi * qy[q])
                s1 = math.pi * math.sin(math.pi * qx[q]) * math.cos(math.pi * qy[q])
                H1 += qw[q] * ((s0 - g0) ** 2 + (s1 - g1) ** 2)
    return math.sqrt(H1), msh
