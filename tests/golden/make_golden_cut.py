#!/usr/bin/env python3
"""Generate tests/golden/cut_ops.npz: 50-digit (mpmath) evaluations of the CUT-cell operators of
apps/cuthho/cuthho_square.cpp for a handful of cut cells -- regular cuts and slivers, k = 0, 1, 2:

  * make_hho_laplacian(msh, cl, level_set, di, where)   cuthho_square.cpp:308-388
      full rbs x rbs system (the constant mode is kept), Nitsche terms
      -phi (grad phi.n)^T - (grad phi.n) phi^T + (eta / h_T) phi phi^T with h_T = the WHOLE cell's area and the
      level set's normal (:344-363), face terms on the `where` part of every face (:366-383);
  * make_hho_cut_stabilization                           :566-621  (h = whole-cell area, faces without points skipped);
  * cut make_rhs                                         :623-666  (interface term at degree k + 1, not 2(k + 1));
  * make_hho_laplacian_interface                         :390-502  (`data` only: it does not depend on the kernel
      component the reference's pivoted LDL^T leaves to rounding).

INPUTS that are held fixed (and stored in the file): the cell's four points and point ids, the level set, and the
quadrature lists (x, y, w) in double precision exactly as the C oracle's restatement of cuthho_geom.hpp:798-895
produces them (cut_cell_quadrature / cut_interface_quadrature / cut_face_quadrature) -- the same lists the product
returns from pa_cut_quadrature_points.  Everything downstream is evaluated here in 50-digit arithmetic, independently of
oracle/cut_truth.c (binary128, C) which the tests use to judge EVERY cut cell of a mesh and which this file pins.

Run:  python tests/golden/make_golden_cut.py      (about two minutes)
"""
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
import make_golden as G            # noqa: E402  (barycenter, diameter, measure, bases, solve)
import oracle_lib as O             # noqa: E402  (geometry + quadrature lists only)

mp.mp.dps = 50
mpf = mp.mpf
ETA = mpf(5)


def F(a):
    return [mpf(float(v)) for v in a]


def ls_normal(ls, x, y):
    gx, gy = 2 * x - 2 * mpf(ls.alpha), 2 * y - 2 * mpf(ls.beta)
    n = mp.sqrt(gx * gx + gy * gy)
    return gx / n, gy / n


def fn(which, x, y):
    s = mp.sin(mp.pi * x) * mp.sin(mp.pi * y)
    return 2 * mp.pi ** 2 * s if which == 1 else s


def normals(P):
    out = []
    for f in range(4):
        e = (P[(f + 1) % 4][0] - P[f][0], P[(f + 1) % 4][1] - P[f][1])
        nl = mp.sqrt(e[0] ** 2 + e[1] ** 2)
        out.append((e[1] / nl, -e[0] / nl))
    return out


def lists_of(ref, c, recdeg, facdeg, celdeg, where):
    """the double-precision quadrature lists of one side of a cut cell, as the oracle produces them"""
    d = {}
    d["cell"] = np.array(ref.cell_quadrature(c, 2 * recdeg, where)).T
    d["iface"] = np.array(ref.interface_quadrature(c, 2 * recdeg, where)).T
    d["faces"] = [np.array(ref.face_quadrature(c, lf, 2 * recdeg, where)).T for lf in range(4)]
    d["sfaces"] = [np.array(ref.face_quadrature(c, lf, 2 * facdeg, where)).T for lf in range(4)]
    d["rcell"] = np.array(ref.cell_quadrature(c, 2 * celdeg, where)).T
    d["riface"] = np.array(ref.interface_quadrature(c, celdeg, where)).T
    return d


def cut_laplacian(P, ids, ls, cd, fd, L):
    rd = fd + 1
    rbs, cbs, fbs = (rd + 2) * (rd + 1) // 2, (cd + 2) * (cd + 1) // 2, fd + 1
    ms = cbs + 4 * fbs
    bar, h, hT = G.barycenter(P), G.diameter(P), G.measure(P)
    stiff = mp.zeros(rbs, rbs)
    for (x, y, w) in L["cell"]:
        x, y, w = mpf(x), mpf(y), mpf(w)
        gx, gy = G.cell_dphi(bar, h, rd, x, y)
        for i in range(rbs):
            for j in range(rbs):
                stiff[i, j] += w * (gx[i] * gx[j] + gy[i] * gy[j])
    for (x, y, w) in L["iface"]:
        x, y, w = mpf(x), mpf(y), mpf(w)
        ph = G.cell_phi(bar, h, rd, x, y)
        gx, gy = G.cell_dphi(bar, h, rd, x, y)
        n = ls_normal(ls, x, y)
        dn = [gx[i] * n[0] + gy[i] * n[1] for i in range(rbs)]
        for i in range(rbs):
            for j in range(rbs):
                stiff[i, j] += w * (ph[i] * ph[j] * ETA / hT - ph[i] * dn[j] - dn[i] * ph[j])
    gr_rhs = mp.zeros(rbs, ms)
    gr_rhs[:, 0:cbs] = stiff[:, 0:cbs]
    nr = normals(P)
    for f in range(4):
        a, b = G.face_pts(P, ids, f)
        for (x, y, w) in L["faces"][f]:
            x, y, w = mpf(x), mpf(y), mpf(w)
            ph = G.cell_phi(bar, h, rd, x, y)
            gx, gy = G.cell_dphi(bar, h, rd, x, y)
            fp = G.face_phi(a, b, fd, x, y)
            for i in range(rbs):
                wdn = w * (gx[i] * nr[f][0] + gy[i] * nr[f][1])
                for j in range(fbs):
                    gr_rhs[i, cbs + f * fbs + j] += wdn * fp[j]
                for j in range(cbs):
                    gr_rhs[i, j] -= wdn * ph[j]
    oper = G.solve(stiff, gr_rhs)
    return oper, gr_rhs.T * oper


def cut_stabilization(P, ids, cd, fd, L):
    cbs, fbs = (cd + 2) * (cd + 1) // 2, fd + 1
    ms = cbs + 4 * fbs
    bar, hd, hT = G.barycenter(P), G.diameter(P), G.measure(P)
    data = mp.zeros(ms, ms)
    for f in range(4):
        if len(L["sfaces"][f]) == 0:
            continue
        a, b = G.face_pts(P, ids, f)
        mass, trace = mp.zeros(fbs, fbs), mp.zeros(fbs, cbs)
        for (x, y, w) in L["sfaces"][f]:
            x, y, w = mpf(x), mpf(y), mpf(w)
            ph = G.cell_phi(bar, hd, cd, x, y)
            fp = G.face_phi(a, b, fd, x, y)
            for i in range(fbs):
                for j in range(fbs):
                    mass[i, j] += w * fp[i] * fp[j]
                for j in range(cbs):
                    trace[i, j] += w * fp[i] * ph[j]
        op = mp.zeros(fbs, ms)
        op[:, 0:cbs] = G.solve(mass, trace)
        for i in range(fbs):
            op[i, cbs + f * fbs + i] = -1
        data += op.T * mass * op / hT
    return data


def cut_rhs(P, ls, cd, L):
    cbs = (cd + 2) * (cd + 1) // 2
    bar, h, hT = G.barycenter(P), G.diameter(P), G.measure(P)
    rhs = mp.zeros(cbs, 1)
    for (x, y, w) in L["rcell"]:
        x, y, w = mpf(x), mpf(y), mpf(w)
        ph = G.cell_phi(bar, h, cd, x, y)
        fv = fn(1, x, y)
        for i in range(cbs):
            rhs[i] += w * ph[i] * fv
    for (x, y, w) in L["riface"]:
        x, y, w = mpf(x), mpf(y), mpf(w)
        ph = G.cell_phi(bar, h, cd, x, y)
        gx, gy = G.cell_dphi(bar, h, cd, x, y)
        n = ls_normal(ls, x, y)
        bv = fn(2, x, y)
        for i in range(cbs):
            rhs[i] += w * bv * (ph[i] * ETA / hT - (gx[i] * n[0] + gy[i] * n[1]))
    return rhs


def interface_data(P, ids, ls, cd, fd, Ln, Lp, kappa):
    """data of make_hho_laplacian_interface (unknown order [cell-, cell+, faces-, faces+]); the constant of the negative
    side is pinned: data does not depend on the kernel component (gr_rhs is orthogonal to e_0 + e_rbs)."""
    rd = fd + 1
    rbs, cbs, fbs = (rd + 2) * (rd + 1) // 2, (cd + 2) * (cd + 1) // 2, fd + 1
    ms = cbs + 4 * fbs
    n2, m2 = 2 * rbs, 2 * ms
    bar, h, hT = G.barycenter(P), G.diameter(P), G.measure(P)
    k = [mpf(kappa[0]), mpf(kappa[1])]
    stiff = mp.zeros(n2, n2)
    for side, L in enumerate((Ln, Lp)):
        o = side * rbs
        for (x, y, w) in L["cell"]:
            x, y, w = mpf(x), mpf(y), mpf(w)
            gx, gy = G.cell_dphi(bar, h, rd, x, y)
            for i in range(rbs):
                for j in range(rbs):
                    stiff[o + i, o + j] += k[side] * w * (gx[i] * gx[j] + gy[i] * gy[j])
    for (x, y, w) in Ln["iface"]:
        x, y, w = mpf(x), mpf(y), mpf(w)
        ph = G.cell_phi(bar, h, rd, x, y)
        gx, gy = G.cell_dphi(bar, h, rd, x, y)
        n = ls_normal(ls, x, y)
        dn = [gx[i] * n[0] + gy[i] * n[1] for i in range(rbs)]
        for i in range(rbs):
            for j in range(rbs):
                a = k[0] * w * ph[i] * dn[j]
                b = k[0] * w * dn[i] * ph[j]
                c = k[0] * w * ph[i] * ph[j] * ETA / hT
                stiff[i, j] += c - a - b
                stiff[rbs + i, j] += a - c
                stiff[i, rbs + j] += b - c
                stiff[rbs + i, rbs + j] += c
    gr_rhs = mp.zeros(n2, m2)
    gr_rhs[:, 0:cbs] = stiff[:, 0:cbs]
    gr_rhs[:, cbs:2 * cbs] = stiff[:, rbs:rbs + cbs]
    nr = normals(P)
    for f in range(4):
        a, b = G.face_pts(P, ids, f)
        for side, L in enumerate((Ln, Lp)):
            ro, cc, cf = side * rbs, side * cbs, 2 * cbs + side * 4 * fbs + f * fbs
            for (x, y, w) in L["faces"][f]:
                x, y, w = mpf(x), mpf(y), mpf(w)
                ph = G.cell_phi(bar, h, rd, x, y)
                gx, gy = G.cell_dphi(bar, h, rd, x, y)
                fp = G.face_phi(a, b, fd, x, y)
                for i in range(rbs):
                    wdn = k[side] * w * (gx[i] * nr[f][0] + gy[i] * nr[f][1])
                    for j in range(cbs):
                        gr_rhs[ro + i, cc + j] -= wdn * ph[j]
                    for j in range(fbs):
                        gr_rhs[ro + i, cf + j] += wdn * fp[j]
    X = G.solve(stiff[1:, 1:], gr_rhs[1:, :])
    return gr_rhs[1:, :].T * X


# (mesh size N, refinement steps, cell, degrees k, with the interface problem?)  Cells: 123 of the 20 x 20 mesh is its
# worst sliver (1-norm condition number of the k = 2 reconstruction system 7.9e6), 328 its median cut (4.6e3); 39176 of the
# 512 x 512 mesh is the worst sliver of BASELINE.json's config 3 (1.9e9), 219343 its median cut (7.9e5).
CASES = [(20, 4, 123, (0, 1, 2), True), (20, 4, 328, (0, 1, 2), True), (512, 4, 39176, (2,), False), (512, 4, 219343, (2,), False)]
KAPPA = (1.0, 7.5)


def main():
    out = {}
    meshes = {}
    for (N, r, c, ks, iface) in CASES:
        ref = meshes.setdefault((N, r), O.CutMesh(N, refsteps=r))
        assert ref.cell_loc[c] == O.CUT_ON_INTERFACE, (N, c)
        pts = ref.points[ref.ptids[c].astype(np.int64)]
        ids = tuple(int(i) for i in ref.ptids[c])
        P = [(mpf(float(pts[i, 0])), mpf(float(pts[i, 1]))) for i in range(4)]
        for k in ks:
            cd, fd = k + 1, k
            tag = f"{N}|{r}|{c}|{k}"
            Ln = lists_of(ref, c, fd + 1, fd, cd, O.CUT_NEG)
            oper, data = cut_laplacian(P, ids, ref.ls, cd, fd, Ln)
            stab = cut_stabilization(P, ids, cd, fd, Ln)
            rhs = cut_rhs(P, ref.ls, cd, Ln)
            out[tag + "|pts"] = pts
            out[tag + "|ids"] = np.array(ids, dtype=np.int64)
            for name, v in Ln.items():
                if isinstance(v, list):
                    for lf, a in enumerate(v):
                        out[f"{tag}|q_{name}{lf}"] = a.reshape(-1, 3)
                else:
                    out[f"{tag}|q_{name}"] = v.reshape(-1, 3)
            out[tag + "|oper"], out[tag + "|data"] = G.to_np(oper), G.to_np(data)
            out[tag + "|stab"], out[tag + "|rhs"] = G.to_np(stab), G.to_np(rhs)
            if iface:
                Lp = lists_of(ref, c, fd + 1, fd, cd, O.CUT_POS)
                out[tag + "|idata"] = G.to_np(interface_data(P, ids, ref.ls, cd, fd, Ln, Lp, KAPPA))
            print(tag, "points:", len(Ln["cell"]), len(Ln["iface"]), flush=True)
    path = os.path.join(HERE, "cut_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
