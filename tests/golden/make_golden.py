#!/usr/bin/env python3
"""Generate tests/golden/local_ops.npz: 50-digit (mpmath) evaluations of the HHO local
operators, independent of the C oracle (different language, different linear algebra,
exact-to-50-digits arithmetic) but implementing the same formulas *including the
reference's quirks*:

  * scaled monomials centred at the polygon centroid, scaled by the cell diameter
    (bases.hpp:85-133), face coordinate 4(base.t)/h_F^2 from the lower-id endpoint
    (bases.hpp:253-280);
  * tensor Gauss on the bilinear map with |det J| (quadratures.hpp:311-375), or the
    4-triangle fan with Dunavant rules[deg] == rule_{deg+1} and 15-digit constants
    (quadratures.hpp:238-271,377-402; quadratures_dunavant.hpp);
  * gradient reconstruction hho.hpp:32-96, naive stabilization with h = cell AREA
    hho.hpp:99-148, fancy stabilization with h = cell DIAMETER hho.hpp:155-237;
  * cell rhs for f = 2 pi^2 sin(pi x) sin(pi y) (convergence_test.cpp:100-102) at
    quadrature degree 2*celdeg (utils.hpp:153-174);
  * static condensation of the cell block (new functionality).

The reference itself ships no fixtures for local matrices, so these vectors are the
"truth" both the oracle and the HIP kernels are compared against (to cond*eps).
Run:  python tests/golden/make_golden.py      (about a minute)
"""
import os
import sys

import mpmath as mp
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cases import CELLS, DEGREES  # noqa: E402

mp.mp.dps = 50
M = mp.matrix
mpf = mp.mpf


def gauss(degree):
    d = degree | 1
    n = (d + 1) // 2
    s = mp.sqrt
    if n == 1:
        return [(mpf(0), mpf(2))]
    if n == 2:
        q = 1 / s(3)
        return [(-q, mpf(1)), (q, mpf(1))]
    if n == 3:
        q = s(mpf(3) / 5)
        return [(-q, mpf(5) / 9), (q, mpf(5) / 9), (mpf(0), mpf(8) / 9)]
    if n == 4:
        a1, a2 = mpf(3) / 7, 2 * s(mpf(6) / 5) / 7
        q1, w1 = s(a1 - a2), (18 + s(30)) / 36
        q2, w2 = s(a1 + a2), (18 - s(30)) / 36
        return [(-q1, w1), (q1, w1), (-q2, w2), (q2, w2)]
    if n == 5:
        a2 = 2 * s(mpf(10) / 7)
        q1, w1 = s(5 - a2) / 3, (322 + 13 * s(70)) / 900
        q2, w2 = s(5 + a2) / 3, (322 - 13 * s(70)) / 900
        return [(mpf(0), mpf(128) / 225), (-q1, w1), (q1, w1), (-q2, w2), (q2, w2)]
    raise ValueError("needs golub_welsch")


# Dunavant rules as (kind, a, b, c, w) orbits; constants are the reference's 15-digit decimals,
# converted to the nearest double first (that is what the compiled reference holds).
_D = lambda s: mpf(float(s))  # noqa: E731
DUNAVANT = {
    1: [(1, "0.333333333333333", 0, 0, "1.000000000000000")],
    2: [(3, "0.666666666666667", "0.166666666666667", 0, "0.333333333333333")],
    3: [(1, "0.333333333333333", 0, 0, "-0.562500000000000"),
        (3, "0.600000000000000", "0.200000000000000", 0, "0.520833333333333")],
    4: [(3, "0.108103018168070", "0.445948490915965", 0, "0.223381589678011"),
        (3, "0.816847572980459", "0.091576213509771", 0, "0.109951743655322")],
    5: [(1, "0.333333333333333", 0, 0, "0.225000000000000"),
        (3, "0.059715871789770", "0.470142064105115", 0, "0.132394152788506"),
        (3, "0.797426985353087", "0.101286507323456", 0, "0.125939180544827")],
    6: [(3, "0.501426509658179", "0.249286745170910", 0, "0.116786275726379"),
        (3, "0.873821971016996", "0.063089014491502", 0, "0.050844906370207"),
        (6, "0.053145049844817", "0.310352451033784", "0.636502499121399", "0.082851075618374")],
    7: [(1, "0.333333333333333", 0, 0, "-0.149570044467682"),
        (3, "0.479308067841920", "0.260345966079040", 0, "0.175615257433208"),
        (3, "0.869739794195568", "0.065130102902216", 0, "0.053347235608838"),
        (6, "0.048690315425316", "0.312865496004874", "0.638444188569810", "0.077113760890257")],
    8: [(1, "0.333333333333333", 0, 0, "0.144315607677787"),
        (3, "0.081414823414554", "0.459292588292723", 0, "0.095091634267285"),
        (3, "0.658861384496480", "0.170569307751760", 0, "0.103217370534718"),
        (3, "0.898905543365938", "0.050547228317031", 0, "0.032458497623198"),
        (6, "0.008394777409958", "0.263112829634638", "0.728492392955404", "0.027230314174435")],
}


def dunavant_rows(rule_no):
    rows = []
    for kind, a, b, c, w in DUNAVANT[rule_no]:
        a, b, c, w = _D(a), _D(b), _D(c), _D(w)
        if kind == 1:
            rows.append((a, a, a, w))
        elif kind == 3:
            rows += [(a, b, b, w), (b, a, b, w), (b, b, a, w)]
        else:
            rows += [(a, b, c, w), (a, c, b, w), (b, a, c, w), (b, c, a, w), (c, a, b, w), (c, b, a, w)]
    return rows


def barycenter(P):
    p0 = P[0]
    rx = ry = den = mpf(0)
    for i in range(2, 4):
        a = (P[i - 1][0] - p0[0], P[i - 1][1] - p0[1])
        b = (P[i][0] - p0[0], P[i][1] - p0[1])
        d = (a[0] * b[1] - a[1] * b[0]) / 2
        rx += (a[0] + b[0]) * d
        ry += (a[1] + b[1]) * d
        den += d
    return (p0[0] + rx / (3 * den), p0[1] + ry / (3 * den))


def dist(a, b):
    return mp.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)


def diameter(P):
    return max(dist(P[i], P[j]) for i in range(4) for j in range(i + 1, 4))


def measure(P):
    acc = mpf(0)
    for i in range(1, 3):
        u = (P[i][0] - P[0][0], P[i][1] - P[0][1])
        v = (P[i + 1][0] - P[0][0], P[i + 1][1] - P[0][1])
        acc += abs(u[0] * v[1] - u[1] * v[0]) / 2
    return acc


def cell_qps(P, kind, degree):
    out = []
    if kind == "tensor":
        g = gauss(degree)
        for (eta, wj) in g:
            for (xi, wi) in g:
                N = [(1 - xi) * (1 - eta) / 4, (1 + xi) * (1 - eta) / 4, (1 + xi) * (1 + eta) / 4, (1 - xi) * (1 + eta) / 4]
                x = sum(N[k] * P[k][0] for k in range(4))
                y = sum(N[k] * P[k][1] for k in range(4))
                j11 = ((P[1][0] - P[0][0]) * (1 - eta) + (P[2][0] - P[3][0]) * (1 + eta)) / 4
                j12 = ((P[1][1] - P[0][1]) * (1 - eta) + (P[2][1] - P[3][1]) * (1 + eta)) / 4
                j21 = ((P[3][0] - P[0][0]) * (1 - xi) + (P[2][0] - P[1][0]) * (1 + xi)) / 4
                j22 = ((P[3][1] - P[0][1]) * (1 - xi) + (P[2][1] - P[1][1]) * (1 + xi)) / 4
                out.append((x, y, wi * wj * abs(j11 * j22 - j12 * j21)))
        return out
    bar = barycenter(P)
    deg = max(degree, 1)
    if deg > 8:
        raise ValueError("Quadrature order too high")
    rows = [] if deg == 8 else dunavant_rows(deg + 1)       # rules[deg] == rule_{deg+1}; rules[8] empty
    for i in range(4):
        p0, p1, p2 = P[i], P[(i + 1) % 4], bar
        area = abs(((p1[0] - p0[0]) * (p2[1] - p0[1]) - (p1[1] - p0[1]) * (p2[0] - p0[0])) / 2)
        for (l0, l1, l2, w) in rows:
            out.append((p0[0] * l0 + p1[0] * l1 + p2[0] * l2, p0[1] * l0 + p1[1] * l1 + p2[1] * l2, area * w))
    return out


def face_qps(a, b, degree):
    L = dist(a, b)
    return [((1 - t) / 2 * a[0] + (1 + t) / 2 * b[0], (1 - t) / 2 * a[1] + (1 + t) / 2 * b[1], w * L / 2)
            for (t, w) in gauss(degree)]


def monomials(deg):
    return [(k - i, i) for k in range(deg + 1) for i in range(k + 1)]


def cell_phi(bar, h, deg, x, y):
    bx, by = (x - bar[0]) / (h / 2), (y - bar[1]) / (h / 2)
    return [bx ** px * by ** py for (px, py) in monomials(deg)]


def cell_dphi(bar, h, deg, x, y):
    bx, by = (x - bar[0]) / (h / 2), (y - bar[1]) / (h / 2)
    ih = 2 / h
    gx, gy = [], []
    for (px, py) in monomials(deg):
        gx.append(0 if px == 0 else px * ih * bx ** (px - 1) * by ** py)
        gy.append(0 if py == 0 else py * ih * bx ** px * by ** (py - 1))
    return gx, gy


def face_phi(a, b, deg, x, y):
    bar = ((a[0] + b[0]) / 2, (a[1] + b[1]) / 2)
    hF = dist(a, b)
    base = (bar[0] - a[0], bar[1] - a[1])
    ep = 4 * (base[0] * (x - bar[0]) + base[1] * (y - bar[1])) / hF ** 2
    return [ep ** i for i in range(deg + 1)]


def face_pts(P, ids, f):
    i0, i1 = f, (f + 1) % 4
    if ids[i0] > ids[i1]:
        i0, i1 = i1, i0
    return P[i0], P[i1]


def solve(A, B):
    # 50-digit arithmetic: the algorithm is irrelevant at double accuracy (cond <= 1e4)
    return mp.inverse(A) * B


def local_ops(P, ids, cd, fd, kind):
    rd = fd + 1
    rbs, cbs, fbs = (rd + 2) * (rd + 1) // 2, (cd + 2) * (cd + 1) // 2, fd + 1
    ms, nr = cbs + 4 * fbs, rbs - 1
    bar, h = barycenter(P), diameter(P)
    qps = cell_qps(P, kind, 2 * rd)
    stiff, mass = mp.zeros(rbs, rbs), mp.zeros(rbs, rbs)
    for (x, y, w) in qps:
        gx, gy = cell_dphi(bar, h, rd, x, y)
        ph = cell_phi(bar, h, rd, x, y)
        for i in range(rbs):
            for j in range(rbs):
                stiff[i, j] += w * (gx[i] * gx[j] + gy[i] * gy[j])
                mass[i, j] += w * ph[i] * ph[j]
    gr_lhs = stiff[1:, 1:]
    gr_rhs = mp.zeros(nr, ms)
    gr_rhs[:, 0:cbs] = stiff[1:, 0:cbs]
    fdata = []
    for f in range(4):
        a, b = face_pts(P, ids, f)
        e = (P[(f + 1) % 4][0] - P[f][0], P[(f + 1) % 4][1] - P[f][1])
        nl = mp.sqrt(e[0] ** 2 + e[1] ** 2)
        n = (e[1] / nl, -e[0] / nl)
        fm, ft = mp.zeros(fbs, fbs), mp.zeros(fbs, rbs)
        for (x, y, w) in face_qps(a, b, 2 * fd):
            ph = cell_phi(bar, h, rd, x, y)
            gx, gy = cell_dphi(bar, h, rd, x, y)
            fp = face_phi(a, b, fd, x, y)
            for i in range(nr):
                dn = gx[i + 1] * n[0] + gy[i + 1] * n[1]
                for j in range(fbs):
                    gr_rhs[i, cbs + f * fbs + j] += w * dn * fp[j]
                for j in range(cbs):
                    gr_rhs[i, j] -= w * dn * ph[j]
            for i in range(fbs):
                for j in range(fbs):
                    fm[i, j] += w * fp[i] * fp[j]
                for j in range(rbs):
                    ft[i, j] += w * fp[i] * ph[j]
        fdata.append((fm, ft))
    oper = solve(gr_lhs, gr_rhs)
    data = gr_rhs.T * oper

    # naive stabilization (h = area)
    area = measure(P)
    naive = mp.zeros(ms, ms)
    for f in range(4):
        fm, ft = fdata[f]
        op = mp.zeros(fbs, ms)
        op[:, 0:cbs] = solve(fm, ft[:, 0:cbs])
        for i in range(fbs):
            op[i, cbs + f * fbs + i] = -1
        naive += op.T * fm * op / area

    # fancy stabilization (h = diameter)
    M1, M2 = mass[0:cbs, 0:cbs], mass[0:cbs, 1:]
    proj1 = -solve(M1, M2 * oper)
    for i in range(cbs):
        proj1[i, i] += 1
    fancy = mp.zeros(ms, ms)
    for f in range(4):
        fm, ft = fdata[f]
        proj2 = solve(fm, ft[:, 1:] * oper)
        for i in range(fbs):
            proj2[i, cbs + f * fbs + i] -= 1
        proj3 = solve(fm, ft[:, 0:cbs] * proj1)
        B = proj2 + proj3
        fancy += B.T * fm * B / h

    # cell rhs (utils.hpp:153-174) with the convergence_test source term
    rhs = mp.zeros(cbs, 1)
    for (x, y, w) in cell_qps(P, kind, 2 * cd):
        ph = cell_phi(bar, h, cd, x, y)
        fv = 2 * mp.pi ** 2 * mp.sin(mp.pi * x) * mp.sin(mp.pi * y)
        for i in range(cbs):
            rhs[i] += w * ph[i] * fv

    # static condensation of lc = data + fancy
    lc = data + fancy
    ATT, ATF, AFT, AFF = lc[0:cbs, 0:cbs], lc[0:cbs, cbs:], lc[cbs:, 0:cbs], lc[cbs:, cbs:]
    X = solve(ATT, ATF)
    S = AFF - AFT * X
    g = -AFT * solve(ATT, rhs)
    return dict(oper=oper, data=data, naive=naive, fancy=fancy, rhs=rhs, S=S, g=g)


def to_np(A):
    return np.array([[float(A[i, j]) for j in range(A.cols)] for i in range(A.rows)], dtype=np.float64)


def main():
    out = {}
    for cname, (pts, ids) in CELLS.items():
        P = [(mpf(float(pts[i, 0])), mpf(float(pts[i, 1]))) for i in range(4)]
        for (cd, fd) in DEGREES:
            if (cd, fd) == (1, 2):
                continue
            for kind in ("tensor", "fan"):
                if kind == "fan" and (2 * (fd + 1) >= 8 or cname in ("corner", "thin")):
                    continue        # rules[8] is empty; keep the file small
                res = local_ops(P, ids, cd, fd, kind)
                for k, v in res.items():
                    out[f"{cname}|{cd}|{fd}|{kind}|{k}"] = to_np(v)
                print(cname, cd, fd, kind, flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "local_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
