"""CPU tests of the oracle itself: algebraic properties that are exact consequences of
hho.hpp:32-237, quadrature/basis known answers, mesh index maps."""
import ctypes as C
import math

import numpy as np
import pytest

from cases import CELLS, DEGREES


def _basis_fn(oracle, pts, rd, j):
    L = oracle.lib()
    p = np.ascontiguousarray(pts.reshape(8))
    bar = np.zeros(2)
    L.hho_cell_barycenter(oracle._dp(p), oracle._dp(bar))
    h = L.hho_cell_diameter(oracle._dp(p))
    buf = np.zeros(32)

    def f(x, y):
        L.hho_cell_basis_eval(oracle._dp(bar), h, rd, x, y, oracle._dp(buf))
        return buf[j]
    return f


def test_degree_info(oracle):
    d = oracle.degrees(3, 2)
    assert (d.cell_deg, d.face_deg, d.rec_deg) == (3, 2, 3)
    d = oracle.degrees(0, 1)
    assert (d.cell_deg, d.face_deg, d.rec_deg) == (0, 1, 2)
    fb = C.c_int(0)
    d = oracle.lib().hho_degree_info2(5, 1, C.byref(fb))     # invalid -> equal order (utils.hpp:88-91)
    assert fb.value == 1 and (d.cell_deg, d.face_deg, d.rec_deg) == (1, 1, 2)
    d = oracle.lib().hho_degree_info2(1, 0, C.byref(fb))
    assert fb.value == 0 and (d.cell_deg, d.face_deg, d.rec_deg) == (1, 0, 1)


def test_iexp_pow(oracle):
    L = oracle.lib()
    for x in (0.3, -1.7, 2.0):
        for n in range(0, 9):
            assert L.hho_iexp_pow(x, n) == pytest.approx(x ** n, rel=1e-15)
    assert L.hho_iexp_pow(0.0, 0) == 1.0


@pytest.mark.parametrize("deg", range(0, 10))
def test_gauss_legendre_exactness(oracle, deg):
    nd, wt = np.zeros(5), np.zeros(5)
    n = oracle.lib().hho_gauss_legendre(deg, oracle._dp(nd), oracle._dp(wt))
    assert n == ((deg | 1) + 1) // 2
    for p in range(0, 2 * n):
        exact = 0.0 if p % 2 else 2.0 / (p + 1)
        assert np.dot(wt[:n], nd[:n] ** p) == pytest.approx(exact, abs=1e-15)
    if n == 3:   # emission order matters for summation order (quadratures.hpp:111-119)
        assert nd[0] < 0 < nd[1] and nd[2] == 0.0
    if n == 5:
        assert nd[0] == 0.0


def test_dunavant_off_by_one_and_hole(oracle):
    L = oracle.lib()
    p0, p1, p2 = np.array([0.0, 0.0]), np.array([1.0, 0.0]), np.array([0.0, 1.0])
    qx, qy, qw = np.zeros(16), np.zeros(16), np.zeros(16)
    counts = {}
    for deg in range(0, 9):
        counts[deg] = L.hho_triangle_quadrature(oracle._dp(p0), oracle._dp(p1), oracle._dp(p2), deg,
                                                oracle._dp(qx), oracle._dp(qy), oracle._dp(qw))
    # rules[deg] is rule_{deg+1}; degree 8 hits the {0,NULL} sentinel
    assert counts == {0: 3, 1: 3, 2: 4, 3: 6, 4: 7, 5: 12, 6: 13, 7: 16, 8: 0}
    assert L.hho_triangle_quadrature(oracle._dp(p0), oracle._dp(p1), oracle._dp(p2), 9,
                                     oracle._dp(qx), oracle._dp(qy), oracle._dp(qw)) < 0
    # exactness of the rule actually selected for deg=4 (rule_5, degree 5): int x^2 y^3 over the unit triangle
    n = L.hho_triangle_quadrature(oracle._dp(p0), oracle._dp(p1), oracle._dp(p2), 4,
                                  oracle._dp(qx), oracle._dp(qy), oracle._dp(qw))
    val = np.sum(qw[:n] * qx[:n] ** 2 * qy[:n] ** 3)
    assert val == pytest.approx(math.factorial(2) * math.factorial(3) / math.factorial(7), rel=1e-13)


@pytest.mark.parametrize("name", list(CELLS))
def test_geometry(oracle, name):
    pts, ids = CELLS[name]
    L = oracle.lib()
    p = np.ascontiguousarray(pts.reshape(8))
    x, y = pts[:, 0], pts[:, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    assert L.hho_cell_measure(oracle._dp(p)) == pytest.approx(area, rel=1e-14)
    cx = np.sum((x + np.roll(x, -1)) * (x * np.roll(y, -1) - np.roll(x, -1) * y)) / (6 * area)
    cy = np.sum((y + np.roll(y, -1)) * (x * np.roll(y, -1) - np.roll(x, -1) * y)) / (6 * area)
    bar = np.zeros(2)
    L.hho_cell_barycenter(oracle._dp(p), oracle._dp(bar))
    assert bar == pytest.approx([cx, cy], rel=1e-13, abs=1e-14)
    d = max(np.linalg.norm(pts[i] - pts[j]) for i in range(4) for j in range(i + 1, 4))
    assert L.hho_cell_diameter(oracle._dp(p)) == pytest.approx(d, rel=1e-15)
    n = np.zeros(8)
    L.hho_cell_normals(oracle._dp(p), oracle._dp(n))
    n = n.reshape(4, 2)
    for i in range(4):
        e = pts[(i + 1) % 4] - pts[i]
        assert abs(np.dot(n[i], e)) < 1e-15 and np.linalg.norm(n[i]) == pytest.approx(1.0)
        assert np.dot(n[i], 0.5 * (pts[i] + pts[(i + 1) % 4]) - bar) > 0      # outward
    # quadrature integrates the area (both kinds)
    qx, qy, qw = np.zeros(64), np.zeros(64), np.zeros(64)
    for kind, deg in ((oracle.QUAD_TENSOR, 4), (oracle.QUAD_FAN, 4), (oracle.QUAD_FAN, 6)):
        nq = L.hho_cell_quadrature(oracle._dp(p), kind, deg, oracle._dp(qx), oracle._dp(qy), oracle._dp(qw))
        assert nq > 0
        assert np.sum(qw[:nq]) == pytest.approx(area, rel=1e-13)


@pytest.mark.parametrize("name", list(CELLS))
@pytest.mark.parametrize("cd,fd", [d for d in DEGREES if d != (1, 2)])
def test_polynomial_consistency(oracle, name, cd, fd):
    """oper . I_T(p) == coefficients of p (minus the constant) for p in P^{recdeg} (hho.hpp:63-92)."""
    pts, ids = CELLS[name]
    di = oracle.degrees(cd, fd)
    st, oper, data = oracle.make_laplacian(pts, ids, di)
    assert st == 0
    assert np.allclose(data, data.T, atol=1e-13 * np.abs(data).max())
    assert np.linalg.eigvalsh(0.5 * (data + data.T)).min() > -1e-12 * np.abs(data).max()
    rd = di.rec_deg
    dinc = max(0, (rd - cd + 1) // 2)
    for j in range(di.rbs):
        st, Ip = oracle.project_function(pts, ids, di, _basis_fn(oracle, pts, rd, j), dinc=dinc)
        assert st == 0
        got = oper @ Ip
        want = np.zeros(di.rbs - 1)
        if j > 0:
            want[j - 1] = 1.0
        assert np.abs(got - want).max() < 5e-11, (j, np.abs(got - want).max())
        if j == 0:      # constants are in the kernel of the stiffness part
            assert np.abs(data @ Ip).max() < 1e-11 * max(1.0, np.abs(data).max())


@pytest.mark.parametrize("name", list(CELLS))
@pytest.mark.parametrize("cd,fd", [d for d in DEGREES if d != (1, 2)])
def test_stabilization_kernels(oracle, name, cd, fd):
    pts, ids = CELLS[name]
    di = oracle.degrees(cd, fd)
    st, oper, data = oracle.make_laplacian(pts, ids, di)
    st, fancy = oracle.make_fancy_stabilization(pts, ids, di, oper)
    assert st == 0
    st, naive = oracle.make_naive_stabilization(pts, ids, di)
    assert st == 0
    for S in (fancy, naive):
        assert np.allclose(S, S.T, atol=1e-12 * np.abs(S).max())
        assert np.linalg.eigvalsh(0.5 * (S + S.T)).min() > -1e-11 * np.abs(S).max()
    rd = di.rec_deg
    dinc = max(0, (rd - cd + 1) // 2)
    scale = np.abs(fancy).max()
    for j in range(di.rbs):      # fancy annihilates I_T(p) for p in P^{k+1} (stabilization_test.cpp)
        st, Ip = oracle.project_function(pts, ids, di, _basis_fn(oracle, pts, rd, j), dinc=dinc)
        assert np.abs(fancy @ Ip).max() < 2e-10 * scale, j
    scale = np.abs(naive).max()
    for j in range(di.cbs):      # naive annihilates I_T(p) for p in P^{celdeg}
        st, Ip = oracle.project_function(pts, ids, di, _basis_fn(oracle, pts, rd, j), dinc=dinc)
        assert np.abs(naive @ Ip).max() < 2e-10 * scale, j


def test_face_basis_orientation(oracle):
    """Odd face modes flip sign with the id order of the endpoints (bases.hpp:260-272)."""
    pts, _ = CELLS["square"]
    di = oracle.degrees(2, 1)
    _, operA, _ = oracle.make_laplacian(pts, (6, 7, 12, 11), di)
    _, operB, _ = oracle.make_laplacian(pts, (7, 6, 12, 11), di)     # bottom face now runs p1->p0
    cbs, fbs = di.cbs, di.fbs
    A, B = operA[:, cbs:cbs + fbs], operB[:, cbs:cbs + fbs]
    assert np.allclose(A[:, 0], B[:, 0], atol=1e-14) and np.allclose(A[:, 1], -B[:, 1], atol=1e-14)
    assert np.abs(A[:, 1]).max() > 1e-3
    assert np.allclose(operA[:, cbs + fbs:], operB[:, cbs + fbs:], atol=1e-14)


def test_fan_degree8_hole_is_detected(oracle):
    pts, ids = CELLS["square"]
    di = oracle.degrees(4, 3)        # 2*recdeg = 8 -> zero points -> singular gr_lhs
    st, oper, data = oracle.make_laplacian(pts, ids, di, quad=oracle.QUAD_FAN)
    assert st == 3                   # HHO_ERR_NOT_SPD where Eigen would silently give NaN


def test_static_condensation_identity(oracle):
    pts, ids = CELLS["distorted"]
    di = oracle.degrees(3, 2)
    _, oper, data = oracle.make_laplacian(pts, ids, di)
    _, stab = oracle.make_fancy_stabilization(pts, ids, di, oper)
    lc = data + stab
    rng = np.random.default_rng(0)
    f = rng.standard_normal(di.cbs)
    st, S, g, rec = oracle.static_condensation(lc, f, di.cbs)
    assert st == 0
    c = di.cbs
    ATT, ATF, AFT, AFF = lc[:c, :c], lc[:c, c:], lc[c:, :c], lc[c:, c:]
    assert np.allclose(S, AFF - AFT @ np.linalg.solve(ATT, ATF), rtol=0, atol=1e-12 * np.abs(lc).max())
    assert np.allclose(g, -AFT @ np.linalg.solve(ATT, f), atol=1e-12 * np.abs(lc).max())
    uF = rng.standard_normal(4 * di.fbs)
    uT = rec[:, 0] + rec[:, 1:] @ uF
    assert np.allclose(ATT @ uT + ATF @ uF, f, atol=1e-11 * np.abs(lc).max())


@pytest.mark.parametrize("Nx,Ny", [(1, 1), (2, 3), (4, 4), (7, 5)])
def test_mesh_closed_forms_match_literal_generator(oracle, Nx, Ny):
    """Face ids / boundary flags in closed form == sort+unique of basic_mesh.hpp:266-297."""
    mp, points, ptids = oracle.make_mesh(Nx, Ny)
    faces, bnd = oracle.mesh_faces(mp)
    L = oracle.lib()
    assert faces.shape[0] == Nx * (Ny + 1) + Ny * (Nx + 1)
    assert np.all(faces[:, 0] < faces[:, 1])
    key = faces[:, 0].astype(np.int64) * (1 << 32) + faces[:, 1].astype(np.int64)
    assert np.all(np.diff(key) > 0)
    for j in range(Ny):
        for i in range(Nx):
            c = j * Nx + i
            ids = ptids[c]
            assert list(ids) == [j * (Nx + 1) + i, j * (Nx + 1) + i + 1, (j + 1) * (Nx + 1) + i + 1, (j + 1) * (Nx + 1) + i]
            for lf in range(4):
                a, b = int(ids[lf]), int(ids[(lf + 1) % 4])
                lo, hi = min(a, b), max(a, b)
                fid = L.hho_mesh_face_id(C.byref(mp), i, j, lf)
                assert tuple(faces[fid]) == (lo, hi)
                assert bool(bnd[fid]) == bool(L.hho_mesh_face_is_boundary(C.byref(mp), i, j, lf))
    assert int(bnd.sum()) == 2 * (Nx + Ny)
    assert points[-1] == pytest.approx([1.0, 1.0])


def test_golub_welsch_restatement(oracle):
    """gauss_legendre() beyond five nodes = golub_welsch (quadratures.hpp:32-75): eigenvalues of the Jacobi matrix in
    ascending order, weights from the first eigenvector components -- the Gauss-Legendre rule, exact to degree 2n - 1."""
    L = oracle.lib()
    for degree in range(10, 16):
        n = ((degree | 1) + 1) // 2
        nd, wt = np.zeros(8), np.zeros(8)
        assert L.hho_gauss_legendre(degree, oracle._dp(nd), oracle._dp(wt)) == n
        x, w = np.polynomial.legendre.leggauss(n)
        assert np.all(np.diff(nd[:n]) > 0)
        assert np.abs(nd[:n] - x).max() < 5e-15 and np.abs(wt[:n] - w).max() < 5e-15
        for p in range(0, 2 * n, 2):                       # exactness on monomials
            assert abs((wt[:n] * nd[:n] ** p).sum() - 2.0 / (p + 1)) < 1e-14
    nd, wt = np.zeros(8), np.zeros(8)
    assert L.hho_gauss_legendre(16, oracle._dp(nd), oracle._dp(wt)) < 0      # 9 nodes: beyond the tables of this restatement
