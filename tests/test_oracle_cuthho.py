"""PIN of the cutHHO part of the oracle against the reference's committed numbers:
apps/cuthho/cuthho.xlsx sheet 1, "F.D." table (rows = interface refinement steps r, columns =
N = 10, 20, 40; energy-norm error printed by cuthho_square.cpp:1066).  Exercises the whole chain
of `cuthho_square -k K -M N -N N -r R -f`: level-set tagging, bisection zero crossings, node
displacement, interface refinement, cut quadrature (fan triangulation + Dunavant with the
rules[deg] off-by-one), cut Nitsche operators, cut stabilization, cut rhs, the generic assembler,
and must reproduce the 6 printed digits."""
import math

import numpy as np
import pytest

import cuthho_driver as cd

# cuthho.xlsx B7:D7, G7:I7, L7:N7 (r = 4) and row 8 (r = 5)
FD = {
    (0, 4): {10: 0.188501, 20: 9.66971e-2, 40: 4.84833e-2},
    (1, 4): {10: 1.1089e-2, 20: 3.08508e-3, 40: 7.58577e-4},
    (2, 4): {10: 7.28887e-4, 20: 9.30124e-5, 40: 1.17375e-5},
    (0, 5): {10: 0.188507, 20: 9.67103e-2},
    (1, 5): {10: 1.1089e-2, 20: 3.08498e-3},
    (2, 5): {10: 7.28742e-4, 20: 9.29638e-5},
}


@pytest.mark.parametrize("k,r,N", [(k, r, N) for (k, r), d in FD.items() for N in d if N <= 20] + [(1, 4, 40), (2, 4, 40)])
def test_fictitious_domain_energy_error_matches_xlsx(k, r, N):
    err, msh = cd.run_fictdom(N, k, r)
    ref = FD[(k, r)][N]
    assert abs(err - ref) / ref < 6e-6, (err, ref)          # 6 printed digits


# cuthho.xlsx, "Interface" table: B23:D23, G23:I23, L23:N23 (r = 4) and row 24 (r = 5)
INTERFACE = {
    (0, 4): {10: 0.285023, 20: 0.143641, 40: 7.17622e-2},
    (1, 4): {10: 2.01456e-2, 20: 5.22389e-3, 40: 1.33102e-3},
    (2, 4): {10: 1.13312e-3, 20: 1.38029e-4, 40: 1.71019e-5},
    (0, 5): {10: 0.285023, 20: 0.143641},
    (1, 5): {10: 2.01455e-2, 20: 5.22388e-3},
    (2, 5): {10: 1.13325e-3, 20: 1.38115e-4},
}


@pytest.mark.parametrize("k,r,N", [(k, r, N) for (k, r), d in INTERFACE.items() for N in d if N <= 20] + [(0, 4, 40), (2, 4, 40)])
def test_interface_problem_energy_error_matches_xlsx(k, r, N):
    """`cuthho_square -k K -M N -N N -r R -i` (run_cuthho_interface, cuthho_square.cpp:1625-1846):
    make_hho_laplacian_interface, both cut stabilizations, the one-sided right-hand sides and the
    interface_assembler with duplicated unknowns reproduce the 6 printed digits."""
    err, msh = cd.run_interface(N, k, r)
    ref = INTERFACE[(k, r)][N]
    assert abs(err - ref) / ref < 6e-6, (err, ref)


def test_interface_operator_properties(oracle):
    """2rbs x 2rbs reconstruction of a cut cell: data symmetric positive semi-definite, constants
    (the same on both sides) in its kernel, and kappa scaling of the one-sided blocks."""
    m = oracle.CutMesh(10, refsteps=4)
    di = oracle.degrees(2, 1)
    cbs, fbs, ms = di.cbs, di.fbs, di.msize
    u = np.zeros(2 * ms)
    u[0] = u[cbs] = 1.0
    u[2 * cbs::fbs] = 1.0
    for c in np.nonzero(m.cell_loc == oracle.CUT_ON_INTERFACE)[0]:
        st, oper, data = m.laplacian_interface(int(c), di)
        assert st == 0 and oper.shape == (2 * di.rbs, 2 * ms) and data.shape == (2 * ms, 2 * ms)
        scale = np.abs(data).max()
        assert np.abs(data - data.T).max() < 1e-11 * scale
        assert np.linalg.eigvalsh((data + data.T) / 2)[0] > -1e-9 * scale
        assert np.abs(data @ u).max() < 1e-9 * scale
    st, oper, data = m.laplacian_interface(0, di)
    assert st != 0                                                          # "The cell is not cut" (:397-398)


def test_cut_geometry_integrates_the_disc(oracle):
    """cuthho_square.cpp:681-732 (test_integration_domain): area and perimeter of the circle."""
    m = oracle.CutMesh(20, refsteps=4)
    area = per = 0.0
    ncut = 0
    for c in range(m.nc):
        if m.cell_loc[c] != oracle.CUT_POS:
            area += m.cell_quadrature(c, 2)[2].sum()
        if m.cell_loc[c] == oracle.CUT_ON_INTERFACE:
            ncut += 1
            per += m.interface_quadrature(c, 2)[2].sum()
            assert abs(m.cell_quadrature(c, 2)[2].sum() - oracle.lib().cut_cell_measure(m.h, c, oracle.CUT_NEG)) < 1e-15
    assert ncut > 0 and (m.cell_loc == oracle.CUT_NEG).sum() > 0
    assert abs(area - math.pi * 0.35 ** 2) < 2e-5 and abs(per - 2 * math.pi * 0.35) < 2e-5
    # every cut cell is crossed by exactly two cut faces; its interface polyline has 2^r + 1 points on the circle
    for c in np.nonzero(m.cell_loc == oracle.CUT_ON_INTERFACE)[0]:
        cut_faces = [f for f in m.cell_faces[c] if m.face_loc[int(f)] == oracle.CUT_ON_INTERFACE]
        assert len(cut_faces) == 2
        ifc = m.interface(int(c))
        assert ifc.shape == (17, 2)
        rad = np.hypot(ifc[:, 0] - 0.5, ifc[:, 1] - 0.5)
        assert np.abs(rad - 0.35).max() < 1e-4 * 0.1


def test_cut_operator_shapes_and_symmetry(oracle):
    m = oracle.CutMesh(10, refsteps=4)
    di = oracle.degrees(2, 1)
    cut = int(np.nonzero(m.cell_loc == oracle.CUT_ON_INTERFACE)[0][0])
    reg = int(np.nonzero(m.cell_loc == oracle.CUT_NEG)[0][0])
    st, oper, data = m.laplacian(cut, di)
    assert st == 0 and oper.shape == (di.rbs, di.msize)                     # cut cells keep the constant mode
    st, oper_r, data_r = m.laplacian(reg, di)
    assert st == 0 and oper_r.shape == (di.rbs - 1, di.msize)
    assert np.allclose(data, data.T, atol=1e-12 * np.abs(data).max())
    st, stab = m.cut_stabilization(cut, di)
    assert st == 0 and np.allclose(stab, stab.T, atol=1e-12 * np.abs(stab).max())
    # a cell in the positive side gets a zero right-hand side (cuthho_square.cpp:659-664)
    pos = int(np.nonzero(m.cell_loc == oracle.CUT_POS)[0][0])
    st, f = m.rhs(pos, di.cell_deg)
    assert st == 0 and np.all(f == 0.0)


@pytest.mark.parametrize("N", [10, 16, 33])
def test_agglomeration_branch_classification(oracle, N):
    """`-A` (cuthho_square.cpp:2039-2044): no node displacement, detect_cell_agglo_set (cuthho_geom.hpp:163-273).
    Every cut cell gets a class, uncut cells none; the classes inherit the symmetries of the circle on the
    uniform mesh; single-node cuts are re-derived here from the face intersection points."""
    m = oracle.CutMesh(N, agglomeration=True)
    plain = oracle.make_mesh(N, N)[1]
    assert np.array_equal(m.points, plain)                                   # nothing was displaced
    a = m.agglo_set()
    cut = m.cell_loc == oracle.CUT_ON_INTERFACE
    assert np.all(a[cut] > 0) and np.all(a[~cut] == 0) and cut.sum() > 0
    A = a.reshape(N, N)
    assert np.array_equal(A, A[::-1, :]) and np.array_equal(A, A[:, ::-1]) and np.array_equal(A, A.T)
    fip = np.ctypeslib.as_array(oracle.lib().cut_mesh_face_intersection(m.h), shape=(m.nf, 2))
    checked = 0
    for c in np.nonzero(cut)[0]:
        fcs = [int(f) for f in m.cell_faces[c]]
        cf = [m.face_loc[f] == oracle.CUT_ON_INTERFACE for f in fcs]
        pts = m.points[m.ptids[c].astype(np.int64)]
        for i in range(4):
            f1, f2, n = i, (i + 1) % 4, (i + 1) % 4
            if not (cf[f1] and cf[f2]):
                continue
            d = []
            for f in (fcs[f1], fcs[f2]):
                p0, p1 = m.points[int(m.faces[f, 0])], m.points[int(m.faces[f, 1])]
                d.append(np.linalg.norm(pts[n] - fip[f]) / np.linalg.norm(p1 - p0))
            node_neg = m.node_loc[int(m.ptids[c][n])] == oracle.CUT_NEG
            want = 1 if min(d) > 0.3 else (2 if node_neg else 3)
            assert a[c] == want
            checked += 1
    assert checked > 0


def test_neighbors_literal_search_is_the_eight_neighbourhood(oracle):
    """make_neighbors_info (cuthho_geom.hpp:343-370) restated literally (all pairs) == closed form"""
    N = 7
    m = oracle.CutMesh(N, agglomeration=True)
    nb = m.neighbors()
    for j in range(N):
        for i in range(N):
            want = sorted(jj * N + ii for jj in range(max(0, j - 1), min(N, j + 2)) for ii in range(max(0, i - 1), min(N, i + 2))
                          if (ii, jj) != (i, j))
            got = [int(v) for v in nb[j * N + i] if v >= 0]
            assert got == want
