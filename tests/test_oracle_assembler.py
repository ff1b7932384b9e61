"""The oracle's restatement of assembler<Mesh> (hho.hpp:252-463): index maps, Dirichlet
elimination, and -- end to end -- the convergence orders convergence_test.cpp:312-325 prints
(k+2 in L2 for the (k+1, k) pairing, the only acceptance criterion the reference has for config 1)."""
import math

import numpy as np
import pytest

import poisson_driver as pd


def test_index_maps_and_symmetry(oracle):
    N, cd, fd = 4, 2, 1
    LHS, RHS, asm, di = pd.oracle_assembly(N, cd, fd)
    nc, nf = N * N, 2 * N * (N + 1)
    assert asm.num_other == nf - 4 * N
    assert asm.system_size == di.cbs * nc + di.fbs * (nf - 4 * N)           # hho.hpp:331
    # compress table: running count over non-Dirichlet faces (hho.hpp:313-323)
    other = np.nonzero(asm.is_dir == 0)[0]
    assert list(asm.compress[other]) == list(range(len(other)))
    A = LHS.toarray()
    assert np.allclose(A, A.T, atol=1e-12 * np.abs(A).max())
    assert np.linalg.eigvalsh(0.5 * (A + A.T)).min() > 0                    # SPD after Dirichlet elimination
    # an interior cell assembles msize^2 triplets, a corner cell drops two faces
    mp, points, ptids = oracle.make_mesh(N, N)
    ms = di.msize
    lc = np.eye(ms)
    tr, tc, tv, rr, rv = asm.assemble_cell(5, lc, np.zeros(di.cbs))
    assert len(tr) == ms * ms
    tr, tc, tv, rr, rv = asm.assemble_cell(0, lc, np.zeros(di.cbs))
    assert len(tr) == (ms - 2 * di.fbs) ** 2 and (rr < 0).sum() == 2 * di.fbs
    # push order: local row-major over assembled pairs, cell dofs first at cell_offset*cbs (hho.hpp:362-366)
    assert list(tr[:3]) == [0, 0, 0] and list(tc[:di.cbs]) == list(range(di.cbs))


@pytest.mark.parametrize("cd,fd", [(1, 0), (2, 1), (3, 2)])
def test_convergence_orders(cd, fd):
    errs = []
    for N in (4, 8, 16):
        LHS, RHS, asm, di = pd.oracle_assembly(N, cd, fd)
        sol = pd.solve(LHS, RHS)
        errs.append(pd.l2_error(asm, di, sol))
    rates = [math.log2(errs[i] / errs[i + 1]) for i in range(2)]
    k = fd
    assert rates[-1] > k + 2 - 0.35, (errs, rates)                          # L2 order k+2
    assert errs[-1] < errs[0]
