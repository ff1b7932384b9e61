"""cutHHO on the GPU (config 3 of BASELINE.json, `cuthho_square -f`): host preprocessing of the
product (proton_amd/csrc/cut_host.hpp) and the cut-cell kernel (cut_device.hpp) through the C ABI,
against the oracle's restatement cell by cell, and end to end against apps/cuthho/cuthho.xlsx."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def asm():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from proton_amd.batch import BatchAssembler
    return BatchAssembler(0)


def nerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("N,k,r", [(10, 0, 4), (10, 1, 4), (20, 2, 4), (16, 1, 2), (12, 2, 5)])
def test_cut_operators_match_oracle(asm, oracle, N, k, r):
    from proton_amd.batch import to_rowcol
    ncut = asm.cut_preprocess(N, refsteps=r)
    ref = oracle.CutMesh(N, refsteps=r)
    # tags of the product's own preprocessing == the oracle's (bit-exact classification)
    assert np.array_equal(asm.cell_loc, ref.cell_loc)
    cut_cells = np.nonzero(ref.cell_loc == oracle.CUT_ON_INTERFACE)[0]
    assert ncut == len(cut_cells) and ncut > 0
    assert np.array_equal(np.nonzero(asm.cut_index >= 0)[0], cut_cells)
    out = asm.cut_local_ops(k)
    asm.synchronize()
    assert int(out["info"].abs().max().cpu()) == 0
    di = oracle.degrees(k + 1, k)
    oper, data, stab, lc, rhs = (to_rowcol(out["oper"]), to_rowcol(out["data"]), to_rowcol(out["stab"]),
                                 to_rowcol(out["lc"]), out["rhs"].cpu().numpy())
    errs = []
    for i, c in enumerate(cut_cells):
        st, o_oper, o_data = ref.laplacian(int(c), di)
        assert st == 0 and o_oper.shape == oper[i].shape
        st, o_stab = ref.cut_stabilization(int(c), di)
        st, o_rhs = ref.rhs(int(c), di.cell_deg)
        # The Nitsche-penalised rbs x rbs system of a sliver cut is badly conditioned (the cut part of
        # the cell can be a tiny fraction of it): both sides carry cond * eps, so the bound is looser
        # than for regular cells; the median over the cut cells must still be at rounding level.
        errs.append(max(nerr(oper[i], o_oper), nerr(data[i], o_data), nerr(lc[i], o_data + o_stab)))
        assert errs[-1] < 5e-9, (int(c), errs[-1])
        assert nerr(stab[i], o_stab) < TOL
        assert np.abs(rhs[i] - o_rhs).max() < 1e-12 * max(1.0, np.abs(o_rhs).max())
    assert np.median(errs) < 1e-11, np.median(errs)


@pytest.mark.parametrize("N,k", [(10, 0), (20, 1), (20, 2)])
def test_fictitious_domain_end_to_end_matches_xlsx(asm, oracle, N, k):
    """cuthho_square -k K -M N -N N -r 4 -f with the GPU's operators (uncut: fan + naive kernel; cut:
    cut kernel) reproduces the energy errors of apps/cuthho/cuthho.xlsx."""
    import cuthho_driver as cd
    from proton_amd.batch import to_rowcol
    FD = {(0, 10): 0.188501, (1, 20): 3.08508e-3, (2, 20): 9.30124e-5}

    def gpu_provider(msh, di):
        asm.cut_preprocess(N, refsteps=4)
        lc, rhs = asm.fictdom_local_ops(k)
        asm.synchronize()
        L, R = to_rowcol(lc), rhs.cpu().numpy()
        return [(L[c], R[c]) for c in range(msh.nc)]

    err, msh = cd.run_fictdom(N, k, 4, provider=gpu_provider)
    assert abs(err - FD[(k, N)]) / FD[(k, N)] < 6e-6


def test_cut_error_codes(asm):
    import ctypes as C
    import proton_amd as pa
    L = pa.capi.lib()
    ls = pa.capi.LevelSet(0, 0.35, 0.5, 0.5, 0.0)
    asm.cut_preprocess(10)
    # k = 3: 2*recdeg = 8 hits the empty Dunavant rule (cuthho_square -k 3 is broken in the reference)
    assert L.pa_cut_local_ops_batch(asm.ctx.h, 3, C.byref(ls), 0, 1, 2, None, None, None, None, None, None) == 3
    assert L.pa_cut_local_ops_batch(asm.ctx.h, 1, C.byref(ls), 2, 1, 2, None, None, None, None, None, None) == 1
    # a radius that makes the interface graze a node: the reference throws, the ABI reports
    bad = pa.capi.LevelSet(0, 2.0, 0.5, 0.5, 0.0)          # circle outside the unit square: no cut cells
    assert L.pa_cut_preprocess(asm.ctx.h, 8, 8, 0.0, 1.0, 0.0, 1.0, C.byref(bad), 4) == 0
    n, loc, idx = asm.ctx.cut_query()
    assert n == 0 and np.all(loc == 0)                     # everything inside (negative side)
