"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against (a) the committed 50-digit golden fixtures, (b) the CPU oracle on seeded meshes, and
(c) size-independent properties at BASELINE.json's full sizes.

Tolerance: 1e-12 normwise per cell, max|A - A*| / max|A*| (north_star).
"""
import math
import os
import sys

import numpy as np
import pytest

from cases import CELLS

pytestmark = pytest.mark.gpu
TOL = 1e-12

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "local_ops.npz"))


def nerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def nerr_cells(a, b):
    """worst per-cell normwise error of [n, r, c] batches"""
    num = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
    den = np.abs(b).reshape(b.shape[0], -1).max(axis=1)
    return (num / np.maximum(den, 1e-300)).max()


def gold_cases():
    seen = []
    for key in GOLD.files:
        c, cd, fd, kind, what = key.split("|")
        t = (c, int(cd), int(fd), kind)
        if t not in seen:
            seen.append(t)
    return seen


@pytest.fixture(scope="module")
def asm():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from proton_amd.batch import BatchAssembler
    return BatchAssembler(0)


def single_cell_mesh(pts, ids):
    npts = max(ids) + 1
    points = np.zeros((npts, 2))
    # unused slots get distinct far-away coordinates
    points[:, 0] = 100.0 + np.arange(npts)
    for v in range(4):
        points[ids[v]] = pts[v]
    return points, np.array([ids], dtype=np.uint32)


@pytest.mark.parametrize("cname,cd,fd,kind", gold_cases())
def test_matches_golden(asm, cname, cd, fd, kind):
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    pts, ids = CELLS[cname]
    points, ptids = single_cell_mesh(pts, ids)
    asm.set_mesh(points, ptids)
    quad = pa.QUAD_TENSOR if kind == "tensor" else pa.QUAD_FAN
    g = lambda what: GOLD[f"{cname}|{cd}|{fd}|{kind}|{what}"]  # noqa: E731
    for stab, key in ((pa.STAB_FANCY, "fancy"), (pa.STAB_NAIVE, "naive")):
        out = asm.local_ops(cd, fd, quad, stab, want=("oper", "data", "stab", "lc", "info"))
        asm.synchronize()
        assert int(out["info"].cpu()[0]) == 0
        assert nerr(to_rowcol(out["oper"])[0], g("oper")) < TOL
        assert nerr(to_rowcol(out["data"])[0], g("data")) < TOL
        assert nerr(to_rowcol(out["stab"])[0], g(key)) < TOL
        assert nerr(to_rowcol(out["lc"])[0], g("data") + g(key)) < TOL
    # lc-only call (no oper requested) takes the forward-substitution-only path
    out = asm.local_ops(cd, fd, quad, pa.STAB_FANCY, want=("lc",))
    assert nerr(to_rowcol(out["lc"])[0], g("data") + g("fancy")) < TOL
    out = asm.local_ops(cd, fd, quad, pa.STAB_NONE, want=("lc",))
    assert nerr(to_rowcol(out["lc"])[0], g("data")) < TOL
    # right-hand side with the convergence_test source term
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, quad)
    assert nerr(rhs.cpu().numpy()[0], g("rhs")[:, 0]) < TOL
    # static condensation of lc = data + fancy
    if kind == "tensor":
        out = asm.local_ops(cd, fd, quad, pa.STAB_FANCY, want=("lc",))
        S, gg, rec, info = asm.static_condensation(cd, fd, out["lc"], rhs)
        assert int(info.cpu()[0]) == 0
        assert nerr(S.cpu().numpy()[0].T, g("S")) < 10 * TOL
        assert np.abs(gg.cpu().numpy()[0] - g("g")[:, 0]).max() < 10 * TOL * max(np.abs(g("g")).max(), np.abs(g("rhs")).max())
        # packed upper triangle (the multi-GPU exchange format) carries the same values
        import torch
        from proton_amd.partition import unpack_symmetric
        nf = S.shape[1]
        Sp = torch.empty((S.shape[0], nf * (nf + 1) // 2), dtype=torch.float64, device=S.device)
        g2 = torch.empty_like(gg)
        di_, _ = pa.degree_info(cd, fd)
        asm.ctx.static_condensation_packed(di_, S.shape[0], out["lc"].data_ptr(), rhs.data_ptr(), Sp.data_ptr(), g2.data_ptr(), None)
        U = unpack_symmetric(Sp, nf)
        iu = torch.triu_indices(nf, nf)
        assert torch.equal(U[:, iu[0], iu[1]], S[:, iu[1], iu[0]]) and torch.equal(g2, gg)      # S[c, j, i] = S(i, j)


def perturbed_mesh(oracle, N, seed, lo=(0.0, 0.0), hi=(1.0, 1.0)):
    """N x N generator mesh with interior nodes displaced by U(-0.1 h, 0.1 h)
    (the commented-out perturbation of convergence_test.cpp:176-187)."""
    mp, points, ptids = oracle.make_mesh(N, N, lo, hi)
    rng = np.random.default_rng(seed)
    h = (hi[0] - lo[0]) / N
    d = rng.uniform(-0.1 * h, 0.1 * h, size=points.shape)
    interior = np.ones(points.shape[0], dtype=bool)
    ij = np.arange(points.shape[0])
    i, j = ij % (N + 1), ij // (N + 1)
    interior &= (i > 0) & (i < N) & (j > 0) & (j < N)
    points[interior] += d[interior]
    return points, ptids


BATCH_CONFIGS = [
    # (cd, fd, quad, stab)  -- the BASELINE.json configs and their neighbours
    (2, 1, "tensor", "fancy"), (3, 2, "tensor", "fancy"), (4, 3, "tensor", "fancy"),
    (0, 1, "tensor", "fancy"), (0, 0, "tensor", "fancy"), (1, 1, "tensor", "fancy"),
    (2, 2, "tensor", "fancy"), (3, 3, "tensor", "fancy"), (1, 0, "tensor", "fancy"),
    (1, 2, "tensor", "fancy"), (2, 3, "tensor", "fancy"),
    (2, 1, "fan", "naive"), (3, 2, "fan", "naive"), (2, 1, "tensor", "naive"), (3, 2, "tensor", "naive"),
    (1, 1, "fan", "fancy"), (2, 2, "fan", "fancy"), (0, 1, "fan", "fancy"),
]


@pytest.mark.parametrize("cd,fd,kind,stabname", BATCH_CONFIGS)
def test_batch_matches_oracle(asm, oracle, cd, fd, kind, stabname):
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    N = 13                                   # 169 cells: not a multiple of any cells-per-wavefront
    points, ptids = perturbed_mesh(oracle, N, seed=1234 + cd * 10 + fd)
    asm.set_mesh(points, ptids)
    quad = pa.QUAD_TENSOR if kind == "tensor" else pa.QUAD_FAN
    stab = pa.STAB_FANCY if stabname == "fancy" else pa.STAB_NAIVE
    di = oracle.degrees(cd, fd)
    st, ref = oracle.local_ops_batch(points, ptids, di, quad, stab, want=("oper", "data", "stab", "lc"))
    assert st == 0
    out = asm.local_ops(cd, fd, quad, stab, want=("oper", "data", "stab", "lc", "info"))
    asm.synchronize()
    assert int(out["info"].abs().max().cpu()) == 0
    for k in ("oper", "data", "stab", "lc"):
        assert nerr_cells(to_rowcol(out[k]), ref[k]) < TOL, k
    # sub-range [first, first+n) with odd sizes
    for first, n in ((0, 1), (5, 7), (160, 9), (3, 0)):
        sub = asm.local_ops(cd, fd, quad, stab, first=first, n=n, want=("lc",))
        if n:
            assert nerr_cells(to_rowcol(sub["lc"]), ref["lc"][first:first + n]) < TOL
    # rhs (utils.hpp:153-174) with f = 2 pi^2 sin sin, and with caller-sampled values
    f = lambda x, y: 2.0 * math.pi ** 2 * math.sin(math.pi * x) * math.sin(math.pi * y)  # noqa: E731
    st, refr = oracle.local_ops_batch(points, ptids, di, quad, pa.STAB_NONE, fn=f, want=())
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, quad)
    assert nerr_cells(rhs.cpu().numpy()[:, :, None], refr["rhs"][:, :, None]) < TOL
    xyw = asm.quadrature_points(2 * cd, quad)
    import torch
    fv = 2.0 * math.pi ** 2 * torch.sin(math.pi * xyw[:, :, 0]) * torch.sin(math.pi * xyw[:, :, 1])
    rhs2 = asm.cell_rhs(cd, pa.capi.FN_SAMPLED, quad, fvals=fv.contiguous())
    assert nerr_cells(rhs2.cpu().numpy()[:, :, None], refr["rhs"][:, :, None]) < TOL


@pytest.mark.parametrize("cd,fd,kind,stabname", [(3, 2, "tensor", "fancy"), (2, 1, "fan", "naive"), (4, 3, "tensor", "fancy"),
                                                 (0, 1, "tensor", "fancy"), (2, 2, "tensor", "fancy")])      # the last two: dense fancy form
def test_pre_pass_in_pieces(asm, oracle, cd, fd, kind, stabname):
    """The split path (one-thread-per-cell pre-pass + cooperative kernel, hho_pre.hpp) runs in pieces when the record
    buffer is capped (pa_context_set_record_cap; 4 GiB by default, and pieces of 192 Ki cells when local matrices are written): same results piece
    by piece, bit for bit, as in one pass, and both match the oracle; pa_context_trim gives the buffer back."""
    import torch
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    N = 96                                   # 9216 cells: 3 pieces of 4096 / 4096 / 1024 under the smallest cap
    points, ptids = perturbed_mesh(oracle, N, seed=77 + cd)
    asm.set_mesh(points, ptids)
    quad = pa.QUAD_TENSOR if kind == "tensor" else pa.QUAD_FAN
    stab = pa.STAB_FANCY if stabname == "fancy" else pa.STAB_NAIVE
    whole = asm.local_ops(cd, fd, quad, stab, want=("oper", "lc", "info"))
    asm.synchronize()
    whole = {k: v.clone() for k, v in whole.items()}
    asm.ctx.trim()
    asm.ctx.set_record_cap(1 << 20)          # clamps to the minimum piece (4096 cells)
    pieces = asm.local_ops(cd, fd, quad, stab, want=("oper", "lc", "info"))
    asm.synchronize()
    asm.ctx.set_record_cap(4 << 30)
    for k in ("oper", "lc", "info"):
        assert torch.equal(whole[k], pieces[k]), k
    di = oracle.degrees(cd, fd)
    sel = np.r_[0:64, 4090:4102, 8190:8200, 9200:9216]
    st, ref = oracle.local_ops_batch(points, ptids[sel], di, quad, stab, want=("lc",))
    assert st == 0
    assert nerr_cells(to_rowcol(pieces["lc"][torch.as_tensor(sel, device=pieces["lc"].device)]), ref["lc"]) < TOL


def test_error_codes(asm):
    import ctypes as C
    import proton_amd as pa
    L = pa.capi.lib()
    asm.generate_mesh(4, 4)
    di, fell_back = pa.degree_info(7, 1)              # invalid pair reverts to equal order (utils.hpp:88-91)
    assert fell_back and (di.cell_deg, di.face_deg) == (1, 1)
    di, _ = pa.degree_info(4, 3)
    assert L.pa_local_ops_batch(asm.ctx.h, di, pa.QUAD_FAN, pa.STAB_NAIVE, 0, 16, None, None, None, None, None) == 3
    di5 = pa.DegreeInfo(5, 5, 6)                      # 2*recdeg = 12 needs golub_welsch
    assert L.pa_local_ops_batch(asm.ctx.h, di5, pa.QUAD_TENSOR, pa.STAB_NAIVE, 0, 16, None, None, None, None, None) == 3
    di, _ = pa.degree_info(3, 2)
    assert L.pa_local_ops_batch(asm.ctx.h, di, pa.QUAD_TENSOR, pa.STAB_FANCY, 10, 7, None, None, None, None, None) == 1
    assert L.pa_local_ops_batch(asm.ctx.h, di, pa.QUAD_TENSOR, 9, 0, 16, None, None, None, None, None) == 1
    # out-of-range point id is refused at upload (the kernels gather unchecked)
    pts = np.zeros((4, 2)); ids = np.array([[0, 1, 2, 7]], dtype=np.uint32)
    assert L.pa_mesh_upload(asm.ctx.h, pts.ctypes.data, 4, ids.ctypes.data, 1) == 1


def test_degenerate_cell_reports_pivot(asm):
    """A zero-area cell: Eigen's LLT would silently produce NaNs; the info word flags it."""
    import proton_amd as pa
    points = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [3.0, 0.0]])
    asm.set_mesh(points, np.array([[0, 1, 2, 3]], dtype=np.uint32))
    out = asm.local_ops(2, 1, pa.QUAD_TENSOR, pa.STAB_NAIVE, want=("lc", "info"))
    asm.synchronize()
    assert int(out["info"].cpu()[0]) != 0


def test_generated_mesh_matches_reference_generator(asm, oracle):
    import torch
    from proton_amd.batch import to_rowcol
    import proton_amd as pa
    N = 9
    asm.generate_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))       # obstacle.cpp:234-238 domain
    mp, points, ptids = oracle.make_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
    di = oracle.degrees(0, 1)
    st, ref = oracle.local_ops_batch(points, ptids, di, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    out = asm.local_ops(0, 1, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    assert nerr_cells(to_rowcol(out["lc"]), ref["lc"]) < TOL
    # row partition [3, 7): local cell 0 is global cell 3*N
    asm.generate_mesh(N, N, (-1.0, -1.0), (1.0, 1.0), rows=(3, 7))
    assert asm.ncells == 4 * N
    out = asm.local_ops(0, 1, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    assert nerr_cells(to_rowcol(out["lc"]), ref["lc"][3 * N:7 * N]) < TOL


@pytest.mark.parametrize("N,cd,fd,kind,stabname", [(1024, 3, 2, "tensor", "fancy"), (256, 2, 1, "tensor", "fancy"), (512, 0, 1, "tensor", "fancy"),
                                                    (256, 2, 1, "fan", "naive"),       # configs[1]: cuthho_square -M 256 -N 256 -k 1, uncut cells
                                                    (512, 3, 2, "fan", "naive")])      # configs[2] without its cut cells (those: test_gpu_cuthho.py)
def test_full_size_properties(asm, oracle, N, cd, fd, kind, stabname):
    """BASELINE.json sizes: properties that do not need the oracle on every cell.
    lc is symmetric, annihilates the interpolant of constants (cell dofs (1,0..), face dofs (1,0..)),
    and on the uniform generator mesh every cell equals the oracle's cell 0 up to rounding."""
    import torch
    import proton_amd as pa
    QUAD = pa.QUAD_TENSOR if kind == "tensor" else pa.QUAD_FAN
    STAB = pa.STAB_FANCY if stabname == "fancy" else pa.STAB_NAIVE
    asm.generate_mesh(N, N)
    out = asm.local_ops(cd, fd, QUAD, STAB, want=("lc", "info"))
    asm.synchronize()
    lc = out["lc"]
    assert int(out["info"].abs().max().cpu()) == 0
    assert bool(torch.isfinite(lc).all())
    scale = lc.abs().amax(dim=(1, 2))
    assert float(((lc - lc.transpose(1, 2)).abs().amax(dim=(1, 2)) / scale).max()) < TOL
    di = oracle.degrees(cd, fd)
    # lc times the interpolant of the constant 1 (cell dof 0 and the first dof of every face): the sum of those five columns.
    # (Not `lc @ one`: torch's batched matrix-vector product of the whole 4 GB array returned wrong values for the cells beyond an
    # offset into a REUSED block of its caching allocator -- after test_config5_slabs_equal_whole_mesh_at_full_size had left 17 GB
    # cached; the same product slice by slice, on the host, or after torch.cuda.empty_cache() was right: tools/r03_diag.py.)
    ones = [0] + [di.cbs + f * di.fbs for f in range(4)]
    res = lc[:, :, ones].sum(dim=2).abs().amax(dim=1) / scale
    bad = torch.nonzero(res > 1e-11).flatten()
    assert bad.numel() == 0, "lc does not annihilate the constants on %d cells, first %s last %s" % (bad.numel(), bad[:4].tolist(), bad[-4:].tolist())
    mp, points, ptids = oracle.make_mesh(N, N)
    st, ref = oracle.local_ops_batch(points, ptids, di, QUAD, STAB, first=0, n=1, want=("lc",))
    ref0 = torch.from_numpy(ref["lc"][0].T.copy()).to(lc.device)
    err = (lc - ref0).abs().amax(dim=(1, 2)) / ref0.abs().max()
    assert float(err.max()) < 1e-10        # coordinates i*h differ in the last bits from cell to cell
    # a few scattered cells against the oracle proper
    idx = [0, 1, N - 1, N * N // 2 + 17, N * N - 1]
    for c in idx:
        st, r = oracle.local_ops_batch(points, ptids, di, QUAD, STAB, first=c, n=1, want=("lc",))
        got = lc[c].cpu().numpy().T
        assert nerr(got, r["lc"][0]) < TOL


@pytest.mark.parametrize("cd,fd,nsample", [(3, 2, 4096), (2, 1, 4096), (4, 3, 1024)])
def test_headline_mesh_of_general_quadrilaterals_sampled_against_oracle(asm, oracle, cd, fd, nsample):
    """The bench workload `quad1024_k2_general` (1024 x 1024 cells, interior nodes displaced: no two cells
    congruent), attached as caller-owned device arrays: thousands of randomly chosen cells against the
    oracle cell by cell, and the whole batch through symmetry / kernel-of-constants."""
    import sys
    import os
    import torch
    import proton_amd as pa
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    N = 1024
    w = dict(bench.WORKLOADS["quad1024_k2_general"])
    pts_d, ids_d = bench.general_quad_mesh(torch, N, w["lo"], w["hi"], w["perturb"], asm.device)
    asm.ctx.mesh_attach_device(pts_d.data_ptr(), (N + 1) * (N + 1), ids_d.data_ptr(), N * N)
    out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc", "info"))
    asm.synchronize()
    lc = out["lc"]
    assert int(out["info"].abs().max().cpu()) == 0
    di = oracle.degrees(cd, fd)
    one = torch.zeros(di.msize, dtype=torch.float64, device=lc.device)
    one[0] = 1.0
    one[di.cbs::di.fbs] = 1.0
    for a in range(0, lc.shape[0], 131072):
        blk = lc[a:a + 131072]
        scale = blk.abs().amax(dim=(1, 2))
        assert float(((blk - blk.transpose(1, 2)).abs().amax(dim=(1, 2)) / scale).max()) < TOL
        assert float(((blk @ one).abs().amax(dim=1) / scale).max()) < 1e-10
    points, ptids = bench.workload_mesh(w)
    rng = np.random.default_rng(2026)
    cells = np.sort(rng.choice(N * N, size=nsample, replace=False))
    got = lc[torch.from_numpy(cells).to(lc.device)].cpu().numpy().transpose(0, 2, 1)
    errs = np.zeros(nsample)
    for i, c in enumerate(cells):
        st, r = oracle.local_ops_batch(points, ptids, di, pa.QUAD_TENSOR, pa.STAB_FANCY, first=int(c), n=1, want=("lc",))
        errs[i] = nerr(got[i], r["lc"][0])
    asm.generate_mesh(4, 4)              # detach from the caller-owned arrays before they go away
    assert np.median(errs) < 0.5 * TOL, np.median(errs)
    # On these small distorted cells (h ~ 1e-3, 10 % displacement) the ORACLE -- the reference's operation
    # order in double precision -- carries cond * eps itself at k = 3 (measured against the 50-digit
    # evaluation: oracle 1.0e-12, GPU 3.7e-13 on the worst cell of this sample).  Cells beyond the bar
    # against the oracle are therefore judged against the multiprecision evaluation of the same formulas.
    assert errs.max() < 5 * TOL, errs.max()
    suspects = np.nonzero(errs >= TOL)[0]
    assert len(suspects) <= max(5, nsample // 50)            # a tail, not a population: each one is checked below
    if len(suspects):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
        import mpmath as mpm
        import make_golden as mg
        for i in suspects:
            c = int(cells[i])
            P = [(mpm.mpf(float(x)), mpm.mpf(float(y))) for x, y in points[ptids[c].astype(np.int64)]]
            truth = mg.local_ops(P, [int(v) for v in ptids[c]], cd, fd, "tensor")
            ref = mg.to_np(truth["data"]) + mg.to_np(truth["fancy"])
            assert nerr(got[i], ref) < TOL, (c, nerr(got[i], ref))


def test_config5_slabs_2048_k3(asm, oracle):
    """configs[4] of BASELINE.json: 2048 x 2048, hho_degree_info(4,3), cell rows block-partitioned over
    8 ranks -- here the 8 slabs run one after the other on the one GPU (4 GB of lc each).  Every
    slab: finite, SPD pivots, symmetric, constants in the kernel, equal to the oracle's cell 0 up to
    the rounding of the coordinates; the first cell of each slab against the oracle proper."""
    import torch
    import proton_amd as pa
    from proton_amd.partition import row_partition
    N, cd, fd = 2048, 4, 3
    di = oracle.degrees(cd, fd)
    mp, points, ptids = oracle.make_mesh(N, N)
    st, ref = oracle.local_ops_batch(points, ptids, di, pa.QUAD_TENSOR, pa.STAB_FANCY, first=0, n=1, want=("lc",))
    ref0 = None
    out = None
    for rank in range(8):
        r0, r1 = row_partition(N, 8, rank)
        asm.generate_mesh(N, N, rows=(r0, r1))
        assert asm.ncells == (r1 - r0) * N
        out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc", "info"), out=out)
        asm.synchronize()
        lc = out["lc"]
        if ref0 is None:
            ref0 = torch.from_numpy(ref["lc"][0].T.copy()).to(lc.device)
            one = torch.zeros(di.msize, dtype=torch.float64, device=lc.device)
            one[0] = 1.0
            one[di.cbs::di.fbs] = 1.0
        assert int(out["info"].abs().max().cpu()) == 0
        worst_sym = worst_ker = worst_ref = 0.0
        for a in range(0, lc.shape[0], 65536):                 # chunks: no second 4 GB temporary
            blk = lc[a:a + 65536]
            scale = float(ref0.abs().max())
            worst_sym = max(worst_sym, float((blk - blk.transpose(1, 2)).abs().max()) / scale)
            worst_ker = max(worst_ker, float((blk @ one).abs().max()) / scale)
            worst_ref = max(worst_ref, float((blk - ref0).abs().max()) / scale)
        assert worst_sym < TOL and worst_ker < 1e-10 and worst_ref < 1e-9, (rank, worst_sym, worst_ker, worst_ref)
        c = r0 * N
        st, r = oracle.local_ops_batch(points, ptids, di, pa.QUAD_TENSOR, pa.STAB_FANCY, first=c, n=1, want=("lc",))
        got0 = lc[0].cpu().numpy().T
        e0 = nerr(got0, r["lc"][0])
        assert e0 < 5 * TOL, (rank, e0)
        if e0 >= TOL:
            # h = 1/2048 at k = 3: the entries that vanish on a square come out as +-5e-13 of the scale on EITHER side (the
            # rounding of x - barycenter), so the two can be 1e-12 apart; such a cell is judged against the 50-digit evaluation
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
            import mpmath as mpm
            import make_golden as mg
            P = [(mpm.mpf(float(x)), mpm.mpf(float(y))) for x, y in points[ptids[c].astype(np.int64)]]
            truth = mg.local_ops(P, [int(v) for v in ptids[c]], cd, fd, "tensor")
            ref = mg.to_np(truth["data"]) + mg.to_np(truth["fancy"])
            print("config 5, slab %d: %.2e against the oracle, %.2e (GPU) and %.2e (oracle) against the 50-digit evaluation"
                  % (rank, e0, nerr(got0, ref), nerr(r["lc"][0], ref)))
            assert nerr(got0, ref) < TOL, (rank, nerr(got0, ref))


@pytest.mark.parametrize("N,degree", [(16, 0), (16, 1), (32, 1)])
def test_obstacle_end_to_end_on_gpu_operators(asm, N, degree):
    """configs[3] plumbing: the obstacle driver (apps/obstacle/obstacle.cpp:47-227) fed with the GPU's
    local operators reproduces apps/obstacle/results/convergence.txt to its printed digits."""
    import obstacle_driver as od
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    REF = {16: (1.2833, 0.0588187), 32: (0.650286, 0.0171607)}

    def gpu_provider(msh, deg):
        asm.set_mesh(msh.points, msh.ptids)
        out = asm.local_ops(0, deg, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
        rhs = asm.cell_rhs(0, pa.capi.FN_OBSTACLE_RHS, pa.QUAD_TENSOR, dinc=1)
        asm.synchronize()
        return to_rowcol(out["lc"]), rhs.cpu().numpy()

    err, iters = od.run_obstacle(N, degree, local_provider=gpu_provider)
    assert abs(err - REF[N][degree]) / REF[N][degree] < 5e-6


@pytest.mark.parametrize("degree", [10, 11, 12, 14, 15])
def test_gauss_rules_beyond_five_nodes(asm, oracle, degree):
    """gauss_legendre() hands quadrature degrees >= 10 to golub_welsch (quadratures.hpp:32-75): 6 to 8 nodes in ascending
    order.  The points of integrate(msh, cl, degree) and integrate(msh, fc, ...) from the product against the oracle's
    restatement of the eigen-solve (and numpy's Gauss-Legendre rule), and make_rhs / project_function at cell degree 4
    with a degree increase (quadrature degree 10), which is where a driver meets these rules."""
    import torch
    import proton_amd as pa
    N = 3
    points, ptids = perturbed_mesh(oracle, N, seed=3)
    asm.set_mesh(points, ptids)
    xyw = asm.quadrature_points(degree, pa.QUAD_TENSOR).cpu().numpy()
    n = ((degree | 1) + 1) // 2
    assert xyw.shape == (N * N, n * n, 3)
    L = oracle.lib()
    nd, wt = np.zeros(8), np.zeros(8)
    assert L.hho_gauss_legendre(degree, oracle._dp(nd), oracle._dp(wt)) == n
    gx, gw = np.polynomial.legendre.leggauss(n)
    assert np.abs(nd[:n] - gx).max() < 5e-15 and np.abs(wt[:n] - gw).max() < 5e-15
    qx, qy, qw = np.zeros(64), np.zeros(64), np.zeros(64)
    for c in range(N * N):
        pts = np.ascontiguousarray(points[ptids[c].astype(np.int64)].reshape(8))
        nq = L.hho_cell_quadrature(oracle._dp(pts), oracle.QUAD_TENSOR, degree, oracle._dp(qx), oracle._dp(qy), oracle._dp(qw))
        assert nq == n * n
        ref = np.stack([qx[:nq], qy[:nq], qw[:nq]], axis=1)
        assert np.abs(xyw[c] - ref).max() < 1e-14
    if degree == 10:
        di = oracle.degrees(4, 3)
        rhs = asm.cell_rhs(4, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR, dinc=1).cpu().numpy()
        st, ref = oracle.local_ops_batch(points, ptids, di, oracle.QUAD_TENSOR, oracle.STAB_FANCY, fn=1, rhs_di=1, want=("lc",))
        assert st == 0 and np.abs(rhs - ref["rhs"]).max() < 1e-13 * np.abs(ref["rhs"]).max()
