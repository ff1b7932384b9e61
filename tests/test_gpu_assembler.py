"""GPU assembler stage (assembler<Mesh>, hho.hpp:252-463) through the C ABI against the oracle's
restatement: triplet (row, col) sequences and right-hand-side row maps BIT-EXACT, values exact copies
of the local matrices, Dirichlet data and right-hand-side updates to rounding."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asm():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from proton_amd.batch import BatchAssembler
    return BatchAssembler(0)


def gpu_assembly(asm, N, cd, fd, rows=None):
    import proton_amd as pa
    asm.generate_mesh(N, N, rows=rows)
    out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    r, c, v, rr, rv = asm.triplets(cd, fd, out["lc"], rhs, g)
    asm.synchronize()
    return out["lc"], rhs, g, r, c, v, rr, rv


@pytest.mark.parametrize("N,cd,fd", [(5, 2, 1), (6, 3, 2), (4, 4, 3), (7, 0, 1), (5, 1, 1)])
def test_triplets_bit_exact_index_maps(asm, oracle, N, cd, fd):
    from proton_amd.batch import to_rowcol
    lc, rhs, g, r, c, v, rr, rv = gpu_assembly(asm, N, cd, fd)
    info = asm.assembler_info(cd, fd)
    mp, points, ptids = oracle.make_mesh(N, N)
    di = oracle.degrees(cd, fd)
    ref = oracle.Assembler(mp, points, ptids, di, bf_id=2)
    assert info.system_size == ref.system_size and info.num_other_faces == ref.num_other
    assert info.ncells_global == N * N and info.cell_base == 0 and info.face_base == 0
    # Dirichlet data (hho.hpp:381-386) on every face of the generator mesh
    gh = g.cpu().numpy()
    assert gh.shape[0] >= ref.nf
    assert np.abs(gh[:ref.nf] - ref.g).max() < 1e-13 * max(1.0, np.abs(ref.g).max())
    lch = to_rowcol(lc)
    rhsh = rhs.cpu().numpy()
    R, Cc, V, RR, RV = r.cpu().numpy(), c.cpu().numpy(), v.cpu().numpy(), rr.cpu().numpy(), rv.cpu().numpy()
    ms = di.msize
    for cell in range(N * N):
        tr, tc, tv, rrow, rval = ref.assemble_cell(cell, lch[cell], rhsh[cell])
        keep = R[cell] >= 0
        assert np.array_equal(keep, Cc[cell] >= 0)
        assert np.array_equal(R[cell][keep], tr)                      # bit-exact, in the reference's push order
        assert np.array_equal(Cc[cell][keep], tc)
        assert np.array_equal(V[cell][keep], tv)                      # exact copies of lhs(i,j)
        assert np.array_equal(RR[cell].astype(np.int64), rrow)
        assert np.abs(RV[cell] - rval).max() <= 1e-13 * max(1.0, np.abs(rval).max())
        # slot layout: i*msize + j
        assert np.array_equal(V[cell].reshape(ms, ms), lch[cell])


def test_partition_and_uploaded_mesh_agree_with_generated(asm, oracle):
    """cell rows [r0, r1) of the slab carry global indices; an uploaded mesh with explicit face tables
    (pa_mesh_set_faces) gives the same triplets as the closed-form generator tables."""
    import proton_amd as pa
    N, cd, fd = 6, 2, 1
    full = gpu_assembly(asm, N, cd, fd)
    R, Cc, RR = full[3].cpu().numpy(), full[4].cpu().numpy(), full[6].cpu().numpy()
    part = gpu_assembly(asm, N, cd, fd, rows=(2, 5))
    info = asm.assembler_info(cd, fd)
    assert info.cell_base == 2 * N and info.ncells_global == N * N
    assert np.array_equal(part[3].cpu().numpy(), R[2 * N:5 * N])
    assert np.array_equal(part[4].cpu().numpy(), Cc[2 * N:5 * N])
    assert np.array_equal(part[6].cpu().numpy(), RR[2 * N:5 * N])
    assert np.abs(part[7].cpu().numpy() - full[7].cpu().numpy()[2 * N:5 * N]).max() < 1e-13
    # uploaded mesh + explicit tables
    mp, points, ptids = oracle.make_mesh(N, N)
    di = oracle.degrees(cd, fd)
    ref = oracle.Assembler(mp, points, ptids, di)
    asm.set_mesh(points, ptids)
    asm.set_faces(ref.cell_faces, ref.faces, ref.is_dir)
    out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    r, c, v, rr, rv = asm.triplets(cd, fd, out["lc"], rhs, g)
    assert np.array_equal(r.cpu().numpy(), R) and np.array_equal(c.cpu().numpy(), Cc)
    assert np.array_equal(rr.cpu().numpy(), RR)
    # caller-sampled boundary function == built-in
    import torch
    xyw = asm.face_quadrature_points(fd)
    fv = torch.sin(math.pi * xyw[:, :, 0]) * torch.sin(math.pi * xyw[:, :, 1])
    g2 = asm.dirichlet_data(fd, pa.capi.FN_SAMPLED, fvals=fv.contiguous())
    assert float((g2 - g).abs().max()) < 1e-13


@pytest.mark.parametrize("cd,fd", [(2, 1), (3, 2)])
def test_config1_plumbing_poisson_solve(asm, oracle, cd, fd):
    """configs[0]: 32x32 k=1 Laplacian, GPU local operators + GPU triplets, host sparse solve:
    same solution as the all-oracle path and the expected L2 order."""
    import scipy.sparse as sp
    import poisson_driver as pd
    errs = []
    for N in (16, 32):
        lc, rhs, g, r, c, v, rr, rv = gpu_assembly(asm, N, cd, fd)
        info = asm.assembler_info(cd, fd)
        R, Cc, V = r.cpu().numpy().ravel(), c.cpu().numpy().ravel(), v.cpu().numpy().ravel()
        keep = R >= 0
        LHS = sp.csr_matrix((V[keep], (R[keep], Cc[keep])), shape=(info.system_size, info.system_size))
        RHS = np.zeros(info.system_size)
        RR, RV = rr.cpu().numpy().ravel(), rv.cpu().numpy().ravel()
        ok = RR >= 0
        np.add.at(RHS, RR[ok], RV[ok])
        sol = pd.solve(LHS, RHS)
        LHS_o, RHS_o, ref, di = pd.oracle_assembly(N, cd, fd)
        assert abs(LHS - LHS_o).max() < 1e-11 * abs(LHS_o).max()
        assert np.abs(RHS - RHS_o).max() < 1e-12 * max(1.0, np.abs(RHS_o).max())
        sol_o = pd.solve(LHS_o, RHS_o)
        assert np.abs(sol - sol_o).max() < 1e-9
        errs.append(pd.l2_error(ref, di, sol))
    assert math.log2(errs[0] / errs[1]) > fd + 2 - 0.3


@pytest.mark.parametrize("N,cd,fd", [(6, 2, 1), (9, 3, 2), (5, 0, 1)])
def test_device_csr_equals_set_from_triplets(asm, N, cd, fd):
    """pa_csr_from_triplets == scipy's COO -> CSR with summed duplicates (SparseMatrix::setFromTriplets,
    hho.hpp:451-455): structure bit-exact, values exact (a face-face entry is the sum of at most two
    contributions, and two-term sums do not depend on the order)."""
    import scipy.sparse as sp
    lc, rhs, g, r, c, v, rr, rv = gpu_assembly(asm, N, cd, fd)
    info = asm.assembler_info(cd, fd)
    rowptr, colind, values = asm.csr_from_triplets(r, c, v, info.system_size)
    asm.synchronize()
    R, Cc, V = r.cpu().numpy().ravel(), c.cpu().numpy().ravel(), v.cpu().numpy().ravel()
    keep = R >= 0
    ref = sp.coo_matrix((V[keep], (R[keep], Cc[keep])), shape=(info.system_size, info.system_size)).tocsr()
    ref.sum_duplicates(); ref.sort_indices()
    assert np.array_equal(rowptr.cpu().numpy(), ref.indptr.astype(np.int64))
    assert np.array_equal(colind.cpu().numpy(), ref.indices.astype(np.int32))
    assert np.array_equal(values.cpu().numpy(), ref.data)
    assert int(rowptr[-1]) == ref.nnz


@pytest.mark.parametrize("N,cd,fd", [(6, 2, 1), (9, 3, 2), (5, 0, 1), (7, 4, 3), (8, 1, 1), (6, 0, 0), (33, 3, 2)])
def test_direct_assembler_csr_equals_set_from_triplets(asm, N, cd, fd):
    """pa_assembler_csr_pattern / _fill -- assembler<Mesh>'s own system (cell + face unknowns, hho.hpp:298-335, 344-406,
    451-455) built from the face adjacency, no triplets, no sort -- is bit for bit the CSR pa_csr_from_triplets makes of
    pa_triplets_batch: row pointers, column indices, values; the right-hand side is the scatter-add of the triplet path's
    per-row sums in cell order (hho.hpp:401, 405)."""
    import torch
    lc, rhs, g, r, c, v, rr, rv = gpu_assembly(asm, N, cd, fd)
    info = asm.assembler_info(cd, fd)
    rowptr, colind, values = asm.csr_from_triplets(r, c, v, info.system_size)
    rp2, ci2 = asm.assembler_csr_pattern(cd, fd)
    va2, RHS2 = asm.assembler_csr_fill(cd, fd, lc, rhs, g)
    asm.synchronize()
    assert rp2.numel() == info.system_size + 1 and int(rp2[-1]) == int(rowptr[-1]) == ci2.numel()
    assert torch.equal(rp2, rowptr) and torch.equal(ci2, colind)
    assert torch.equal(va2, values)
    RR, RV = rr.cpu().numpy().ravel(), rv.cpu().numpy().ravel()
    want = np.zeros(info.system_size)
    keep = RR >= 0
    np.add.at(want, RR[keep], RV[keep])
    assert np.array_equal(RHS2.cpu().numpy(), want)
    # homogeneous data, no cell right-hand side: the values are the same, the right-hand side is zero
    va3, RHS3 = asm.assembler_csr_fill(cd, fd, lc)
    asm.synchronize()
    assert torch.equal(va3, values) and float(RHS3.abs().max()) == 0.0


def test_direct_assembler_csr_refuses_a_slab(asm):
    """a slab of a partitioned mesh owns a block of FACE rows (pa_condensed_*): the cell + face system is whole-mesh only"""
    import ctypes as C
    import proton_amd as pa
    asm.generate_mesh(8, 8, rows=(2, 5))
    di, _ = pa.capi.degree_info(2, 1)
    out = pa.capi.AssemblerCsrInfo()
    assert pa.capi.lib().pa_assembler_csr_query(asm.ctx.h, di, C.byref(out)) == 1
    asm.generate_mesh(8, 8)
    assert pa.capi.lib().pa_assembler_csr_query(asm.ctx.h, di, C.byref(out)) == 0 and out.nrows == 6 * 64 + 2 * (2 * 8 * 9 - 32)


def test_device_csr_edge_cases(asm):
    """empty input, all slots dropped, long runs of duplicates summed in push order, sizes around the scan tile"""
    import torch
    dev = asm.device
    rp, ci, va = asm.csr_from_triplets(torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev),
                                       torch.empty(0, dtype=torch.float64, device=dev), 4)
    assert rp.tolist() == [0, 0, 0, 0, 0] and ci.numel() == 0
    m1 = torch.full((10,), -1, dtype=torch.int32, device=dev)
    rp, ci, va = asm.csr_from_triplets(m1, m1, torch.ones(10, dtype=torch.float64, device=dev), 3)
    assert rp.tolist() == [0, 0, 0, 0] and ci.numel() == 0
    rng = np.random.default_rng(0)
    for n in (1, 2047, 2048, 2049, 100_000):
        rows = rng.integers(0, 50, n).astype(np.int32)
        cols = rng.integers(0, 7, n).astype(np.int32)
        vals = rng.standard_normal(n)
        drop = rng.random(n) < 0.2
        rows[drop] = -1
        rp, ci, va = asm.csr_from_triplets(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev), torch.from_numpy(vals).to(dev), 50)
        want = {}
        for i in range(n):                                  # left-to-right sums, the order of the pushes
            if rows[i] >= 0:
                want[(int(rows[i]), int(cols[i]))] = want.get((int(rows[i]), int(cols[i])), 0.0) + vals[i]
        keys = sorted(want)
        rp, ci, va = rp.cpu().numpy(), ci.cpu().numpy(), va.cpu().numpy()
        assert len(keys) == len(ci) == rp[-1]
        got_rows = np.repeat(np.arange(50), np.diff(rp))
        assert [(int(a), int(b)) for a, b in zip(got_rows, ci)] == keys
        assert np.array_equal(va, np.array([want[k] for k in keys]))


def test_device_conjugated_gradient(asm):
    """pa_conjugated_gradient (solver_cg.hpp:63-144) on the device CSR of a Poisson system: solution equals the
    sparse direct solve, iteration count close to the same recurrences run in numpy, exit reasons in the reference's order."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    N, cd, fd = 16, 2, 1
    lc, rhs, g, r, c, v, rr, rv = gpu_assembly(asm, N, cd, fd)
    info = asm.assembler_info(cd, fd)
    n = info.system_size
    rowptr, colind, values = asm.csr_from_triplets(r, c, v, n)
    b = torch.zeros(n, dtype=torch.float64, device=asm.device)
    ok = rr.reshape(-1) >= 0
    b.index_add_(0, rr.reshape(-1)[ok].long(), rv.reshape(-1)[ok])
    x, reason, iters, relres = asm.conjugated_gradient(rowptr, colind, values, b, tol=1e-12, max_iter=3 * n, precond=True)
    asm.synchronize()
    A = sp.csr_matrix((values.cpu().numpy(), colind.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n))
    ref = spla.spsolve(A.tocsc(), b.cpu().numpy())
    assert reason == 0 and relres < 1e-12
    assert np.abs(x.cpu().numpy() - ref).max() < 1e-9 * np.abs(ref).max()
    # the same recurrences in numpy (Eigen's sequential sums replaced by numpy's): same iteration count +- a few
    bb = b.cpu().numpy(); iA = 1.0 / A.diagonal()
    xx = np.zeros(n); rres = bb.copy(); d = iA * rres; nr0 = np.linalg.norm(rres); it = 0
    while True:
        y = A @ d; z = iA * rres; rho = rres @ z; alpha = rho / (d @ y)
        xx += alpha * d; rres -= alpha * y
        if np.linalg.norm(rres) / nr0 < 1e-12:
            break
        z = iA * rres; d = z + (rres @ z) / rho * d; it += 1
    assert abs(it - iters) <= max(3, it // 50), (it, iters)
    # max_iter exit: the reference stops when iter > max_iter (solver_cg.hpp:112-115)
    x2, reason2, iters2, _ = asm.conjugated_gradient(rowptr, colind, values, b, tol=1e-30, max_iter=5, precond=True)
    assert reason2 == 2 and iters2 == 6
    # divergence threshold below the first relative residual: DIVERGED
    x3, reason3, iters3, _ = asm.conjugated_gradient(rowptr, colind, values, b, tol=1e-30, div=1e-6, max_iter=50, precond=False)
    assert reason3 == 1 and iters3 == 0


def test_multi_gpu_step_in_pieces_equals_whole_mesh(asm):
    """The N > 1 step of bench.py computes a slab's top cell row first (its packed top-face rows travel while the rest
    is computed) and the slabs rank after rank: cell for cell and bit for bit the records of one pass over the whole mesh."""
    import torch
    import proton_amd as pa
    from proton_amd.partition import row_partition
    N, cd, fd, world = 37, 3, 2, 3
    di, _ = pa.degree_info(cd, fd)

    def records(rows, split_top):
        asm.generate_mesh(N, N, rows=rows)
        n = asm.ncells
        rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
        ci = asm.condensed_info(cd, fd)
        rec = torch.empty((n, ci.cond_doubles), dtype=torch.float64, device=asm.device)
        top = N if split_top else 0
        if top:
            asm.ctx.condensed_ops(di, pa.QUAD_TENSOR, pa.STAB_FANCY, n - top, top, rhs[n - top:].data_ptr(), rec[n - top:].data_ptr(), None)
        asm.ctx.condensed_ops(di, pa.QUAD_TENSOR, pa.STAB_FANCY, 0, n - top, rhs.data_ptr(), rec.data_ptr(), None)
        asm.synchronize()
        return rec.cpu()

    whole = records((0, N), False)
    got = [records(row_partition(N, world, r), True) for r in range(world)]
    assert torch.equal(torch.cat(got), whole)
