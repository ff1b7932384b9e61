"""Condensed mode (SURVEY section 8 row A15 + the face-only numbering of row F1) through the C ABI.

The reference has no static condensation (its assemblers keep cell and face unknowns, hho.hpp:331): PARITY
UNPINNED BY THE REFERENCE.  What pins it here:
  * the 50-digit mpmath fixtures of S and g (tests/golden/local_ops.npz) and the oracle's restatement;
  * the algebraic identity that IS its definition: the condensed system, solved and followed by the recovery of the
    cell unknowns, gives the solution of the reference's uncondensed system (assembler<Mesh>, hho.hpp:252-463) --
    which the reference's own fixtures pin end to end (tests/test_gpu_obstacle.py, tests/test_gpu_cuthho.py);
  * index maps bit-exact against the reference's numbering with its cell block removed.
"""
import numpy as np
import pytest

from cases import CELLS
from test_gpu_parity import BATCH_CONFIGS, GOLD, gold_cases, nerr, perturbed_mesh, single_cell_mesh

pytestmark = pytest.mark.gpu
TOL = 1e-11          # S / g within 1e-11 of the fixtures


@pytest.fixture(scope="module")
def asm():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from proton_amd.batch import BatchAssembler
    return BatchAssembler(0)


def unpack(rec, nf):
    """packed records [n, nf(nf+1)/2 + nf] -> (S [n, nf, nf] symmetric, g [n, nf]) numpy"""
    rec = rec.cpu().numpy()
    n = rec.shape[0]
    S = np.zeros((n, nf, nf))
    for j in range(nf):
        for i in range(j + 1):
            S[:, i, j] = S[:, j, i] = rec[:, j * (j + 1) // 2 + i]
    return S, rec[:, nf * (nf + 1) // 2:]


@pytest.mark.parametrize("cname,cd,fd,kind", [c for c in gold_cases() if c[3] == "tensor"])
def test_fused_condensation_matches_golden(asm, cname, cd, fd, kind):
    import proton_amd as pa
    pts, ids = CELLS[cname]
    points, ptids = single_cell_mesh(pts, ids)
    asm.set_mesh(points, ptids)
    g = lambda what: GOLD[f"{cname}|{cd}|{fd}|{kind}|{what}"]  # noqa: E731
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    rec, info = asm.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs, want_info=True)
    asm.synchronize()
    assert int(info.cpu()[0]) == 0
    nf = 4 * (fd + 1)
    S, gg = unpack(rec, nf)
    assert nerr(S[0], g("S")) < TOL
    assert np.abs(gg[0] - g("g")[:, 0]).max() < TOL * max(np.abs(g("g")).max(), np.abs(g("rhs")).max())


@pytest.mark.parametrize("cd,fd,quad,stab", [c for c in BATCH_CONFIGS])
def test_fused_condensation_matches_unfused_and_oracle(asm, oracle, cd, fd, quad, stab):
    """Batch of general quadrilaterals: the fused kernel against (a) the two-kernel path (lc to HBM, then
    pa_static_condensation_batch) and (b) the oracle's condensation of the oracle's lc; the recovery against
    rec = A_TT^-1 [f_T | -A_TF] of the two-kernel path."""
    import torch
    import proton_amd as pa
    N = 6
    points, ptids = perturbed_mesh(oracle, N, seed=11)
    asm.set_mesh(points, ptids)
    q = pa.QUAD_TENSOR if quad == "tensor" else pa.QUAD_FAN
    s = pa.STAB_FANCY if stab == "fancy" else pa.STAB_NAIVE
    di, _ = pa.degree_info(cd, fd)
    nf = 4 * (di.face_deg + 1)
    cdd = di.cell_deg
    rhs = asm.cell_rhs(cdd, pa.capi.FN_SIN_SIN_RHS, q)
    rec, info = asm.condensed_ops(cd, fd, q, s, rhs=rhs, want_info=True)
    out = asm.local_ops(cd, fd, q, s, want=("lc",))
    S2, g2, rec2, info2 = asm.static_condensation(cdd, di.face_deg, out["lc"], rhs)
    asm.synchronize()
    assert int(info.abs().max().cpu()) == 0
    S, gg = unpack(rec, nf)
    S2h = S2.cpu().numpy()
    scale = np.abs(S2h).reshape(N * N, -1).max(axis=1)
    assert (np.abs(S - S2h).reshape(N * N, -1).max(axis=1) / scale).max() < TOL
    gscale = np.maximum(np.abs(g2.cpu().numpy()).max(axis=1), np.abs(rhs.cpu().numpy()).max(axis=1))
    assert (np.abs(gg - g2.cpu().numpy()).max(axis=1) / gscale).max() < TOL
    # oracle: condensation of the oracle's own lc
    odi = oracle.degrees(cd, fd)
    oq = oracle.QUAD_TENSOR if quad == "tensor" else oracle.QUAD_FAN
    os_ = oracle.STAB_FANCY if stab == "fancy" else oracle.STAB_NAIVE
    st, ref = oracle.local_ops_batch(points, ptids, odi, oq, os_, want=("lc", "rhs"), fn=1)
    assert st == 0
    for c in range(0, N * N, 5):
        st, So, go, reco = oracle.static_condensation(ref["lc"][c], ref["rhs"][c], odi.cbs)
        assert nerr(S[c], So) < TOL
        assert np.abs(gg[c] - go).max() < TOL * max(np.abs(go).max(), np.abs(ref["rhs"][c]).max())
    # recovery: u_T = rec[:,0] + rec[:,1:] u_F
    gen = torch.Generator(device="cpu").manual_seed(5)
    uF = torch.rand((N * N, nf), dtype=torch.float64, generator=gen).to(asm.device)
    uT = asm.condensed_recover(cd, fd, uF, q, s, rhs=rhs)
    r2 = rec2.cpu().numpy()                                  # [n, nf + 1, cbs]: column c of rec at r2[:, c, :]
    want = r2[:, 0, :] + np.einsum("nfc,nf->nc", r2[:, 1:, :], uF.cpu().numpy())
    assert np.abs(uT.cpu().numpy() - want).max() < 1e-10 * max(1.0, np.abs(want).max())


def _assemble_uncondensed(asm, cd, fd, lc, rhs, g):
    import scipy.sparse as sp
    r, c, v, rr, rv = asm.triplets(cd, fd, lc, rhs, g)
    info = asm.assembler_info(cd, fd)
    R, Cc, V = r.cpu().numpy().ravel(), c.cpu().numpy().ravel(), v.cpu().numpy().ravel()
    keep = R >= 0
    LHS = sp.csr_matrix((V[keep], (R[keep], Cc[keep])), shape=(info.system_size, info.system_size))
    RHS = np.zeros(info.system_size)
    RR, RV = rr.cpu().numpy().ravel(), rv.cpu().numpy().ravel()
    np.add.at(RHS, RR[RR >= 0], RV[RR >= 0])
    return LHS, RHS, info


@pytest.mark.parametrize("N,cd,fd", [(5, 2, 1), (6, 3, 2), (4, 4, 3), (7, 0, 1), (5, 1, 1)])
def test_condensed_triplets_index_maps_and_direct_csr(asm, N, cd, fd):
    """(i) the condensed triplets carry the reference's face numbering with the cell block removed, bit for bit;
    (ii) the directly assembled CSR (no sort) is bit-identical -- structure AND values -- to setFromTriplets of them."""
    import torch
    import proton_amd as pa
    asm.generate_mesh(N, N)
    di, _ = pa.degree_info(cd, fd)
    cbs, fbs = (di.cell_deg + 1) * (di.cell_deg + 2) // 2, di.face_deg + 1
    nf, ms = 4 * fbs, cbs + 4 * fbs
    rhs = asm.cell_rhs(di.cell_deg, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    rec = asm.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
    r, c, v, rr, rv = asm.condensed_triplets(cd, fd, rec, g)
    # the uncondensed assembler's maps (bit-exact against the oracle in test_gpu_assembler.py)
    out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    r0, c0, v0, rr0, rv0 = asm.triplets(cd, fd, out["lc"], rhs, g)
    ainfo = asm.assembler_info(cd, fd)
    cinfo = asm.condensed_info(cd, fd)
    assert cinfo.system_size == ainfo.system_size - cbs * N * N and cinfo.nf == nf
    R0 = r0.cpu().numpy().reshape(N * N, ms, ms)[:, cbs:, cbs:].reshape(N * N, nf * nf)
    C0 = c0.cpu().numpy().reshape(N * N, ms, ms)[:, cbs:, cbs:].reshape(N * N, nf * nf)
    shift = lambda a: np.where(a >= 0, a - cbs * N * N, -1)  # noqa: E731
    assert np.array_equal(r.cpu().numpy(), shift(R0)) and np.array_equal(c.cpu().numpy(), shift(C0))
    assert np.array_equal(rr.cpu().numpy(), shift(rr0.cpu().numpy()[:, cbs:]))
    # values: slot (i, j) = S(i, j)
    S, gg = unpack(rec, nf)
    assert np.array_equal(v.cpu().numpy().reshape(N * N, nf, nf), S)
    # direct CSR == setFromTriplets of the condensed triplets, bit for bit
    rowptr, colind, values = asm.csr_from_triplets(r, c, v, cinfo.system_size)
    rp2, ci2 = asm.condensed_csr_pattern(cd, fd)
    val2, rhs2 = asm.condensed_csr_fill(cd, fd, rec, g)
    asm.synchronize()
    assert cinfo.row_begin == 0 and cinfo.row_end == cinfo.system_size and cinfo.nnz_owned == colind.numel()
    assert torch.equal(rowptr, rp2) and torch.equal(colind, ci2)
    assert torch.equal(values, val2)
    RHS = torch.zeros(cinfo.system_size, dtype=torch.float64, device=asm.device)
    ok = rr.reshape(-1) >= 0
    RHS.index_add_(0, rr.reshape(-1)[ok].long(), rv.reshape(-1)[ok])
    assert torch.equal(RHS, rhs2)            # two addends per row at most: the sum does not depend on their order


@pytest.mark.parametrize("N,cd,fd", [(32, 2, 1), (64, 3, 2), (16, 4, 3), (24, 0, 1)])
def test_condensed_solve_equals_uncondensed(asm, N, cd, fd):
    """SURVEY A15's acceptance test on the HIP path: condensed system -> solve -> recovery gives the same u_F and u_T
    as the reference-shaped uncondensed system of the same GPU operators (config 1: 32^2 k=1; 64^2 k=2)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    import torch
    import proton_amd as pa
    lo, hi = ((-1.0, -1.0), (1.0, 1.0)) if cd == 0 else ((0.0, 0.0), (1.0, 1.0))
    asm.generate_mesh(N, N, lo, hi)
    di, _ = pa.degree_info(cd, fd)
    cbs, fbs = (di.cell_deg + 1) * (di.cell_deg + 2) // 2, di.face_deg + 1
    rhs = asm.cell_rhs(di.cell_deg, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    LHS, RHS, ainfo = _assemble_uncondensed(asm, cd, fd, out["lc"], rhs, g)
    full_ref = spl.spsolve(LHS.tocsc(), RHS)
    # condensed path, everything on the device
    rec = asm.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
    rowptr, colind = asm.condensed_csr_pattern(cd, fd)
    values, b = asm.condensed_csr_fill(cd, fd, rec, g)
    cinfo = asm.condensed_info(cd, fd)
    # (a) device Jacobi-PCG (solver_cg.hpp semantics), (b) host sparse LU of the same CSR
    x, reason, iters, rr = asm.conjugated_gradient(rowptr, colind, values, b.contiguous(), tol=1e-13, max_iter=20000)
    assert reason == 0, (reason, iters, rr)
    A = sp.csr_matrix((values.cpu().numpy(), colind.cpu().numpy(), rowptr.cpu().numpy()), shape=(cinfo.system_size,) * 2)
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()
    x_lu = spl.spsolve(A.tocsc(), b.cpu().numpy())
    uF_ref = full_ref[cbs * N * N:]
    scale = max(1e-300, np.abs(full_ref).max())
    assert np.abs(x_lu - uF_ref).max() < 1e-9 * scale
    assert np.abs(x.cpu().numpy() - uF_ref).max() < 1e-8 * scale
    # recovery of the cell unknowns with the face solution of the LU solve
    xF = torch.from_numpy(x_lu).to(asm.device)
    uF = asm.condensed_take_faces(cd, fd, xF, g)
    uT = asm.condensed_recover(cd, fd, uF, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
    full = asm.condensed_expand_solution(cd, fd, uT, xF)
    asm.synchronize()
    assert full.numel() == ainfo.system_size
    assert np.abs(full.cpu().numpy() - full_ref).max() < 1e-9 * scale
    # and the local data of both agree (assembler::take_local_data, hho.hpp:408-449)
    loc = asm.take_local_data(cd, fd, full, g)
    loc_ref = asm.take_local_data(cd, fd, torch.from_numpy(full_ref).to(asm.device), g)
    assert float((loc - loc_ref).abs().max()) < 1e-9 * scale


@pytest.mark.parametrize("N,cd,fd,parts,perturb", [(8, 2, 1, (0, 3, 8), 0.0), (9, 3, 2, (0, 2, 5, 9), 0.0), (6, 4, 3, (0, 1, 2, 6), 0.0),
                                                   (9, 3, 2, (0, 4, 9), 0.1), (7, 2, 1, (0, 1, 5, 7), 0.1)])
def test_condensed_slabs_equal_whole_mesh(asm, N, cd, fd, parts, perturb):
    """Row partition of the face-only system: every slab assembles the rows it owns from its own cells' records plus
    the packed top-face rows of the slab below (the whole exchange of a step); stacked, the slabs' CSR rows and
    right-hand sides are the whole-mesh system bit for bit.  perturb: general quadrilaterals -- the generator's numbering with
    displaced interior nodes (pa_mesh_set_points: every slab takes its node rows of the same whole-mesh displacement)."""
    import torch
    import proton_amd as pa
    from proton_amd.batch import BatchAssembler
    rng = np.random.default_rng(12345)
    shift = rng.uniform(-perturb / N, perturb / N, size=(N + 1, N + 1, 2))
    shift[0, :] = shift[-1, :] = 0.0
    shift[:, 0] = shift[:, -1] = 0.0
    xs = np.linspace(0.0, 1.0, N + 1)
    pts_all = np.stack(np.meshgrid(xs, xs, indexing="xy"), axis=-1) + shift          # [j][i] = (x_i, y_j)

    def slab(rows):
        a = BatchAssembler(0)
        a.generate_mesh(N, N, rows=rows)
        if perturb:
            p = torch.from_numpy(np.ascontiguousarray(pts_all[rows[0]:rows[1] + 1].reshape(-1, 2))).to(a.device)
            a.ctx.mesh_set_points(p.data_ptr(), p.shape[0])
            a.synchronize()
            with pytest.raises(Exception):
                a.ctx.mesh_set_points(p.data_ptr(), p.shape[0] + 1)
        rhs = a.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
        g = a.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
        rec = a.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
        return a, rec, g

    a, rec, g = slab((0, N))
    rp, ci = a.condensed_csr_pattern(cd, fd)
    va, ba = a.condensed_csr_fill(cd, fd, rec, g)
    a.synchronize()
    whole = (rp.cpu(), ci.cpu(), va.cpu(), ba.cpu())
    halo = None
    row_end, nnz_end = 0, 0
    for r0, r1 in zip(parts[:-1], parts[1:]):
        s, rec, g = slab((r0, r1))
        info = s.condensed_info(cd, fd)
        assert info.row_begin == row_end and bool(info.has_below) == (r0 > 0)
        assert info.halo_cells == (N if r1 < N else 0)
        rps, cis = s.condensed_csr_pattern(cd, fd)
        vs, bs = s.condensed_csr_fill(cd, fd, rec, g, halo_below=halo)
        halo = s.condensed_halo_pack(cd, fd, rec, g).clone() if r1 < N else None
        s.synchronize()
        nrows = info.row_end - info.row_begin
        assert torch.equal(rps.cpu() + nnz_end, whole[0][row_end:row_end + nrows + 1])
        assert torch.equal(cis.cpu(), whole[1][nnz_end:nnz_end + info.nnz_owned])
        assert torch.equal(vs.cpu(), whole[2][nnz_end:nnz_end + info.nnz_owned])
        assert torch.equal(bs.cpu(), whole[3][row_end:row_end + nrows])
        row_end, nnz_end = info.row_end, nnz_end + info.nnz_owned
    assert row_end == whole[3].numel() and nnz_end == whole[2].numel()


def test_config5_slabs_equal_whole_mesh_at_full_size():
    """configs[4] of BASELINE.json in the mode the N > 1 bench runs, at its own size: 2048 x 2048, k = 3, the cell rows
    block-partitioned into the 8 slabs of 8 GPUs -- here one after the other on the one GPU, each with its own context, its own
    records and the REAL packed top-face rows of the slab below (pa_condensed_halo_pack -> d_halo_below, what pa_comm_halo_exchange
    carries).  Stacked, the slabs' row pointers, column indices, CSR values and right-hand sides are the whole-mesh
    pa_condensed_csr_fill bit for bit (7.5 GB of values compared on the device)."""
    import gc
    import torch
    import proton_amd as pa
    from proton_amd.batch import BatchAssembler
    from proton_amd.partition import row_partition
    N, cd, fd, R = 2048, 4, 3, 8

    def slab(rows):
        a = BatchAssembler(0)
        a.generate_mesh(N, N, rows=rows)
        rhs = a.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
        g = a.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
        rec = a.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
        return a, rec, g

    a, rec, g = slab((0, N))
    rp, ci = a.condensed_csr_pattern(cd, fd)
    va, ba = a.condensed_csr_fill(cd, fd, rec, g)
    a.synchronize()
    assert va.numel() > 900_000_000 and bool(torch.isfinite(va[::997]).all())
    del rec
    halo = None
    row_end, nnz_end = 0, 0
    for r in range(R):
        r0, r1 = row_partition(N, R, r)
        s, srec, sg = slab((r0, r1))
        info = s.condensed_info(cd, fd)
        assert info.row_begin == row_end and bool(info.has_below) == (r0 > 0) and info.halo_cells == (N if r1 < N else 0)
        rps, cis = s.condensed_csr_pattern(cd, fd)
        vs, bs = s.condensed_csr_fill(cd, fd, srec, sg, halo_below=halo)
        halo = s.condensed_halo_pack(cd, fd, srec, sg).clone() if r1 < N else None
        s.synchronize()
        nrows = info.row_end - info.row_begin
        assert torch.equal(rps + nnz_end, rp[row_end:row_end + nrows + 1])
        assert torch.equal(cis, ci[nnz_end:nnz_end + info.nnz_owned])
        assert torch.equal(vs, va[nnz_end:nnz_end + info.nnz_owned]), "slab %d: CSR values differ from the whole-mesh fill" % r
        assert torch.equal(bs, ba[row_end:row_end + nrows])
        row_end, nnz_end = info.row_end, nnz_end + info.nnz_owned
        del s, srec, sg, rps, cis, vs, bs
        gc.collect()
    assert row_end == ba.numel() and nnz_end == va.numel()


def test_condensed_mode_on_an_uploaded_mesh_and_status_codes(asm, oracle):
    """explicit face tables (pa_mesh_set_faces) give the same condensed system as the generator's closed forms;
    bad arguments come back as status codes"""
    import torch
    import proton_amd as pa
    from proton_amd.capi import ProtonAmdError
    N, cd, fd = 5, 2, 1
    asm.generate_mesh(N, N)
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    rec = asm.condensed_ops(cd, fd, rhs=rhs)
    rp, ci = asm.condensed_csr_pattern(cd, fd)
    va, ba = asm.condensed_csr_fill(cd, fd, rec, g)
    ref = (rp.clone(), ci.clone(), va.clone(), ba.clone())
    mp, points, ptids = oracle.make_mesh(N, N)
    o = oracle.Assembler(mp, points, ptids, oracle.degrees(cd, fd))
    asm.set_mesh(points, ptids)
    asm.set_faces(o.cell_faces, o.faces, o.is_dir)
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    rec2 = asm.condensed_ops(cd, fd, rhs=rhs)
    rp2, ci2 = asm.condensed_csr_pattern(cd, fd)
    va2, ba2 = asm.condensed_csr_fill(cd, fd, rec2, g)
    asm.synchronize()
    assert torch.equal(rec, rec2)
    assert torch.equal(ref[0], rp2) and torch.equal(ref[1], ci2) and torch.equal(ref[2], va2) and torch.equal(ref[3], ba2)
    di, _ = pa.degree_info(cd, fd)
    with pytest.raises(ProtonAmdError) as e:       # no stabilization: A_TT singular
        asm.ctx.condensed_ops(di, pa.QUAD_TENSOR, pa.STAB_NONE, 0, N * N, None, rec.data_ptr(), None)
    assert e.value.status == 1
    with pytest.raises(ProtonAmdError) as e:       # range beyond the mesh
        asm.ctx.condensed_ops(di, pa.QUAD_TENSOR, pa.STAB_FANCY, 1, N * N, None, rec.data_ptr(), None)
    assert e.value.status == 1
    asm.generate_mesh(N, N, rows=(2, 4))             # a slab with a slab below needs its halo rows
    rec3 = asm.condensed_ops(cd, fd)
    with pytest.raises(ProtonAmdError) as e:
        asm.condensed_csr_fill(cd, fd, rec3, None)
    assert e.value.status == 1


def test_rccl_communicator_of_one_rank(asm):
    """pa_comm_* on the one GPU of the box: RCCL is found and bound at run time, a communicator of one rank comes up on
    the context's device, the collectives run on its side stream and pa_comm_wait orders the context's stream behind
    them (all-gather of one rank = a copy, all-reduce = the identity, the halo exchange has no neighbour)."""
    import torch
    from proton_amd import capi
    uid = capi.comm_unique_id()
    assert len(uid) == capi.COMM_ID_BYTES
    comm = capi.Comm(asm.ctx, 1, 0, uid)
    src = torch.arange(1000, dtype=torch.float64, device=asm.device)
    dst = torch.zeros_like(src)
    comm.allgather_start(src.data_ptr(), dst.data_ptr(), src.numel() * 8)
    comm.wait()
    acc = src.clone()
    comm.allreduce_sum_start(acc.data_ptr(), acc.numel())
    comm.wait()
    comm.halo_exchange_start(src.data_ptr(), 10, dst.data_ptr(), 10)      # rank 0 of 1: nobody above, nobody below
    comm.wait()
    asm.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(dst, src) and torch.equal(acc, src)
    comm.close()
