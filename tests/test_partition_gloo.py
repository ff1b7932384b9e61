"""N > 1 path on CPU: world_size-2 (and 3) gloo, no GPU.  The cells shard by rows; every rank assembles the rows of the
face-only system it owns; the one exchange of a step is the packed top-face rows of each slab's top cell row, one slab
up.  Here the oracle stands in for the GPU kernels (records per cell), numpy for the assembly of a slab, and what is
under test is the product's HOST logic: the closed-form row partition (pa_condensed_partition_info, the library loads
without a GPU), the halo layout it announces, and the host-staged twin of pa_comm_halo_exchange_start
(proton_amd.partition.HostStagedHalo).  The stacked slabs must be the whole-mesh system."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from proton_amd.partition import HostStagedAllgather, HostStagedHalo, cell_counts, row_partition


def test_row_partition_covers_all_rows():
    for N in (1, 5, 8, 1024, 2048):
        for world in (1, 2, 3, 4, 8):
            edges = [row_partition(N, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
            assert sum(cell_counts(N, N, world)) == N * N


def test_partition_info_matches_the_assemblers_compress_table(oracle):
    """row ranges of the slabs from the closed forms == positions in the reference's compress table (hho.hpp:305-323)"""
    from proton_amd import capi
    for N, world, fd in ((5, 2, 1), (7, 3, 2), (8, 8, 0), (6, 4, 3)):
        mp_, points, ptids = oracle.make_mesh(N, N)
        ref = oracle.Assembler(mp_, points, ptids, oracle.degrees(fd + 1, fd))
        di, _ = capi.degree_info(fd + 1, fd)
        fbs, face_row = fd + 1, 2 * N + 1
        end = 0
        for r in range(world):
            r0, r1 = row_partition(N, world, r)
            info = capi.condensed_partition_info(N, N, (r0, r1), di)
            own = [f for f in range(r0 * face_row, min(r1 * face_row, ref.nf)) if not ref.is_dir[f]]
            assert info.num_other_faces == ref.num_other and info.system_size == fbs * ref.num_other
            assert info.row_begin == end and info.row_end == end + fbs * len(own)
            if own:
                assert info.row_begin == fbs * ref.compress[own[0]] and info.row_end == fbs * (ref.compress[own[-1]] + 1)
            assert info.halo_cells == (N if r1 < N else 0) and bool(info.has_below) == (r0 > 0)
            assert info.halo_doubles == fbs * (4 * fbs + 1) and info.nf == 4 * fbs
            end = info.row_end
        assert end == fbs * ref.num_other


def _slab(N, cd, fd, r0, r1):
    """records of the slab's cells by the oracle: S [n, nf, nf], g [n, nf]; plus the assembler of the whole mesh"""
    import oracle_lib as o
    mp_, points, ptids = o.make_mesh(N, N)
    di = o.degrees(cd, fd)
    n = (r1 - r0) * N
    st, out = o.local_ops_batch(points, ptids, di, o.QUAD_TENSOR, o.STAB_FANCY, first=r0 * N, n=n, fn=1, want=("lc",))
    assert st == 0
    nf = 4 * di.fbs
    S, g = np.zeros((n, nf, nf)), np.zeros((n, nf))
    for c in range(n):
        st, S[c], g[c], rec = o.static_condensation(out["lc"][c], out["rhs"][c], di.cbs)
    return S, g, o.Assembler(mp_, points, ptids, di, bf_id=2), di


def _cell_rows(ref, di, c, S, g):
    """assembler::assemble on the condensed block of global cell c -> (rows, cols, vals) kept triplets and rhs (rows, vals)"""
    fbs, nf = di.fbs, 4 * di.fbs
    idx, dd = np.zeros(nf, dtype=np.int64), np.zeros(nf)
    for lf in range(4):
        f = int(ref.cell_faces[c, lf])
        for k in range(fbs):
            idx[lf * fbs + k] = -1 if ref.is_dir[f] else ref.compress[f] * fbs + k
            dd[lf * fbs + k] = ref.g[f, k] if ref.is_dir[f] else 0.0
    keep = idx >= 0
    rr, cc = np.meshgrid(idx, idx, indexing="ij")
    m = keep[:, None] & keep[None, :]
    b = g - S[:, ~keep] @ dd[~keep]
    return rr[m], cc[m], S[m], idx[keep], b[keep]


def _pack_halo(ref, di, N, r1, S, g):
    """the layout pa_condensed_halo_pack announces: per cell of the slab's top row fbs x nf values S(2 fbs + k, :), then
    the fbs right-hand-side contributions with the Dirichlet columns already moved over"""
    fbs, nf = di.fbs, 4 * di.fbs
    out = np.zeros((N, fbs * (nf + 1)))
    for i in range(N):
        c = (r1 - 1) * N + i
        loc = S.shape[0] - N + i
        dd = np.zeros(nf)
        dirichlet = np.zeros(nf, dtype=bool)
        for lf in range(4):
            f = int(ref.cell_faces[c, lf])
            if ref.is_dir[f]:
                dirichlet[lf * fbs:(lf + 1) * fbs] = True
                dd[lf * fbs:(lf + 1) * fbs] = ref.g[f]
        rows = slice(2 * fbs, 3 * fbs)
        out[i, :fbs * nf] = S[loc, rows, :].reshape(-1)
        out[i, fbs * nf:] = g[loc, rows] - S[loc, rows][:, dirichlet] @ dd[dirichlet]
    return out


def _assemble_owned(ref, di, N, r0, r1, info, S, g, halo):
    """dense owned rows x all columns of the face-only system, and their right-hand side"""
    fbs, nf = di.fbs, 4 * di.fbs
    nrows = info.row_end - info.row_begin
    A, b = np.zeros((nrows, info.system_size)), np.zeros(nrows)
    own = lambda r: (r >= info.row_begin) & (r < info.row_end)  # noqa: E731
    for cl in range(S.shape[0]):
        rr, cc, vv, br, bv = _cell_rows(ref, di, r0 * N + cl, S[cl], g[cl])
        m = own(rr)
        np.add.at(A, (rr[m] - info.row_begin, cc[m]), vv[m])
        mb = own(br)
        np.add.at(b, br[mb] - info.row_begin, bv[mb])
    if halo is not None:                     # rows of this slab's bottom faces: the contribution of the cells below
        for i in range(N):
            c = (r0 - 1) * N + i
            top = int(ref.cell_faces[c, 2])
            assert not ref.is_dir[top]
            for k in range(fbs):
                row = ref.compress[top] * fbs + k - info.row_begin
                for lf in range(4):
                    f = int(ref.cell_faces[c, lf])
                    if ref.is_dir[f]:
                        continue
                    for kp in range(fbs):
                        A[row, ref.compress[f] * fbs + kp] += halo[i, k * nf + lf * fbs + kp]
                b[row] += halo[i, fbs * nf + k]
    return A, b


def _worker(rank, world, port, N, cd, fd, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from proton_amd import capi
        r0, r1 = row_partition(N, world, rank)
        S, g, ref, di = _slab(N, cd, fd, r0, r1)
        pdi, _ = capi.degree_info(cd, fd)
        info = capi.condensed_partition_info(N, N, (r0, r1), pdi)
        send = torch.from_numpy(_pack_halo(ref, di, N, r1, S, g)) if info.halo_cells else None
        recv = torch.zeros((N, info.halo_doubles), dtype=torch.float64) if info.has_below else None
        if send is not None:
            assert tuple(send.shape) == (info.halo_cells, info.halo_doubles)
        dist.barrier()
        HostStagedHalo(rank, world)(send, recv)
        A, b = _assemble_owned(ref, di, N, r0, r1, info, S, g, None if recv is None else recv.numpy())
        # the north star's literal collective behind the halo exchange: all-gather of every rank's owned rows (padded to the
        # largest slab's count), after which EVERY rank holds the whole face-only system (HostStagedAllgather = the host twin of
        # pa_comm_allgather_start)
        rows_all = [capi.condensed_partition_info(N, N, row_partition(N, world, r), pdi) for r in range(world)]
        nrows = [int(i.row_end - i.row_begin) for i in rows_all]
        count = max(nrows) * (info.system_size + 1)
        mine = torch.from_numpy(np.concatenate([A, b[:, None]], axis=1).reshape(-1))
        gathered = torch.empty(world * count, dtype=torch.float64)
        HostStagedAllgather(rank, world)(mine, gathered, count)
        whole = np.concatenate([gathered[r * count:r * count + nrows[r] * (info.system_size + 1)].numpy().reshape(nrows[r], info.system_size + 1)
                                for r in range(world)])
        q.put((rank, (int(info.row_begin), A, b, whole)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,cd,fd,world", [(6, 2, 1, 2), (5, 3, 2, 2), (7, 0, 1, 3)])
def test_slabs_with_halo_exchange_equal_the_whole_system(N, cd, fd, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, cd, fd, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the whole mesh in one process
    S, g, ref, di = _slab(N, cd, fd, 0, N)
    from proton_amd import capi
    pdi, _ = capi.degree_info(cd, fd)
    info = capi.condensed_partition_info(N, N, (0, N), pdi)
    A, b = _assemble_owned(ref, di, N, 0, N, info, S, g, None)
    assert A.shape[0] == info.system_size and abs(A - A.T).max() < 1e-12 * abs(A).max()
    row = 0
    for r in range(world):
        begin, Ar, br, whole = results[r]
        # after the all-gather every rank holds the same whole system [A | b]
        assert whole.shape == (info.system_size, info.system_size + 1)
        assert np.abs(whole[:, :-1] - A).max() <= 1e-15 * np.abs(A).max() and np.abs(whole[:, -1] - b).max() <= 1e-15 * max(1.0, np.abs(b).max())
        assert np.array_equal(whole, results[0][3])
        assert begin == row
        # the same sums of the same two addends: equal up to the order of the additions of a shared face's two cells
        assert np.abs(Ar - A[row:row + Ar.shape[0]]).max() <= 1e-15 * np.abs(A).max()
        assert np.abs(br - b[row:row + Ar.shape[0]]).max() <= 1e-15 * max(1.0, np.abs(b).max())
        row += Ar.shape[0]
    assert row == info.system_size


def _cut_worker(rank, world, port, N, k, q):
    """One rank of the cut workload (config 3) under the row rule, the oracle standing in for the kernels: every rank runs the SAME
    whole-mesh preprocessing (what pa_cut_preprocess_rows does on the host), keeps the cells of its rows, computes their merged local
    matrices -- and the ranks' checksums meet in an all-reduce, as bench.py's `exchange_checked` of mode L does."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as o
        msh = o.CutMesh(N, refsteps=3)
        di = o.degrees(k + 1, k)
        r0, r1 = row_partition(N, world, rank)
        mine = range(r0 * N, r1 * N)
        cut = [c for c in mine if msh.cell_loc[c] == o.CUT_ON_INTERFACE]
        sums = np.zeros(3)
        for c in mine:
            st, oper, data = msh.laplacian(c, di)
            assert st == 0
            st, stab = msh.cut_stabilization(c, di)
            assert st == 0
            lc = data + stab
            sums += (lc.sum(), np.abs(lc).sum(), 1.0)
        t = torch.from_numpy(sums.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        counts = [None] * world
        dist.all_gather_object(counts, cut)
        q.put((rank, (sums, t.numpy().copy(), counts)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,k,world", [(10, 1, 2), (9, 0, 3)])
def test_cut_cells_shard_by_the_same_row_rule(N, k, world):
    """"cut cells (config 3) are distributed by the same row rule" -- the ranks' cut cells are disjoint, their union
    is the whole mesh's, and the all-reduced checksums of the ranks' local matrices are those of one process."""
    import oracle_lib as o
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cut_worker, args=(r, world, port, N, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    msh = o.CutMesh(N, refsteps=3)
    di = o.degrees(k + 1, k)
    all_cut = [c for c in range(msh.nc) if msh.cell_loc[c] == o.CUT_ON_INTERFACE]
    ref = np.zeros(3)
    for c in range(msh.nc):
        st, oper, data = msh.laplacian(c, di)
        st, stab = msh.cut_stabilization(c, di)
        lc = data + stab
        ref += (lc.sum(), np.abs(lc).sum(), 1.0)
    gathered = results[0][2]
    assert sorted(sum(gathered, [])) == all_cut and sum(len(g) for g in gathered) == len(all_cut) > 0
    for r in range(world):
        assert results[r][2] == gathered
        tot = results[r][1]
        assert tot[2] == msh.nc
        assert abs(tot[0] - ref[0]) <= 1e-12 * ref[1] and abs(tot[1] - ref[1]) <= 1e-12 * ref[1]
