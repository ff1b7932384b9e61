"""N > 1 path on CPU: world_size-2 gloo.  Each rank computes the condensed face blocks of its cell
rows (with the oracle standing in for the GPU kernels), the exchange gathers them, and the result
must equal the single-process computation in global cell order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from proton_amd.partition import ChunkedExchange, CondensedExchange, cell_counts, condensed_per_cell, row_partition, unpack_symmetric


def test_row_partition_covers_all_rows():
    for N in (1, 5, 8, 1024, 2048):
        for world in (1, 2, 3, 4, 8):
            edges = [row_partition(N, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
            assert sum(cell_counts(N, N, world)) == N * N


def _condensed_blocks(N, cd, fd, first, n):
    import oracle_lib as o
    mp_, points, ptids = o.make_mesh(N, N)
    di = o.degrees(cd, fd)
    st, out = o.local_ops_batch(points, ptids, di, o.QUAD_TENSOR, o.STAB_FANCY, first=first, n=n, fn=1, want=("lc",))
    assert st == 0
    nf = 4 * di.fbs
    Sb, gb = np.zeros((n, nf, nf)), np.zeros((n, nf))
    for c in range(n):
        st, S, g, rec = o.static_condensation(out["lc"][c], out["rhs"][c], di.cbs)
        Sb[c] = S.T                                  # column-major, as the device kernel writes it
        gb[c] = g
    return Sb, gb


def _worker(rank, world, port, N, cd, fd, q, packed=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as o
        di = o.degrees(cd, fd)
        per = condensed_per_cell(di.fbs, packed)
        counts = cell_counts(N, N, world)
        r0, r1 = row_partition(N, world, rank)
        ex = CondensedExchange(counts, per, rank, torch.device("cpu"))
        nf = 4 * di.fbs
        Sb, gb = _condensed_blocks(N, cd, fd, r0 * N, (r1 - r0) * N)
        S_view, g_view = ex.local_S_g(nf)
        if packed:                                   # upper triangle, column-packed: the bench's exchange format
            iu = np.triu_indices(nf)
            Sp = np.zeros((Sb.shape[0], nf * (nf + 1) // 2))
            Sp[:, iu[1] * (iu[1] + 1) // 2 + iu[0]] = Sb[:, iu[1], iu[0]]      # Sb is column-major: Sb[c, j, i] = S(i, j)
            S_view.copy_(torch.from_numpy(Sp))
        else:
            S_view.copy_(torch.from_numpy(Sb))
        g_view.copy_(torch.from_numpy(gb))
        dist.barrier()
        ex.exchange()
        parts = [ex.gathered_S_g(r, nf)[0] for r in range(world)]
        if packed:
            parts = [unpack_symmetric(p_, nf) for p_ in parts]
        fullS = torch.cat(parts, dim=0).numpy()
        fullg = torch.cat([ex.gathered_S_g(r, nf)[1] for r in range(world)], dim=0).numpy()
        q.put((rank, (fullS, fullg)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,cd,fd,packed", [(6, 2, 1, False), (5, 3, 2, False), (5, 3, 2, True), (7, 0, 1, True)])
def test_two_rank_exchange_matches_single_process(N, cd, fd, packed):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, cd, fd, q, packed)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refS, refg = _condensed_blocks(N, cd, fd, 0, N * N)
    iu = np.triu_indices(refS.shape[1])
    for r in range(world):
        if packed:                                       # the upper triangle travels, the lower one is its mirror
            assert np.array_equal(results[r][0][:, iu[0], iu[1]], refS[:, iu[1], iu[0]])
            assert np.array_equal(results[r][0], results[r][0].transpose(0, 2, 1))
        else:
            assert np.array_equal(results[r][0], refS)   # every rank holds the full set, in global cell order
        assert np.array_equal(results[r][1], refg)


def _chunk_worker(rank, world, port, N, cd, fd, chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as o
        di = o.degrees(cd, fd)
        nf = 4 * di.fbs
        ex = ChunkedExchange(N, world, rank, condensed_per_cell(di.fbs), torch.device("cpu"), chunks)
        r0, r1 = row_partition(N, world, rank)
        covered = 0
        for k in range(ex.chunks):                       # the bench's loop: piece k computed, its gather started
            first, n = ex.piece_cells(k)
            assert first == covered
            covered += n
            Sb, gb = _condensed_blocks(N, cd, fd, r0 * N + first, n)
            S_view, g_view = ex.local_S_g(k, nf)
            S_view.copy_(torch.from_numpy(Sb))
            g_view.copy_(torch.from_numpy(gb))
            ex.exchange_async(k)
        assert covered == (r1 - r0) * N
        ex.wait()
        fullS = torch.cat([ex.gathered_S_g(r, nf)[0] for r in range(world)], dim=0).numpy()
        fullg = torch.cat([ex.gathered_S_g(r, nf)[1] for r in range(world)], dim=0).numpy()
        q.put((rank, (fullS, fullg)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,chunks", [(7, 3), (5, 8)])
def test_chunked_overlapped_exchange_matches_single_process(N, chunks):
    """the N > 1 step of bench.py: pieces of the local rows, asynchronous all_gather per piece (uneven pieces,
    more pieces asked for than rows available)"""
    cd, fd = 2, 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, N, cd, fd, chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    refS, refg = _condensed_blocks(N, cd, fd, 0, N * N)
    for r in range(world):
        assert np.array_equal(results[r][0], refS) and np.array_equal(results[r][1], refg)
