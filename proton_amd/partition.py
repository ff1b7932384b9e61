"""Multi-GPU plumbing of the hot path: block row partition of the structured mesh and the one
exchange step (all_gather of the condensed face-dof blocks before the host-side solve).
Backend-agnostic: `nccl` (= RCCL over xGMI) on GPUs, `gloo` in the CPU tests."""
import torch
import torch.distributed as dist


def row_partition(N, world, rank):
    """Cell rows [r0, r1) of rank `rank`: contiguous blocks, sizes differ by at most one row."""
    return (rank * N) // world, ((rank + 1) * N) // world


def cell_counts(Nx, Ny, world):
    return [(row_partition(Ny, world, r)[1] - row_partition(Ny, world, r)[0]) * Nx for r in range(world)]


def condensed_per_cell(fbs, packed=False):
    """values exchanged per cell: S (4 fbs)^2 -- or its upper triangle when packed -- plus g (4 fbs);
    values only, the indices are closed-form"""
    nf = 4 * fbs
    return (nf * (nf + 1) // 2 if packed else nf * nf) + nf


def unpack_symmetric(Sp, nf):
    """[n, nf(nf+1)/2] column-packed upper triangles -> [n, nf, nf] symmetric matrices"""
    iu = torch.triu_indices(nf, nf, device=Sp.device)          # row-major pairs (i <= j)
    pos = iu[1] * (iu[1] + 1) // 2 + iu[0]
    S = torch.zeros((Sp.shape[0], nf, nf), dtype=Sp.dtype, device=Sp.device)
    S[:, iu[0], iu[1]] = Sp[:, pos]
    S[:, iu[1], iu[0]] = Sp[:, pos]
    return S


class CondensedExchange:
    """Preallocated buffers for the per-step all_gather; ranks may own different numbers of cells
    (the send buffer is padded to the largest block, the result is compacted by views)."""

    def __init__(self, counts, per_cell, rank, device, dtype=torch.float64, host_staged=False):
        self.counts, self.per_cell, self.rank = list(counts), per_cell, rank
        self.world = len(counts)
        self.slot = max(counts) * per_cell
        self.send = torch.zeros(self.slot, dtype=dtype, device=device)
        self.recv = torch.empty(self.world * self.slot, dtype=dtype, device=device)
        # rehearsal mode (several ranks on ONE GPU cannot form an RCCL communicator): the collective
        # runs on host copies through gloo; the device buffers and their layout are the same
        self.host_staged = host_staged and self.send.is_cuda
        if self.host_staged:
            self.h_send = torch.empty(self.slot, dtype=dtype).pin_memory()
            self.h_recv = torch.empty(self.world * self.slot, dtype=dtype).pin_memory()

    def local_view(self):
        """where this rank writes its n_local * per_cell values before exchange()"""
        return self.send[: self.counts[self.rank] * self.per_cell]

    def _split(self, v, n, nf):
        ns = self.per_cell - nf                            # nf*nf, or nf(nf+1)/2 when packed
        S = v[: n * ns]
        return (S.view(n, nf, nf) if ns == nf * nf else S.view(n, ns)), v[n * ns:].view(n, nf)

    def local_S_g(self, nf):
        """the layout the condensation kernel writes: all S blocks ([n, nf, nf], or [n, nf(nf+1)/2]
        packed upper triangles), then all g [n, nf]"""
        return self._split(self.local_view(), self.counts[self.rank], nf)

    def gathered_S_g(self, r, nf):
        n = self.counts[r]
        return self._split(self.recv[r * self.slot: r * self.slot + n * self.per_cell], n, nf)

    def exchange(self):
        if self.world == 1:
            self.recv[: self.slot].copy_(self.send)
        elif self.host_staged:
            self.h_send.copy_(self.send)                   # synchronizes with the producing stream
            dist.all_gather_into_tensor(self.h_recv, self.h_send)
            self.recv.copy_(self.h_recv, non_blocking=True)
        else:
            dist.all_gather_into_tensor(self.recv, self.send)
        return self.recv

    def gathered(self, r):
        """view of rank r's block [counts[r], per_cell] after exchange()"""
        return self.recv[r * self.slot: r * self.slot + self.counts[r] * self.per_cell].view(self.counts[r], self.per_cell)


class ChunkedExchange:
    """The same exchange in `chunks` pieces of the local cell rows, so that the all_gather of one
    piece runs (asynchronously, on the collective's own stream) while the kernels of the next piece
    run.  Every piece is a CondensedExchange over the cells of that piece on every rank; pieces
    are contiguous row blocks of the rank's slab (row_partition of the local rows)."""

    def __init__(self, N, world, rank, per_cell, device, chunks, dtype=torch.float64, host_staged=False):
        self.N, self.world, self.rank, self.per_cell = N, world, rank, per_cell
        rows = [row_partition(N, world, r) for r in range(world)]
        self.chunks = max(1, min(chunks, min(b - a for a, b in rows)))
        # piece k of rank r: local rows [lo, hi) of its slab
        self.pieces = [[row_partition(b - a, self.chunks, k) for k in range(self.chunks)] for a, b in rows]
        self.ex = [CondensedExchange([(self.pieces[r][k][1] - self.pieces[r][k][0]) * N for r in range(world)], per_cell, rank,
                                     device, dtype, host_staged) for k in range(self.chunks)]
        self.pending = []

    def piece_cells(self, k):
        """(first local cell, number of cells) of piece k on this rank"""
        lo, hi = self.pieces[self.rank][k]
        return lo * self.N, (hi - lo) * self.N

    def local_S_g(self, k, nf):
        return self.ex[k].local_S_g(nf)

    def exchange_async(self, k):
        e = self.ex[k]
        if e.world == 1 or e.host_staged:
            e.exchange()
        else:
            self.pending.append(dist.all_gather_into_tensor(e.recv, e.send, async_op=True))

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def gathered_S_g(self, r, nf):
        """rank r's blocks in its local cell order: concatenation over the pieces (copies)"""
        parts = [self.ex[k].gathered_S_g(r, nf) for k in range(self.chunks)]
        return torch.cat([p[0] for p in parts], dim=0), torch.cat([p[1] for p in parts], dim=0)
