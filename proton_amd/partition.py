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
