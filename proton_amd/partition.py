"""Multi-GPU plumbing of the hot path (host side): the block row partition of the structured mesh and the host-staged
twin of the one exchange of an assembly step.

The exchange itself lives behind the C ABI (include/proton_amd.h, pa_comm_*: RCCL send / recv of the packed top-face
rows of a slab's top cell row, one slab up).  `HostStagedHalo` moves the same buffers through torch.distributed's gloo
backend on host copies: it is what the CPU tests (world_size 2, no GPU) and the one-GPU rehearsal of bench.py use."""
import torch
import torch.distributed as dist


def row_partition(N, world, rank):
    """Cell rows [r0, r1) of rank `rank`: contiguous blocks, sizes differ by at most one row."""
    return (rank * N) // world, ((rank + 1) * N) // world


def cell_counts(Nx, Ny, world):
    return [(row_partition(Ny, world, r)[1] - row_partition(Ny, world, r)[0]) * Nx for r in range(world)]


def unpack_symmetric(Sp, nf):
    """[n, nf(nf+1)/2] column-packed upper triangles -> [n, nf, nf] symmetric matrices"""
    iu = torch.triu_indices(nf, nf, device=Sp.device)          # row-major pairs (i <= j)
    pos = iu[1] * (iu[1] + 1) // 2 + iu[0]
    S = torch.zeros((Sp.shape[0], nf, nf), dtype=Sp.dtype, device=Sp.device)
    S[:, iu[0], iu[1]] = Sp[:, pos]
    S[:, iu[1], iu[0]] = Sp[:, pos]
    return S


class HostStagedHalo:
    """pa_comm_halo_exchange_start + pa_comm_wait on host copies over gloo: rank r sends `send_up` (its
    pa_condensed_halo_pack output) to rank r + 1 and receives the rows of rank r - 1 into `recv_below`.  Blocking;
    non-blocking point-to-point requests so that the chain of ranks cannot deadlock."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def exchange_host(self, send_host, recv_host):
        """the same exchange on HOST tensors (pinned staging buffers of a caller that overlaps its device copies itself)"""
        reqs = []
        if recv_host is not None and self.rank > 0:
            reqs.append(dist.irecv(recv_host, src=self.rank - 1))
        if send_host is not None and self.rank + 1 < self.world:
            reqs.append(dist.isend(send_host, dst=self.rank + 1))
        for r in reqs:
            r.wait()

    def __call__(self, send_up, recv_below):
        reqs, staged = [], None
        if recv_below is not None and self.rank > 0:
            staged = torch.empty(recv_below.shape, dtype=recv_below.dtype)
            reqs.append(dist.irecv(staged, src=self.rank - 1))
        if send_up is not None and self.rank + 1 < self.world:
            reqs.append(dist.isend(send_up.detach().cpu().contiguous(), dst=self.rank + 1))
        for r in reqs:
            r.wait()
        if staged is not None:
            recv_below.copy_(staged)
