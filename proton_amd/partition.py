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


class HostStagedAllgather:
    """pa_comm_allgather_start + pa_comm_wait on host copies over gloo -- the north star's literal collective: every rank
    contributes `count` doubles (its owned CSR values / right-hand side, padded to the largest rank's count) and ends with
    all ranks' pieces, rank r's at [r * count, (r + 1) * count).  Blocking."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def __call__(self, send, recv, count):
        assert recv.numel() >= self.world * count
        mine = torch.zeros(count, dtype=send.dtype)
        mine[:min(count, send.numel())] = send.detach().reshape(-1)[:count].cpu()
        parts = [torch.empty(count, dtype=send.dtype) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        recv.reshape(-1)[:self.world * count].copy_(torch.cat(parts))


class HostStagedCgTransport:
    """The transport of pa_conjugated_gradient_rows (include/proton_amd.h: pa_cg_transport) on host copies over gloo -- the twin
    of pa_comm_cg_transport for the tests and the one-GPU rehearsal: ranks in slab order, neighbours r - 1 and r + 1."""

    def __init__(self, rank, world, ctx):
        from . import capi
        self.rank, self.world = rank, world

        def allreduce(user, vals, n):
            try:
                t = torch.tensor([vals[i] for i in range(n)], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                for i in range(n):
                    vals[i] = float(t[i])
                return 0
            except Exception:       # noqa: BLE001  (a callback must not raise through C)
                return 1

        def dev_to_host(ptr, n):
            out = torch.empty(n, dtype=torch.float64)
            ctx.copy_to_host(out.data_ptr(), ptr, 8 * n)
            return out

        def host_to_dev(ptr, t):
            ctx.copy_to_device(ptr, t.data_ptr(), 8 * t.numel())

        def halo(user, send_lo, n_send_lo, send_hi, n_send_hi, recv_lo, n_recv_lo, recv_hi, n_recv_hi, stream):
            try:
                ctx.synchronize()
                reqs, got_lo, got_hi = [], None, None
                lo, hi = self.rank > 0, self.rank + 1 < self.world
                if lo and n_recv_lo:
                    got_lo = torch.empty(n_recv_lo, dtype=torch.float64)
                    reqs.append(dist.irecv(got_lo, src=self.rank - 1))
                if hi and n_recv_hi:
                    got_hi = torch.empty(n_recv_hi, dtype=torch.float64)
                    reqs.append(dist.irecv(got_hi, src=self.rank + 1))
                if lo and n_send_lo:
                    reqs.append(dist.isend(dev_to_host(send_lo, n_send_lo), dst=self.rank - 1))
                if hi and n_send_hi:
                    reqs.append(dist.isend(dev_to_host(send_hi, n_send_hi), dst=self.rank + 1))
                for r in reqs:
                    r.wait()
                if got_lo is not None:
                    host_to_dev(recv_lo, got_lo)
                if got_hi is not None:
                    host_to_dev(recv_hi, got_hi)
                return 0
            except Exception:       # noqa: BLE001
                return 1

        def counts(user, need_lo, need_hi, give_lo, give_hi):
            try:
                give_lo[0] = 0
                give_hi[0] = 0
                reqs, from_lo, from_hi = [], None, None
                lo, hi = self.rank > 0, self.rank + 1 < self.world
                if lo:
                    from_lo = torch.zeros(1, dtype=torch.int64)
                    reqs.append(dist.irecv(from_lo, src=self.rank - 1))
                    reqs.append(dist.isend(torch.tensor([need_lo], dtype=torch.int64), dst=self.rank - 1))
                if hi:
                    from_hi = torch.zeros(1, dtype=torch.int64)
                    reqs.append(dist.irecv(from_hi, src=self.rank + 1))
                    reqs.append(dist.isend(torch.tensor([need_hi], dtype=torch.int64), dst=self.rank + 1))
                for r in reqs:
                    r.wait()
                if lo:
                    give_lo[0] = int(from_lo[0])      # rank - 1 reads that many of my first entries (its need_hi)
                if hi:
                    give_hi[0] = int(from_hi[0])      # rank + 1 reads that many of my last entries (its need_lo)
                return 0
            except Exception:       # noqa: BLE001
                return 1

        self._cbs = (capi.CG_ALLREDUCE(allreduce), capi.CG_HALO(halo), capi.CG_COUNTS(counts))      # keep the thunks alive
        self.struct = capi.CgTransport(None, *self._cbs)
