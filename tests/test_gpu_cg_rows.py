"""pa_conjugated_gradient_rows: the reference's conjugated_gradient (solver_cg.hpp:45-144) on the row-partitioned face-only
system, every rank solving the rows it assembled.  The reference has one process: what pins this is (a) the one-rank call
being pa_conjugated_gradient bit for bit and (b) several ranks -- here: threads of one process, each with its own context and
its own slab of the mesh, behind an in-process transport with the semantics of the RCCL one -- reproducing the whole-mesh
solve."""
import threading

import pytest

pytestmark = pytest.mark.gpu


def _slab_system(N, cd, fd, rows, halo):
    import proton_amd as pa
    from proton_amd.batch import BatchAssembler
    a = BatchAssembler(0)
    a.generate_mesh(N, N, rows=rows)
    rhs = a.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = a.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    rec = a.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
    info = a.condensed_info(cd, fd)
    rp, ci = a.condensed_csr_pattern(cd, fd)
    va, b = a.condensed_csr_fill(cd, fd, rec, g, halo_below=halo)
    out = a.condensed_halo_pack(cd, fd, rec, g).clone() if rows[1] < N else None
    a.synchronize()
    return a, info, rp, ci, va, b, out


class ThreadTransport:
    """pa_cg_transport for R ranks that are threads of this process: mailboxes and barriers instead of messages"""

    def __init__(self, R, fail_rank=None, fail_at=None):
        """fail_rank / fail_at: that rank's halo callback returns an error at its fail_at-th call (a one-sided failure)"""
        self.R = R
        self.fail_rank, self.fail_at, self.calls = fail_rank, fail_at, 0
        self.bar = threading.Barrier(R)
        self.vals = [None] * R
        self.need = [None] * R
        self.mail = {}

    def make(self, r, ctx):
        import torch
        from proton_amd import capi
        R, me = self.R, self

        def allreduce(user, vals, n):
            me.vals[r] = [vals[i] for i in range(n)]
            me.bar.wait()
            tot = [sum(me.vals[q][i] for q in range(R)) for i in range(n)]      # rank order: the same bits on every rank
            me.bar.wait()
            for i in range(n):
                vals[i] = tot[i]
            return 0

        def counts(user, need_lo, need_hi, give_lo, give_hi):
            me.need[r] = (need_lo, need_hi)
            me.bar.wait()
            give_lo[0] = me.need[r - 1][1] if r > 0 else 0          # rank - 1 reads above its range: my first entries
            give_hi[0] = me.need[r + 1][0] if r + 1 < R else 0      # rank + 1 reads below its range: my last entries
            me.bar.wait()
            return 0

        def d2h(ptr, n):
            t = torch.empty(n, dtype=torch.float64)
            ctx.copy_to_host(t.data_ptr(), ptr, 8 * n)
            return t

        def halo(user, send_lo, n_send_lo, send_hi, n_send_hi, recv_lo, n_recv_lo, recv_hi, n_recv_hi, stream):
            if r == me.fail_rank:
                me.calls += 1
                if me.calls == me.fail_at:
                    # this rank's exchange fails before it posts anything; like an asynchronous send / recv the neighbours'
                    # exchanges still return (their data is stale -- the solver must not use it: everybody leaves at the reduction)
                    me.bar.wait(); me.bar.wait()
                    return 1
            ctx.synchronize()
            if r > 0 and n_send_lo:
                me.mail[(r, r - 1)] = d2h(send_lo, n_send_lo)
            if r + 1 < R and n_send_hi:
                me.mail[(r, r + 1)] = d2h(send_hi, n_send_hi)
            me.bar.wait()
            if r > 0 and n_recv_lo:
                t = me.mail.get((r - 1, r))                 # (after a neighbour's failure: its previous message, or none yet)
                if t is not None:
                    assert t.numel() == n_recv_lo
                    ctx.copy_to_device(recv_lo, t.data_ptr(), 8 * n_recv_lo)
            if r + 1 < R and n_recv_hi:
                t = me.mail.get((r + 1, r))
                if t is not None:
                    assert t.numel() == n_recv_hi
                    ctx.copy_to_device(recv_hi, t.data_ptr(), 8 * n_recv_hi)
            me.bar.wait()
            return 0

        cbs = (capi.CG_ALLREDUCE(allreduce), capi.CG_HALO(halo), capi.CG_COUNTS(counts))
        return capi.CgTransport(None, *cbs), cbs


@pytest.mark.parametrize("N,cd,fd", [(12, 2, 1), (8, 3, 2)])
def test_one_rank_equals_conjugated_gradient(N, cd, fd):
    import torch
    a, info, rp, ci, va, b, _ = _slab_system(N, cd, fd, (0, N), None)
    n = b.numel()
    x0, x1 = torch.zeros_like(b), torch.zeros_like(b)
    r0 = a.ctx.conjugated_gradient(n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr(), x0.data_ptr(), tol=1e-10, max_iter=5000)
    r1 = a.ctx.conjugated_gradient_rows(None, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr(), x1.data_ptr(), tol=1e-10,
                                        max_iter=5000)
    a.synchronize()
    assert r0[0] == 0 and r1[:2] == r0[:2] and r1[2] == r0[2]
    assert torch.equal(x0, x1)


@pytest.mark.parametrize("N,cd,fd,parts", [(12, 2, 1, (0, 5, 12)), (9, 3, 2, (0, 2, 5, 9)), (8, 4, 3, (0, 1, 4, 8)), (4, 2, 1, (0, 1, 2, 3, 4))])
def test_ranks_reproduce_the_whole_mesh_solve(N, cd, fd, parts):
    import torch
    a, info, rp, ci, va, b, _ = _slab_system(N, cd, fd, (0, N), None)
    n = b.numel()
    # (the assembled right-hand side of the sin sin problem is close to an eigenvector of the preconditioned operator on
    # this uniform mesh -- CG is done in 1-3 steps; a random one takes hundreds)
    gen = torch.Generator().manual_seed(1234 + N)
    b = (b.cpu() + torch.rand(n, generator=gen, dtype=torch.float64) - 0.5).to(a.device)
    xw = torch.zeros_like(b)
    rw = a.ctx.conjugated_gradient(n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr(), xw.data_ptr(), tol=1e-11, max_iter=10000)
    a.synchronize()
    assert rw[0] == 0 and (rw[1] > 30 or n < 200)
    slabs, halo = [], None
    for r0, r1 in zip(parts[:-1], parts[1:]):
        s = _slab_system(N, cd, fd, (r0, r1), halo)
        halo = s[6]
        slabs.append(s)
    R = len(slabs)
    tt = ThreadTransport(R)
    out, keep = [None] * R, []

    def run(r):
        s, inf, rps, cis, vs, bs, _ = slabs[r]
        tp, cbs = tt.make(r, s.ctx)
        keep.append(cbs)
        bs = b[inf.row_begin:inf.row_end].clone()           # this rank's rows of the common right-hand side
        x = torch.zeros_like(bs)
        try:
            res = s.ctx.conjugated_gradient_rows(tp, inf.row_begin, inf.row_end, rps.data_ptr(), cis.data_ptr(), vs.data_ptr(), bs.data_ptr(),
                                                 x.data_ptr(), tol=1e-11, max_iter=10000)
            s.synchronize()
            out[r] = (res, x.cpu())
        except BaseException as e:      # noqa: BLE001  (a failed rank must not leave the others at a barrier)
            out[r] = e
            tt.bar.abort()

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert all(not t.is_alive() for t in th)
    for o in out:
        assert not isinstance(o, BaseException), o
    assert all(o[0][0] == 0 for o in out)                      # every rank converged ...
    assert len({o[0][1] for o in out}) == 1                   # ... in the same iteration
    assert abs(out[0][0][1] - rw[1]) <= 3                     # (the sums are grouped by rank: a step more or less at the threshold)
    x = torch.cat([o[1] for o in out])
    assert x.numel() == n
    scale = float(xw.abs().max())
    assert float((x - xw.cpu()).abs().max()) < 1e-7 * scale


@pytest.mark.parametrize("fail_rank,fail_at", [(1, 3), (0, 1), (2, 7)])
def test_one_sided_failure_ends_the_solve_on_every_rank(fail_rank, fail_at):
    """One rank's neighbour exchange fails in iteration fail_at: that rank returns status 1 (PA_ERR_COMM), the others status 3
    ("another rank failed"), all from the same reduction -- nobody is left waiting in a collective."""
    import torch
    from proton_amd.capi import ProtonAmdError
    N, cd, fd, parts = 12, 2, 1, (0, 4, 8, 12)
    slabs, halo = [], None
    for r0, r1 in zip(parts[:-1], parts[1:]):
        s = _slab_system(N, cd, fd, (r0, r1), halo)
        halo = s[6]
        slabs.append(s)
    R = len(slabs)
    tt = ThreadTransport(R, fail_rank, fail_at)
    out, keep = [None] * R, []
    gen = torch.Generator().manual_seed(99)

    def run(r):
        s, inf, rps, cis, vs, bs, _ = slabs[r]
        tp, cbs = tt.make(r, s.ctx)
        keep.append(cbs)
        bs = (bs.cpu() + torch.rand(bs.numel(), generator=torch.Generator().manual_seed(r), dtype=torch.float64) - 0.5).to(s.device)
        x = torch.zeros_like(bs)
        try:
            s.ctx.conjugated_gradient_rows(tp, inf.row_begin, inf.row_end, rps.data_ptr(), cis.data_ptr(), vs.data_ptr(), bs.data_ptr(),
                                           x.data_ptr(), tol=1e-13, max_iter=10000)
            out[r] = "converged"
        except ProtonAmdError as e:
            out[r] = (e.status, str(e))
        except BaseException as e:      # noqa: BLE001
            out[r] = e
            tt.bar.abort()

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert all(not t.is_alive() for t in th), "a rank is still inside the solve"
    for r, o in enumerate(out):
        assert isinstance(o, tuple), (r, o)
        assert o[0] == 7, o                                              # PA_ERR_COMM on every rank
        assert ("transport status 1" in o[1]) == (r == fail_rank) and ("transport status 3" in o[1]) == (r != fail_rank), (r, o)


def test_rccl_transport_of_one_rank():
    """pa_comm_cg_transport over a communicator of one rank (the only RCCL configuration one GPU allows): its callbacks are
    taken -- no neighbour, sums unchanged -- and the solve is the one-rank solve."""
    import torch
    from proton_amd import capi
    N, cd, fd = 10, 2, 1
    a, info, rp, ci, va, b, _ = _slab_system(N, cd, fd, (0, N), None)
    n = b.numel()
    comm = capi.Comm(a.ctx, 1, 0, capi.comm_unique_id())
    tp = comm.cg_transport()
    x0, x1 = torch.zeros_like(b), torch.zeros_like(b)
    r0 = a.ctx.conjugated_gradient_rows(None, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr(), x0.data_ptr(), tol=1e-10, max_iter=5000)
    r1 = a.ctx.conjugated_gradient_rows(tp, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr(), x1.data_ptr(), tol=1e-10, max_iter=5000)
    a.synchronize()
    assert r0[0] == 0 and r1 == r0 and torch.equal(x0, x1)
    src = torch.arange(8, dtype=torch.float64, device=a.device)
    dst = torch.zeros_like(src)
    comm.neighbour_exchange_start(src.data_ptr(), 4, src.data_ptr(), 4, dst.data_ptr(), 4, dst.data_ptr(), 4)      # no neighbours: nothing moves
    comm.wait()
    a.synchronize()
    assert float(dst.abs().max()) == 0.0
    comm.close()


def test_rows_reading_beyond_the_range_need_a_transport():
    """a slab's rows read the slab below: without a transport the call refuses (transport_status 2), it does not solve a
    truncated system"""
    import torch
    from proton_amd.capi import ProtonAmdError
    N, cd, fd = 8, 2, 1
    lower = _slab_system(N, cd, fd, (0, 4), None)
    s, inf, rps, cis, vs, bs, _ = _slab_system(N, cd, fd, (4, N), lower[6])
    x = torch.zeros_like(bs)
    with pytest.raises(ProtonAmdError) as e:
        s.ctx.conjugated_gradient_rows(None, inf.row_begin, inf.row_end, rps.data_ptr(), cis.data_ptr(), vs.data_ptr(), bs.data_ptr(), x.data_ptr())
    assert e.value.status == 1      # PA_ERR_INVALID_ARG
