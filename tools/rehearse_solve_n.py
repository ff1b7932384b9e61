#!/usr/bin/env python3
"""The whole condensed pipeline on R ranks -- R processes sharing the visible GPU, host-staged gloo transports
(proton_amd.partition: HostStagedHalo, HostStagedCgTransport): every rank assembles the face-only rows of its slab (halo rows
from the slab below), solves them where they are (pa_conjugated_gradient_rows), takes the face unknowns of its cells (its own
rows plus the first band of the slab above) and recovers its cell unknowns; rank 0 also runs the whole mesh on its own and
compares the face solution and the cell unknowns.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/rehearse_solve_n.py [N cd fd]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import proton_amd as pa
from proton_amd.batch import BatchAssembler
from proton_amd.partition import HostStagedCgTransport, HostStagedHalo, row_partition

N, cd, fd = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 3, 2)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)


def system(rows, halo_fn):
    a = BatchAssembler(0)
    a.generate_mesh(N, N, rows=rows)
    rhs = a.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = a.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    rec = a.condensed_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, rhs=rhs)
    info = a.condensed_info(cd, fd)
    halo_in = halo_fn(a, info, rec, g) if halo_fn else None
    rp, ci = a.condensed_csr_pattern(cd, fd)
    va, b = a.condensed_csr_fill(cd, fd, rec, g, halo_below=halo_in)
    a.synchronize()
    return a, info, rp, ci, va, b, rhs, g


def exchange(a, info, rec, g):
    out = a.condensed_halo_pack(cd, fd, rec, g) if info.halo_cells else None
    inn = torch.zeros((N, info.halo_doubles), dtype=torch.float64, device=a.device) if info.has_below else None
    a.synchronize()
    HostStagedHalo(rank, world)(out, inn)
    return inn


a, info, rp, ci, va, b, rhs, g = system(row_partition(N, world, rank), exchange)
tp = HostStagedCgTransport(rank, world, a.ctx)
x = torch.zeros_like(b)
res = a.ctx.conjugated_gradient_rows(tp.struct, info.row_begin, info.row_end, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr(),
                                     x.data_ptr(), tol=1e-10, max_iter=20000)
a.synchronize()
# the face unknowns of this slab's cells: its own rows and, for the top faces of its top cell row, the first band of the
# slab above (at most one mesh row of faces)
band = (2 * N + 1) * (fd + 1)
xfull = torch.zeros(info.system_size, dtype=torch.float64, device=a.device)
xfull[info.row_begin:info.row_end] = x
reqs, up = [], None
if rank + 1 < world:
    up = torch.zeros(band, dtype=torch.float64)
    reqs.append(dist.irecv(up, src=rank + 1))
if rank > 0:
    first = torch.zeros(band, dtype=torch.float64)
    m = min(band, x.numel())
    first[:m] = x[:m].cpu()
    reqs.append(dist.isend(first, dst=rank - 1))
for r in reqs:
    r.wait()
if up is not None:
    m = min(band, info.system_size - info.row_end)
    xfull[info.row_end:info.row_end + m] = up[:m].to(a.device)
uF = a.condensed_take_faces(cd, fd, xfull, g)
uT = a.condensed_recover(cd, fd, uF, rhs=rhs)
a.synchronize()
parts = [None] * world
dist.all_gather_object(parts, (res, x.cpu(), uT.cpu()))
if rank == 0:
    w, winfo, wrp, wci, wva, wb, wrhs, wg = system((0, N), None)
    xw = torch.zeros_like(wb)
    rw = w.ctx.conjugated_gradient(wb.numel(), wrp.data_ptr(), wci.data_ptr(), wva.data_ptr(), wb.data_ptr(), xw.data_ptr(), tol=1e-10,
                                   max_iter=20000)
    w.synchronize()
    xs = torch.cat([p[1] for p in parts])
    err = float((xs - xw.cpu()).abs().max()) / float(xw.abs().max())
    wuT = w.condensed_recover(cd, fd, w.condensed_take_faces(cd, fd, xw, wg), rhs=wrhs)
    w.synchronize()
    uTs = torch.cat([p[2] for p in parts])
    errT = float((uTs - wuT.cpu()).abs().max()) / float(wuT.abs().max())
    print("ranks %d, %dx%d k=%d: unknowns %d, whole-mesh CG %s, per rank %s, max |x - x_whole| / max |x_whole| = %.2e, cell unknowns %.2e"
          % (world, N, N, fd, xs.numel(), rw, [p[0] for p in parts], err, errT))
    assert err < 1e-7 and errT < 1e-7 and all(p[0][0] == 0 for p in parts)
dist.barrier()
dist.destroy_process_group()
