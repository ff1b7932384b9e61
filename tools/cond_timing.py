"""Kernel time of the static condensation next to the local-operator kernel (one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proton_amd as pa
from proton_amd.batch import BatchAssembler
asm = BatchAssembler(0)
for (N, cd, fd) in ((1024, 3, 2), (1024, 2, 1), (1024, 4, 3)):
    asm.generate_mesh(N, N)
    di, _ = pa.degree_info(cd, fd)
    lc = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    n = lc.shape[0]; nf = 4 * (fd + 1)
    Sp = torch.empty((n, nf * (nf + 1) // 2), dtype=torch.float64, device=asm.device)
    g = torch.empty((n, nf), dtype=torch.float64, device=asm.device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for it in range(3):
        ev[0].record(); asm.ctx.local_ops(di, pa.QUAD_TENSOR, pa.STAB_FANCY, 0, n, None, None, None, lc.data_ptr(), None); ev[1].record()
        ev[2].record(); asm.ctx.static_condensation_packed(di, n, lc.data_ptr(), rhs.data_ptr(), Sp.data_ptr(), g.data_ptr(), None); ev[3].record()
        torch.cuda.synchronize()
    print("N %d (%d,%d): local_ops %.3f ms, condensation %.3f ms" % (N, cd, fd, ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])), flush=True)
