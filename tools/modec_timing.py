"""Kernel times of the condensed mode next to mode L on one GPU (HIP events, best of a few):
  L: pre-pass + local-operator kernel (lc to HBM);  L+SC: plus the stand-alone condensation kernel;
  C: pre-pass + fused condensed kernel;  fill: direct CSR fill of the face-only system."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proton_amd as pa
from proton_amd.batch import BatchAssembler
asm = BatchAssembler(0)
cases = [(1024, 3, 2), (1024, 2, 1), (1024, 4, 3), (512, 0, 1)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (N, cd, fd) in cases:
    lo, hi = ((-1.0, -1.0), (1.0, 1.0)) if cd == 0 else ((0.0, 0.0), (1.0, 1.0))
    asm.generate_mesh(N, N, lo, hi)
    di, _ = pa.degree_info(cd, fd)
    n = N * N; nf = 4 * (fd + 1)
    lc = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    Sp = torch.empty((n, nf * (nf + 1) // 2), dtype=torch.float64, device=asm.device)
    gg = torch.empty((n, nf), dtype=torch.float64, device=asm.device)
    rec = asm.condensed_ops(cd, fd, rhs=rhs)
    rp, ci = asm.condensed_csr_pattern(cd, fd)
    vals, b = asm.condensed_csr_fill(cd, fd, rec, g)
    def t(f, reps=5):
        best = 1e9
        for _ in range(reps):
            a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); c.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(c))
        return best
    tL = t(lambda: asm.ctx.local_ops(di, pa.QUAD_TENSOR, pa.STAB_FANCY, 0, n, None, None, None, lc.data_ptr(), None))
    tS = t(lambda: asm.ctx.static_condensation_packed(di, n, lc.data_ptr(), rhs.data_ptr(), Sp.data_ptr(), gg.data_ptr(), None))
    tC = t(lambda: asm.ctx.condensed_ops(di, pa.QUAD_TENSOR, pa.STAB_FANCY, 0, n, rhs.data_ptr(), rec.data_ptr(), None))
    tF = t(lambda: asm.ctx.condensed_csr_fill(di, rec.data_ptr(), g.data_ptr(), None, vals.data_ptr(), b.data_ptr()))
    tR = t(lambda: asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR, out=rhs))
    print("N %d (%d,%d): L %.3f ms | stand-alone condensation %.3f | C (fused) %.3f | csr fill %.3f (nnz %d) | rhs %.3f"
          % (N, cd, fd, tL, tS, tC, tF, vals.numel(), tR), flush=True)
