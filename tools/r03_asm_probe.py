"""direct CSR of the assembler's system at growing sizes, with progress lines (gpurun_out/r03_asm_probe.log)"""
import sys, time, torch
sys.path.insert(0, ".")
import proton_amd as pa
from proton_amd.batch import BatchAssembler
log = open("gpurun_out/r03_asm_probe.log", "a")
def say(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
asm = BatchAssembler(0)
for N, cd, fd in ((256, 3, 2), (512, 3, 2), (1024, 2, 1), (1024, 3, 2)):
    asm.generate_mesh(N, N)
    say("mesh", N, cd, fd)
    lc = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    asm.synchronize(); say(" ops done")
    t0 = time.time(); rp, ci = asm.assembler_csr_pattern(cd, fd); asm.synchronize(); say(" pattern %.3f s nnz %d" % (time.time() - t0, ci.numel()))
    va = torch.empty(ci.numel(), dtype=torch.float64, device=asm.device); b = torch.empty(rp.numel() - 1, dtype=torch.float64, device=asm.device)
    for rep in range(3):
        t0 = time.time(); asm.assembler_csr_fill(cd, fd, lc, rhs, g, values=va, RHS=b); asm.synchronize(); say(" fill %.3f ms" % ((time.time() - t0) * 1e3))
    say(" sums", float(va.sum()), float(b.sum()))
    del lc, rp, ci, va, b
