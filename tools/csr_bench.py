"""Timing of the assembly tail on the device: triplets + CSR build (pa_csr_from_triplets) for one workload."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proton_amd as pa
from proton_amd.batch import BatchAssembler

N, cd, fd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
asm = BatchAssembler(0)
asm.generate_mesh(N, N)
lc = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS, pa.QUAD_TENSOR)
g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
info = asm.assembler_info(cd, fd)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r, c, v, rr, rv = asm.triplets(cd, fd, lc, rhs, g)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rowptr, colind, values = asm.csr_from_triplets(r, c, v, info.system_size)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("N %d (%d,%d): %d slots, system %d, nnz %d: triplets %.2f ms, csr %.2f ms" %
          (N, cd, fd, r.numel(), info.system_size, colind.numel(), (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
