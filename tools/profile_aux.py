"""One launch set of every kernel DESIGN.md quotes outside the bench step, for `rocprofv3 --kernel-trace --stats`:
stand-alone static condensation (k = 1, 2, 3), triplets, device CSR build, condensed triplets / CSR pattern / fill,
Jacobi-PCG, take_local_data, dirichlet data, obstacle tables / triplets -- at the headline sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proton_amd as pa
from proton_amd.batch import BatchAssembler
asm = BatchAssembler(0)
REPS = 3
for (N, cd, fd) in ((1024, 2, 1), (1024, 3, 2), (1024, 4, 3)):
    asm.generate_mesh(N, N)
    di, _ = pa.degree_info(cd, fd)
    n, nf = N * N, 4 * (fd + 1)
    lc = asm.local_ops(cd, fd, want=("lc",))["lc"]
    rhs = asm.cell_rhs(cd, pa.capi.FN_SIN_SIN_RHS)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    Sp = torch.empty((n, nf * (nf + 1) // 2), dtype=torch.float64, device=asm.device)
    gg = torch.empty((n, nf), dtype=torch.float64, device=asm.device)
    for _ in range(REPS):
        asm.ctx.static_condensation_packed(di, n, lc.data_ptr(), rhs.data_ptr(), Sp.data_ptr(), gg.data_ptr(), None)
    rec = asm.condensed_ops(cd, fd, rhs=rhs)
    for _ in range(REPS):
        asm.condensed_csr_fill(cd, fd, rec, g)
    if (cd, fd) == (3, 2):
        for _ in range(REPS):
            trip = asm.triplets(cd, fd, lc, rhs, g)
        info = asm.assembler_info(cd, fd)
        rowptr, colind, values = asm.csr_from_triplets(trip[0], trip[1], trip[2], info.system_size)
        del trip
        loc = asm.take_local_data(cd, fd, torch.zeros(info.system_size, dtype=torch.float64, device=asm.device), g)
        del rowptr, colind, values, loc
        ct = asm.condensed_triplets(cd, fd, rec, g)
        del ct
        rp, ci = asm.condensed_csr_pattern(cd, fd)
        va, b = asm.condensed_csr_fill(cd, fd, rec, g)
        x, reason, iters, rr = asm.conjugated_gradient(rp, ci, va, b.contiguous(), tol=1e-30, max_iter=30)
        uF = asm.condensed_take_faces(cd, fd, x, g)
        uT = asm.condensed_recover(cd, fd, uF, rhs=rhs)
    torch.cuda.synchronize()
    del lc, rec
# obstacle assembler at config 4's size
N = 512
asm.generate_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
lc = asm.local_ops(0, 1, want=("lc",))["lc"]
rhs = asm.cell_rhs(0, pa.capi.FN_OBSTACLE_RHS, dinc=1)
g = asm.dirichlet_data(1, pa.capi.FN_OBSTACLE_SOL)
in_A = (torch.arange(N * N, device=asm.device) % 3 == 0).to(torch.uint8)
A_ct, B_ct, num_I, num_A = asm.obstacle_tables(in_A)
gamma = torch.zeros(N * N, dtype=torch.float64, device=asm.device)
for _ in range(REPS):
    asm.obstacle_triplets(0, 1, lc, rhs, g, gamma, in_A, A_ct, B_ct, num_I)
torch.cuda.synchronize()
print("aux kernels launched")
