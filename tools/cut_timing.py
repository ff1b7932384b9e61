"""Wall times of the host-side cutHHO steps of the product (preprocessing, quadrature lists) next to the kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import proton_amd as pa
from proton_amd.batch import BatchAssembler
asm = BatchAssembler(0)
for N in (256, 512, 1024, 2048):
    t0 = time.perf_counter(); asm.cut_preprocess(N, refsteps=4); torch.cuda.synchronize(); t1 = time.perf_counter()
    out = asm.cut_local_ops(2); torch.cuda.synchronize(); t2 = time.perf_counter()      # builds + uploads the lists, then the kernel
    out = asm.cut_local_ops(2); torch.cuda.synchronize(); t3 = time.perf_counter()      # kernel only
    print("N %d: cut cells %d, preprocess %.1f ms, lists+upload+kernel %.1f ms, kernel %.2f ms" %
          (N, asm.ncut, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
