"""cut_interface_kernel (cuthho_square -i) at 512 x 512, k = 2: time per launch of the cut cells' two-sided operators"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proton_amd import capi
from proton_amd.batch import BatchAssembler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
asm = BatchAssembler(0)
asm.cut_preprocess(N, refsteps=4)
parms = capi.InterfaceParams(1.0, 7.5, 5.0)
cbs = (k + 3) * (k + 2) // 2
ms = cbs + 4 * (k + 1)
n = asm.ncut
f64 = dict(dtype=torch.float64, device=asm.device)
lc = torch.empty((n, 2 * ms, 2 * ms), **f64); rhs = torch.empty((n, 2 * cbs), **f64)
info = torch.empty(n, dtype=torch.int32, device=asm.device)
def run():
    asm.ctx.cut_interface_ops(k, asm.level_set, parms, 1, None, None, lc.data_ptr(), rhs.data_ptr(), info.data_ptr())
run(); torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print("N %d k %d: %d cut cells, interface operators (interface kernel + both sides' stabilization + scatter) %.4f ms per call (best %.4f)" % (N, k, n, sum(ts) / len(ts), min(ts)), flush=True)
