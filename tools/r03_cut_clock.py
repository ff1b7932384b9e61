"""The cut kernel of config 3 (512 x 512, k = 2): time per launch (HIP events) and, with PA_LIB = a tuning build and PA_CUT_CLOCK=1,
the clocks of block 0 per stage (printed by the library on stderr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proton_amd.batch import BatchAssembler
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
asm = BatchAssembler(0)
asm.cut_preprocess(N, refsteps=4)
out = asm.cut_local_ops(k, want=("lc", "rhs"))
torch.cuda.synchronize()
clock = os.environ.pop("PA_CUT_CLOCK", None)
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        asm.ctx.cut_local_ops(k, asm.level_set, 0, 1, 2, None, None, None, out["lc"].data_ptr(), out["rhs"].data_ptr(), None)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print("N %d k %d: %d cut cells, cut kernel %.4f ms per launch (best of 5 x 20: %.4f)" % (N, k, asm.ncut, sum(ts) / len(ts), min(ts)), flush=True)
