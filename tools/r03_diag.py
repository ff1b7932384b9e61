import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import proton_amd as pa
from proton_amd.batch import BatchAssembler
import test_gpu_condensed as T
if len(sys.argv) > 1 and sys.argv[1] == "config5":
    T.test_config5_slabs_equal_whole_mesh_at_full_size()
    print("config5 done; torch reserved %.1f GB allocated %.1f GB" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9), flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "empty":
    torch.cuda.empty_cache()
    print("emptied; reserved %.1f GB" % (torch.cuda.memory_reserved() / 1e9))
b = BatchAssembler(0)
N = 1024
b.generate_mesh(N, N)
for rep in range(3):
    out = b.local_ops(3, 2, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc", "info"))
    b.synchronize()
    lc = out["lc"]
    one = torch.zeros(22, dtype=torch.float64, device=lc.device)
    one[0] = 1.0
    for f in range(4):
        one[10 + 3 * f] = 1.0
    r = (lc @ one).abs().amax(dim=1) / lc.abs().amax(dim=(1, 2))
    bad = torch.nonzero(r > 1e-11).flatten()
    if rep == 0:
        for seg in torch.cuda.memory_snapshot():
            if seg["address"] <= lc.data_ptr() < seg["address"] + seg["total_size"]:
                print("lc lives in segment %x size %.2f GB" % (seg["address"], seg["total_size"] / 1e9), "blocks", [(bl["size"], bl["state"]) for bl in seg["blocks"]][:6])
    print("rep", rep, "lc ptr %x" % lc.data_ptr(), "bad cells:", bad.numel(), (bad[:3].tolist(), bad[-3:].tolist()) if bad.numel() else "", flush=True)
    if bad.numel():
        torch.cuda.synchronize()
        for lo, hi in ((285900, 286100), (1048476, 1048576)):
            h = lc[lo:hi].cpu()
            rc = (h @ one.cpu()).abs().amax(dim=1) / h.abs().amax(dim=(1, 2))
            rg = r[lo:hi].cpu()
            r2 = ((lc[lo:hi].clone() @ one).abs().amax(dim=1) / lc[lo:hi].abs().amax(dim=(1, 2))).cpu()
            print("   cells %d..%d: CPU check max %.2e | GPU (whole-array matmul) max %.2e | GPU (slice matmul) max %.2e" % (lo, hi, float(rc.max()), float(rg.max()), float(r2.max())))
        c = int(bad[0]) + 5
        m_gpu = lc[c].clone()
        m_cpu = lc[c].cpu()
        print("   cell", c, "gpu-read vs cpu copy equal:", bool(torch.equal(m_gpu.cpu(), m_cpu)), "row0", m_cpu[0, :4].tolist(), "good cell row0", lc[0].cpu()[0, :4].tolist())
        # the same cells again, only them, through first/n
        o2 = b.local_ops(3, 2, pa.QUAD_TENSOR, pa.STAB_FANCY, first=c, n=8, want=("lc",))
        b.synchronize()
        print("   recomputed alone (first=%d, n=8): equal to the good cell:" % c, float((o2["lc"][0] - lc[0]).abs().max()), "equal to the bad one:", float((o2["lc"][0] - lc[c]).abs().max()))
    del out, lc
