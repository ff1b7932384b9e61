"""Dump the cell of the general-quadrilateral headline mesh where GPU and oracle differ most (k = 3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, oracle_lib as o, proton_amd as pa
from proton_amd.batch import BatchAssembler
asm = BatchAssembler(0)
N, cd, fd = 1024, 4, 3
w = dict(bench.WORKLOADS["quad1024_k2_general"])
pts_d, ids_d = bench.general_quad_mesh(torch, N, w["lo"], w["hi"], w["perturb"], asm.device)
asm.ctx.mesh_attach_device(pts_d.data_ptr(), (N + 1) * (N + 1), ids_d.data_ptr(), N * N)
lc = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
asm.synchronize()
points, ptids = bench.workload_mesh(w)
di = o.degrees(cd, fd)
rng = np.random.default_rng(2026)
cells = np.sort(rng.choice(N * N, size=1024, replace=False))
got = lc[torch.from_numpy(cells).to(lc.device)].cpu().numpy().transpose(0, 2, 1)
errs = []
refs = []
for i, c in enumerate(cells):
    st, r = o.local_ops_batch(points, ptids, di, pa.QUAD_TENSOR, pa.STAB_FANCY, first=int(c), n=1, want=("lc",))
    refs.append(r["lc"][0])
    errs.append(np.abs(got[i] - r["lc"][0]).max() / np.abs(r["lc"][0]).max())
errs = np.array(errs)
k = int(errs.argmax())
print("worst", errs[k], "cell", cells[k], "median", np.median(errs), "p99", np.quantile(errs, 0.99))
np.savez(os.path.join(ROOT, "gpurun_out", "worst_cell.npz"), gpu=got[k], oracle=refs[k], pts=points[ptids[cells[k]].astype(np.int64)],
         ids=ptids[cells[k]], errs=errs)
