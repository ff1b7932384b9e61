"""per-cell errors of the cut kernel against binary128, and the product's quadrature lists against the oracle's"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import oracle_lib as o
from proton_amd.batch import BatchAssembler, to_rowcol
def nerr(a, b): return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
asm = BatchAssembler(0)
for N, k, r in ((12, 2, 5), (64, 2, 4)):
    asm.cut_preprocess(N, refsteps=r)
    ref = o.CutMesh(N, refsteps=r)
    cut = np.nonzero(ref.cell_loc == o.CUT_ON_INTERFACE)[0]
    di = o.degrees(k + 1, k)
    out = asm.cut_local_ops(k); asm.synchronize()
    oper, data, stab = to_rowcol(out["oper"]), to_rowcol(out["data"]), to_rowcol(out["stab"])
    nd = 0
    for which, fn, deg in ((0, ref.cell_quadrature, 2 * (k + 1)), (1, ref.interface_quadrature, 2 * (k + 1))):
        off, xyw = asm.ctx.cut_quadrature_points(k, 0, which)
        for i, c in enumerate(cut):
            want = np.array(fn(int(c), deg, 0)).T.reshape(-1, 3)
            got = xyw[off[i]:off[i + 1]]
            if got.shape != want.shape or not np.array_equal(got, want):
                nd += 1
    print("N=%d k=%d r=%d: %d cut cells, quadrature lists differing from the oracle's: %d" % (N, k, r, len(cut), nd))
    rows = []
    for i, c in enumerate(cut):
        st, to_, td = ref.truth_laplacian(int(c), di)
        st, ts = ref.truth_stabilization(int(c), di)
        rows.append((ref.truth_cond(int(c), di), nerr(data[i], td), nerr(oper[i], to_), nerr(stab[i], ts), np.abs(data[i] - data[i].T).max() / np.abs(data[i]).max()))
    rows = np.array(rows)
    idx = np.argsort(-rows[:, 1])[:8]
    for j in idx:
        print("  cell %6d cond %.2e data %.2e oper %.2e stab %.2e asym %.2e  data/cond %.1e" % (cut[j], *rows[j], rows[j, 1] / rows[j, 0]))
