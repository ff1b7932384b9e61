"""The exact FP64 operation counts bench.py's roofline_fp64 uses (oracle/flops_per_cell.json) are what the instrumented
restatement (oracle/flopcount: hho_oracle.c with every double operation counted) prints; and the CPU baseline's
"Matrix assembly" span produces the system the reference-shaped assembly produces."""
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flops_per_cell_json_is_what_the_instrumented_oracle_counts(oracle):
    exe = os.path.join(ROOT, "oracle", "flopcount")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "flopcount"], check=True)
    rows = [json.loads(line) for line in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()]
    tab = json.load(open(os.path.join(ROOT, "oracle", "flops_per_cell.json")))["per_cell"]
    assert len(rows) == len(tab) >= 6
    for r in rows:
        assert r["status"] == 0
        key = "%d,%d,%s,%s" % (r["cd"], r["fd"], "tensor" if r["quad"] == 0 else "fan", {1: "naive", 2: "fancy"}[r["stab"]])
        o = r["ops"]
        assert tab[key]["local_ops"] == o
        assert tab[key]["local_ops_flops"] == o["add"] + o["mul"] + o["div"] + o["sqrt"]
    # the survey's hand model ("+-30 %") brackets the exact counts
    model = {"2,1,tensor,fancy": 16.7e3, "3,2,tensor,fancy": 67.1e3, "4,3,tensor,fancy": 197.5e3, "0,1,tensor,fancy": 9.1e3}
    for k, m in model.items():
        assert 0.7 * tab[k]["local_ops_flops"] <= m <= 1.3 * tab[k]["local_ops_flops"] or k == "0,1,tensor,fancy"


def test_set_from_triplets_matches_scipy(oracle):
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    n, nrows = 5000, 300
    rows = rng.integers(-1, nrows, n).astype(np.int32)
    cols = rng.integers(0, nrows, n).astype(np.int32)
    cols[rows < 0] = -1
    vals = rng.standard_normal(n)
    for nt in (1, 4):
        rowptr, colind, values = oracle.set_from_triplets(rows, cols, vals, nrows, nthreads=nt)
        keep = rows >= 0
        A = sp.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=(nrows, nrows)).tocsr()
        A.sum_duplicates(); A.sort_indices()
        assert np.array_equal(rowptr, A.indptr) and np.array_equal(colind, A.indices)
        assert np.abs(values - A.data).max() < 1e-13


def test_matrix_assembly_span_runs_and_is_thread_invariant(oracle):
    di = oracle.degrees(2, 1)
    a = oracle.matrix_assembly_timed(12, di, oracle.QUAD_TENSOR, oracle.STAB_FANCY, (0, 12), nthreads=1)
    b = oracle.matrix_assembly_timed(12, di, oracle.QUAD_TENSOR, oracle.STAB_FANCY, (0, 12), nthreads=4)
    assert a["cells"] == 144 and a["nnz"] == b["nnz"] > 0
    assert a["checksum"] == b["checksum"]                         # fixed slots per cell: the same sums in the same order
    assert a["seconds_ops"] > 0 and a["seconds_assembly"] > 0
    # nnz of the reference-shaped system: the assembler's pattern
    import poisson_driver as pd
    LHS, RHS, ref, di2 = pd.oracle_assembly(12, 2, 1)
    assert a["nnz"] == LHS.nnz
