"""PIN of the oracle against the reference's own committed numbers for this path:
apps/obstacle/results/convergence.txt (columns: N, error k=0, error k=1; legend in
convergence.plot:13-14).  The whole chain is exercised -- mesh generator numbering, cell/face
bases, tensor Gauss rules, make_hho_laplacian, make_hho_fancy_stabilization, make_rhs(di=1),
project_function, the obstacle_assembler's three compress tables, the primal-dual active set loop
(obstacle.cpp:47-227) -- and must reproduce the 6 printed significant digits."""
import pytest

import obstacle_driver as od

# apps/obstacle/results/convergence.txt:1-5
REFERENCE = {8: (2.26205, 0.197735), 16: (1.2833, 0.0588187), 32: (0.650286, 0.0171607),
             64: (0.326314, 0.00529786), 128: (0.163344, 0.00168321)}


def printed(x):
    return float("%.6g" % x)


@pytest.mark.parametrize("N", [8, 16, 32])
@pytest.mark.parametrize("degree", [0, 1])
def test_obstacle_error_matches_committed_results(N, degree):
    err, iters = od.run_obstacle(N, degree)
    assert iters < 50
    assert printed(err) == pytest.approx(REFERENCE[N][degree], rel=2e-5)     # last printed digit
    assert abs(err - REFERENCE[N][degree]) / REFERENCE[N][degree] < 5e-6


def test_obstacle_error_n64_k1():
    err, iters = od.run_obstacle(64, 1)
    assert abs(err - REFERENCE[64][1]) / REFERENCE[64][1] < 5e-6
