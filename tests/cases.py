"""Shared geometric test cases (single cells) for oracle / golden / GPU parity tests."""
import numpy as np

# (name, pts[4][2] CCW, point ids) -- ids decide the face-basis orientation (bases.hpp:260-261)
CELLS = {
    # cell (1,1) of a 4x4 mesh on [0,1]^2: ids p, p+1, p+Nx+2, p+Nx+1 with p = 6
    "square": (np.array([[0.25, 0.25], [0.5, 0.25], [0.5, 0.5], [0.25, 0.5]]), (6, 7, 12, 11)),
    # non-affine quad, ids chosen so that every face flips differently from the generated mesh
    "distorted": (np.array([[0.10, 0.05], [0.62, 0.11], [0.55, 0.58], [0.02, 0.47]]), (9, 3, 12, 7)),
    # obstacle-style cell on [-1,1]^2 touching the corner
    "corner": (np.array([[-1.0, -1.0], [-0.75, -1.0], [-0.75, -0.75], [-1.0, -0.75]]), (0, 1, 10, 9)),
    # thin rectangle (anisotropic)
    "thin": (np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.125], [0.0, 0.125]]), (0, 1, 3, 2)),
}

# (cell_degree, face_degree) pairs used by the reference drivers:
#   (k+1,k) convergence_test.cpp:163 / cuthho_square.cpp:871 ; (0,k) obstacle.cpp:51 ; (k,k) hho_degree_info(k)
DEGREES = [(2, 1), (3, 2), (4, 3), (0, 0), (0, 1), (1, 1), (2, 2), (3, 3), (1, 0), (1, 2)]
