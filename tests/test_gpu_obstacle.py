"""GPU obstacle_assembler (hho.hpp:471-751) through the C ABI against the oracle's C restatement:
compress tables, triplet (row, col) sequences and right-hand-side row maps BIT-EXACT, values exact
copies, right-hand-side updates to rounding; expand_solution / take_local_data exact; and the
whole primal-dual active set loop of apps/obstacle/obstacle.cpp:47-227 with the GPU's operators
AND assembler reproduces apps/obstacle/results/convergence.txt."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asm():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    from proton_amd.batch import BatchAssembler
    return BatchAssembler(0)


def _dev(a, asm, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(asm.device)


@pytest.mark.parametrize("n,p", [(1, 0.5), (7, 0.0), (64, 1.0), (2048, 0.3), (2049, 0.5), (100_000, 0.01), (1 << 20, 0.7)])
def test_tables_bit_exact(asm, oracle, n, p):
    """A_ct / B_ct (hho.hpp:538-578) for ragged sizes around the scan tile, empty and full sets."""
    import ctypes as C
    N = int(np.ceil(np.sqrt(n)))
    asm.generate_mesh(N, -(-n // N))
    nc = asm.ncells
    rng = np.random.default_rng(n)
    in_A = (rng.random(nc) < p).astype(np.uint8)
    A_ct, B_ct, num_I, num_A = asm.obstacle_tables(_dev(in_A, asm))
    ra, rb = np.zeros(nc, dtype=np.int64), np.zeros(nc, dtype=np.int64)
    ni, na = C.c_size_t(0), C.c_size_t(0)
    oracle.lib().hho_obstacle_tables(oracle._u8p(in_A), nc, oracle._i64p(ra), oracle._i64p(rb), C.byref(ni), C.byref(na))
    assert (num_I, num_A) == (ni.value, na.value)
    assert np.array_equal(A_ct.cpu().numpy().astype(np.int64), ra)
    assert np.array_equal(B_ct.cpu().numpy().astype(np.int64), rb)


@pytest.mark.parametrize("N,cd,fd", [(6, 0, 0), (7, 0, 1), (5, 1, 1), (4, 1, 2)])
def test_obstacle_triplets_expand_take(asm, oracle, N, cd, fd):
    """cd = 0 is what obstacle.cpp:51 uses; cd = 1 exercises the formulas as written for cbs > 1
    (rows at cell + i, quirk 10 of the survey) -- restated literally on both sides."""
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    asm.generate_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
    out = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))
    rhs = asm.cell_rhs(cd, pa.capi.FN_OBSTACLE_RHS, pa.QUAD_TENSOR, dinc=1)
    g = asm.dirichlet_data(fd, pa.capi.FN_OBSTACLE_SOL)
    nc = N * N
    di0 = oracle.degrees(cd, fd)
    rng = np.random.default_rng(N * 10 + fd)
    in_A = (rng.random(nc) < 0.4).astype(np.uint8)
    gamma = rng.standard_normal(nc * di0.cbs)
    d_in_A, d_gamma = _dev(in_A, asm), _dev(gamma, asm)
    A_ct, B_ct, num_I, num_A = asm.obstacle_tables(d_in_A)
    r, c, v, rr, rv = asm.obstacle_triplets(cd, fd, out["lc"], rhs, g, d_gamma, d_in_A, A_ct, B_ct, num_I)
    asm.synchronize()

    mp, points, ptids = oracle.make_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
    di = oracle.degrees(cd, fd)
    ref = oracle.ObstacleAssembler(mp, points, ptids, di, in_A, bf_id=4)
    assert (num_I, num_A) == (ref.num_I, ref.num_A)
    gh = g.cpu().numpy()[:ref.nf]
    assert np.abs(gh - ref.g).max() < 1e-13 * max(1.0, np.abs(ref.g).max())
    ref.g = gh.copy()                       # same Dirichlet data on both sides from here on
    lch, rhsh = to_rowcol(out["lc"]), rhs.cpu().numpy()
    R, Cc, V, RR, RV = (t.cpu().numpy() for t in (r, c, v, rr, rv))
    ms = di.msize
    assert R.shape == (nc, ms * ms + 1)
    for cell in range(nc):
        tr, tc, tv, rrow, rval = ref.assemble_cell(cell, lch[cell], rhsh[cell], gamma)
        keep = R[cell] >= 0
        assert np.array_equal(keep, Cc[cell] >= 0)
        assert np.array_equal(R[cell][keep], tr)                      # bit-exact, in the reference's push order
        assert np.array_equal(Cc[cell][keep], tc)
        assert np.array_equal(V[cell][keep], tv)
        assert keep[-1] == bool(in_A[cell])                           # multiplier coupling, hho.hpp:688-693
        assert np.array_equal(RR[cell].astype(np.int64), rrow)
        assert np.abs(RV[cell] - rval).max() <= 1e-13 * max(1.0, np.abs(rval).max())

    # expand_solution (hho.hpp:698-744) and take_local_data (hho.hpp:753-782): pure gathers, exact
    sol = rng.standard_normal(ref.system_size)
    alpha, beta = asm.obstacle_expand_solution(cd, fd, _dev(sol, asm), g, d_gamma, d_in_A, A_ct, B_ct, num_I, ref.nf)
    ra, rb = ref.expand_solution(sol, gamma)
    assert np.array_equal(alpha.cpu().numpy(), ra)
    assert np.array_equal(beta.cpu().numpy(), rb)
    loc = asm.obstacle_take_local_data(cd, fd, alpha).cpu().numpy()
    for cell in range(nc):
        assert np.array_equal(loc[cell], ref.take_local_data(cell, ra))


@pytest.mark.parametrize("N,cd,fd", [(5, 2, 1), (6, 0, 1), (4, 3, 2)])
def test_take_local_data_of_the_plain_assembler(asm, oracle, N, cd, fd):
    """assembler::take_local_data (hho.hpp:408-449): cell dofs, compressed face dofs, Dirichlet data."""
    import proton_amd as pa
    asm.generate_mesh(N, N)
    g = asm.dirichlet_data(fd, pa.capi.FN_SIN_SIN_SOL)
    mp, points, ptids = oracle.make_mesh(N, N)
    di = oracle.degrees(cd, fd)
    ref = oracle.Assembler(mp, points, ptids, di, bf_id=2)
    rng = np.random.default_rng(3)
    sol = rng.standard_normal(ref.system_size)
    got = asm.take_local_data(cd, fd, _dev(sol, asm), g).cpu().numpy()
    gh = g.cpu().numpy()
    cbs, fbs = di.cbs, di.fbs
    for cell in range(N * N):
        want = np.zeros(di.msize)
        want[:cbs] = sol[cell * cbs:(cell + 1) * cbs]
        for lf in range(4):
            f = int(ref.cell_faces[cell, lf])
            if ref.is_dir[f]:
                want[cbs + lf * fbs: cbs + (lf + 1) * fbs] = gh[f]
            else:
                o0 = cbs * N * N + int(ref.compress[f]) * fbs
                want[cbs + lf * fbs: cbs + (lf + 1) * fbs] = sol[o0:o0 + fbs]
        assert np.array_equal(got[cell], want)


@pytest.mark.parametrize("N,cd,fd,dinc,quad", [(5, 0, 1, 1, 0), (4, 2, 1, 0, 0), (3, 3, 2, 1, 0), (3, 4, 3, 0, 0), (4, 2, 1, 0, 1), (3, 0, 0, 2, 0)])
def test_project_function_and_energy_form(asm, oracle, N, cd, fd, dinc, quad):
    """project_function (utils.hpp:199-227) against the oracle, and diff.dot(lc*diff) per cell."""
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    asm.generate_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
    got = asm.project_function(cd, fd, pa.capi.FN_OBSTACLE_SOL, quad=quad, dinc=dinc).cpu().numpy()
    mp, points, ptids = oracle.make_mesh(N, N, (-1.0, -1.0), (1.0, 1.0))
    di = oracle.degrees(cd, fd)
    worst = 0.0
    sol_host = lambda x, y: max(x * x + y * y - 0.7 * 0.7, 0.0) ** 2          # obstacle.cpp:76-81
    for c in range(N * N):
        pts = points[ptids[c].astype(np.int64)]
        st, want = oracle.project_function(pts, ptids[c], di, sol_host, quad=quad, dinc=dinc)
        assert st == 0
        worst = max(worst, np.abs(got[c] - want).max() / max(1.0, np.abs(want).max()))
    assert worst < 1e-12, worst
    # sampled functor path == built-in path
    import torch
    nq = asm.quadrature_points(2 * (di.cell_deg + dinc), quad)
    fq = asm.face_quadrature_points(di.face_deg + dinc)
    sol = lambda xy: torch.clamp(xy[..., 0] ** 2 + xy[..., 1] ** 2 - 0.7 * 0.7, min=0.0) ** 2
    got2 = asm.project_function(cd, fd, pa.capi.FN_SAMPLED, quad=quad, dinc=dinc, cell_fvals=sol(nq).contiguous(),
                                face_fvals=sol(fq).contiguous()).cpu().numpy()
    assert np.abs(got2 - got).max() < 1e-12
    if quad == 0:
        lc = asm.local_ops(cd, fd, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
        rng = np.random.default_rng(5)
        u = rng.standard_normal(got.shape)
        e = asm.energy_form(cd, fd, lc, _dev(u, asm), _dev(got, asm)).cpu().numpy()
        L = to_rowcol(lc)
        want_e = np.einsum("ci,cij,cj->c", u - got, L, u - got)
        assert np.abs(e - want_e).max() <= 1e-12 * np.abs(want_e).max()
        e0 = asm.energy_form(cd, fd, lc, _dev(u, asm)).cpu().numpy()
        assert np.abs(e0 - np.einsum("ci,cij,cj->c", u, L, u)).max() <= 1e-12 * np.abs(e0).max()


class GpuObstacleAssembler:
    """The shape tests/obstacle_driver.run_obstacle expects, over the C ABI."""

    def __init__(self, asm, msh, di, in_A):
        self.asm, self.di = asm, di
        self.nf = msh.nfaces
        self.in_A = _dev(in_A.astype(np.uint8), asm)
        self.A_ct, self.B_ct, self.num_I, self.num_A = asm.obstacle_tables(self.in_A)
        info = asm.assembler_info(0, di.face_deg)
        self.system_size = info.system_size

    def assemble_all(self, lc_unused, rhs_unused, gamma):
        import torch
        a = self.asm
        self.gamma = _dev(gamma, a)
        r, c, v, rr, rv = a.obstacle_triplets(0, self.di.face_deg, a._lc, a._rhs, a._g, self.gamma, self.in_A, self.A_ct,
                                              self.B_ct, self.num_I)
        keep = r >= 0
        RHS = torch.zeros(self.system_size, dtype=torch.float64, device=a.device)
        ok = rr >= 0
        RHS.index_add_(0, rr[ok].long(), rv[ok])
        return r[keep].cpu().numpy(), c[keep].cpu().numpy(), v[keep].cpu().numpy(), RHS.cpu().numpy()

    def expand_solution(self, sol, gamma):
        a = self.asm
        alpha, beta = a.obstacle_expand_solution(0, self.di.face_deg, _dev(sol, a), a._g, self.gamma, self.in_A, self.A_ct,
                                                 self.B_ct, self.num_I, self.nf)
        self._alpha = alpha
        self._local = None
        return alpha.cpu().numpy(), beta.cpu().numpy()

    def take_local_data(self, c, alpha_host):
        if self._local is None:
            self._local = self.asm.obstacle_take_local_data(0, self.di.face_deg, self._alpha).cpu().numpy()
        return self._local[c]


@pytest.mark.parametrize("N,degree", [(8, 0), (16, 1), (32, 1)])
def test_obstacle_end_to_end_on_gpu_operators_and_assembler(asm, N, degree):
    """configs[3]: obstacle -N N -k degree with the GPU's operators, right-hand sides, Dirichlet data,
    obstacle assembler, expand_solution and take_local_data; only the sparse solve is host-side."""
    import obstacle_driver as od
    import proton_amd as pa
    from proton_amd.batch import to_rowcol
    REF = {8: (2.26205, 0.197735), 16: (1.2833, 0.0588187), 32: (0.650286, 0.0171607)}

    def gpu_provider(msh, deg):
        asm.generate_mesh(msh.N, msh.N, (-1.0, -1.0), (1.0, 1.0))       # obstacle.cpp:234-238
        asm._lc = asm.local_ops(0, deg, pa.QUAD_TENSOR, pa.STAB_FANCY, want=("lc",))["lc"]
        asm._rhs = asm.cell_rhs(0, pa.capi.FN_OBSTACLE_RHS, pa.QUAD_TENSOR, dinc=1)
        asm._g = asm.dirichlet_data(deg, pa.capi.FN_OBSTACLE_SOL)
        asm.synchronize()
        return to_rowcol(asm._lc), asm._rhs.cpu().numpy()

    err, iters = od.run_obstacle(N, degree, local_provider=gpu_provider,
                                 assembler_factory=lambda msh, di, in_A: GpuObstacleAssembler(asm, msh, di, in_A))
    assert iters < 50
    assert abs(err - REF[N][degree]) / REF[N][degree] < 5e-6


def test_obstacle_entry_points_refuse_bad_input(asm):
    """status codes instead of undefined behaviour: the obstacle assembler needs the whole mesh on the
    context (no row slab), sizes are checked before anything is launched"""
    import ctypes as C
    import torch
    import proton_amd as pa
    L = pa.capi.lib()
    h = asm.ctx.h
    di, _ = pa.degree_info(0, 1)
    asm.generate_mesh(8, 8, rows=(2, 6))                         # a slab: obstacle entry points must refuse
    n = asm.ncells
    z8 = torch.zeros(n, dtype=torch.uint8, device=asm.device)
    zi = torch.zeros(n * 82, dtype=torch.int32, device=asm.device)
    zd = torch.zeros(n * 82, dtype=torch.float64, device=asm.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = L.pa_obstacle_triplets_batch(h, di, 0, n, p(zd), None, None, p(zd), p(z8), p(zi), p(zi), 0, p(zi), p(zi), p(zd), p(zi), p(zd))
    assert st == 1 and b"whole mesh" in L.pa_last_error(h)
    assert L.pa_obstacle_expand_solution(h, di, p(zd), None, p(zd), p(z8), p(zi), p(zi), 0, p(zd), p(zd)) == 1
    asm.generate_mesh(8, 8)
    n = asm.ncells
    assert L.pa_obstacle_triplets_batch(h, di, 0, n + 1, p(zd), None, None, p(zd), p(z8), p(zi), p(zi), 0, p(zi), p(zi), p(zd), p(zi), p(zd)) == 1
    assert L.pa_obstacle_triplets_batch(h, di, 0, n, None, None, None, p(zd), p(z8), p(zi), p(zi), 0, p(zi), p(zi), p(zd), p(zi), p(zd)) == 1
    assert L.pa_obstacle_tables(h, None, p(zi), p(zi), None, None) == 1
    bad = pa.capi.DegreeInfo(7, 9, 10)
    assert L.pa_obstacle_triplets_batch(h, bad, 0, n, p(zd), None, None, p(zd), p(z8), p(zi), p(zi), 0, p(zi), p(zi), p(zd), p(zi), p(zd)) == 2
    # device CSR: more than 2^31 - 1 slots cannot be indexed by its int32 tables
    nnz = C.c_size_t(0)
    assert L.pa_csr_from_triplets(h, 1 << 31, p(zi), p(zi), p(zd), 10, p(zd), p(zi), p(zd), C.byref(nnz)) == 1
    assert L.pa_take_local_data_batch(h, di, 0, n + 5, p(zd), None, p(zd)) == 1
