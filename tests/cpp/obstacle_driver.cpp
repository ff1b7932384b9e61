// obstacle_driver.cpp -- a driver shaped like the reference's apps/obstacle (obstacle.cpp:47-227):
// primal-dual active set iteration over make_hho_laplacian / make_hho_fancy_stabilization /
// make_rhs(di = 1) / obstacle_assembler, expand_solution, and the energy error against
// project_function(sol_fun, di = 1).  The reference solves with Eigen::SparseLU (obstacle.cpp:170-175,
// outside the hot path); here the block-triangular structure of the system is used (block_solve below): the
// reference's own conjugate gradient on the device for its SPD block, so that config 4 (512 x 512, k = 1) runs end to end.
// Compiled against proton_amd/host/hho.hpp only: no Eigen, no HIP headers.
//   usage: obstacle_driver <degree> <N> [batched|percell] [max outer iterations, default 50 = obstacle.cpp:119]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../proton_amd/host/hho.hpp"

using RealType = double;
using mesh_type = quad_mesh<RealType>;

// x = A^-1 b for the system of obstacle_assembler (hho.hpp:471-751).  The reference hands it to Eigen::SparseLU
// (obstacle.cpp:170-175, outside the hot path); Eigen is not available here, and no sparse LU ships with the image.  The
// system is block triangular by construction: the multiplier of an active cell appears in that cell's row only, with
// coefficient 1 (hho.hpp:688-693), so with the active rows set aside what remains -- rows of the inactive cells and of
// the faces, columns of the same unknowns -- is the symmetric positive definite HHO matrix with the active cells'
// values fixed.  That block is solved with the reference's own Jacobi-preconditioned conjugate gradient on the device
// (conjugated_gradient, solver_cg.hpp:63-144 -> pa_conjugated_gradient), the multipliers follow from their rows.
static std::vector<RealType> block_solve(const proton_amd::sparse_matrix<RealType> &A, const std::vector<RealType> &b, size_t nk)
{
    const size_t n = A.rows();
    std::vector<int64_t> krow(n, -1);                 // row -> row of the SPD block, -1 for the rows of active cells
    std::vector<size_t> active;
    size_t r = 0;
    for (size_t i = 0; i < n; ++i) {
        bool has_multiplier = false;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) has_multiplier = has_multiplier || (size_t)A.colind[k] >= nk;
        if (has_multiplier) active.push_back(i); else krow[i] = (int64_t)r++;
    }
    if (r != nk) throw std::runtime_error("obstacle system: the block of inactive cells and faces is not square");
    proton_amd::sparse_matrix<RealType> K;
    K.nrows = K.ncols = nk;
    K.rowptr.assign(nk + 1, 0);
    std::vector<RealType> bk(nk);
    for (size_t i = 0; i < n; ++i) {
        if (krow[i] < 0) continue;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) { K.colind.push_back(A.colind[k]); K.values.push_back(A.values[k]); }
        K.rowptr[krow[i] + 1] = (int64_t)K.colind.size();
        bk[krow[i]] = b[i];
    }
    cg_params<RealType> cgp;
    cgp.convergence_threshold = 1e-13; cgp.max_iter = 20 * nk; cgp.apply_preconditioner = true;
    std::vector<RealType> y;
    size_t iters = 0;
    if (conjugated_gradient(K, bk, y, cgp, &iters) != cg_exit_reason::CONVERGED) throw std::runtime_error("obstacle system: CG did not converge");
    std::vector<RealType> x(n, 0.0);
    for (size_t j = 0; j < nk; ++j) x[j] = y[j];
    for (size_t i : active) {                         // beta = b_i - sum_j A_ij y_j, at the row's multiplier column
        RealType s = b[i];
        size_t col = n;
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
            if ((size_t)A.colind[k] >= nk) col = (size_t)A.colind[k];
            else s -= A.values[k] * y[A.colind[k]];
        }
        x[col] = s;
    }
    return x;
}

int main(int argc, char **argv)
{
    const size_t degree = argc > 1 ? std::atoi(argv[1]) : 1;
    const size_t N = argc > 2 ? std::atoi(argv[2]) : 8;
    const bool batched = argc > 3 && std::string(argv[3]) == "batched";
    const size_t max_outer = argc > 4 ? std::atoi(argv[4]) : 50;               // obstacle.cpp:119: while (iter < 50)

    mesh_init_params<RealType> mip;                                         // obstacle.cpp:234-238,276
    mip.Nx = N; mip.Ny = N;
    mip.min_x = -1; mip.max_x = 1; mip.min_y = -1; mip.max_y = 1;
    mesh_type msh(mip);

    hho_degree_info hdi(0, degree);                                         // obstacle.cpp:51
    const RealType r0 = 0.7;
    auto rhs_fun = [=](const mesh_type::point_type &pt) -> RealType {       // obstacle.cpp:67-74
        const RealType r = std::sqrt(pt.x() * pt.x() + pt.y() * pt.y());
        return r > r0 ? -16 * r * r + 8 * r0 * r0 : -8.0 * (r0 * r0 * (r0 * r0 + 1)) + 8 * r0 * r0 * r * r;
    };
    auto sol_fun = [=](const mesh_type::point_type &pt) -> RealType {       // obstacle.cpp:76-81
        const RealType r = std::sqrt(pt.x() * pt.x() + pt.y() * pt.y());
        const RealType t = std::max(r * r - r0 * r0, 0.0);
        return t * t;
    };
    auto bcs_fun = [&](const mesh_type::point_type &pt) -> RealType { return sol_fun(pt); };

    const size_t num_cells = msh.cells.size(), num_faces = msh.faces.size(), fbs = degree + 1;
    std::vector<RealType> alpha(num_cells + fbs * num_faces, 0.0), beta(num_cells, 1.0), gamma(num_cells, 0.0);
    const RealType c = 1.0;
    const size_t quadrature_degree_increase = 1;                            // obstacle.cpp:103

    // the local operators do not change between iterations: computed once on the device
    pa_sizes sz;
    auto &dev = proton_amd::device::instance();
    dev.check(pa_sizes_for(hdi.c_abi(), PA_QUAD_TENSOR, &sz), "pa_sizes_for");
    proton_amd::device_buffer<double> d_lc, d_rhs, d_g;
    if (batched) {
        proton_amd::batch_cache<mesh_type>::instance().ensure_mesh(msh);
        d_lc.resize(num_cells * sz.msize * sz.msize); d_rhs.resize(num_cells * sz.cbs); d_g.resize(num_faces * sz.fbs);
        dev.check(pa_local_ops_batch(dev.ctx(), hdi.c_abi(), PA_QUAD_TENSOR, PA_STAB_FANCY, 0, num_cells, nullptr, nullptr, nullptr,
                                     d_lc.get(), nullptr), "pa_local_ops_batch");
        dev.check(pa_cell_rhs_batch(dev.ctx(), 0, (int)quadrature_degree_increase, PA_QUAD_TENSOR, PA_FN_OBSTACLE_RHS, nullptr, 0,
                                    num_cells, d_rhs.get()), "pa_cell_rhs_batch");
        dev.check(pa_dirichlet_data_batch(dev.ctx(), (int)degree, PA_FN_OBSTACLE_SOL, nullptr, d_g.get()), "pa_dirichlet_data_batch");
    }

    size_t iter = 0;
    bool converged = false;
    for (; iter < max_outer; ++iter) {
        std::vector<bool> in_A(num_cells);
        for (size_t i = 0; i < num_cells; ++i) in_A[i] = (beta[i] + c * (alpha[i] - gamma[i])) < 0;     // obstacle.cpp:133-142

        auto assembler = make_obstacle_assembler(msh, in_A, hdi);
        if (batched) {
            assembler.assemble_all(msh, d_lc.get(), d_rhs.get(), d_g.get(), gamma);
        } else {
            for (auto &cl : msh.cells) {                                    // obstacle.cpp:148-156
                auto gr = make_hho_laplacian(msh, cl, hdi);
                auto stab = make_hho_fancy_stabilization(msh, cl, gr.first, hdi);
                auto lc = gr.second + stab;
                auto f = make_rhs(msh, cl, hdi.cell_degree(), rhs_fun, quadrature_degree_increase);
                assembler.assemble(msh, cl, lc, f, gamma, bcs_fun);
            }
        }
        assembler.finalize();

        size_t num_A = 0;
        for (size_t i = 0; i < num_cells; ++i) num_A += in_A[i] ? 1 : 0;
        const auto sol = block_solve(assembler.LHS, assembler.RHS, assembler.RHS.size() - num_A);
        const auto alpha_prev = alpha;
        assembler.expand_solution(msh, sol, bcs_fun, gamma, alpha, beta);
        RealType d2 = 0.0;
        for (size_t i = 0; i < alpha.size(); ++i) d2 += (alpha_prev[i] - alpha[i]) * (alpha_prev[i] - alpha[i]);
        if (std::sqrt(d2) < 1e-7) { converged = true; break; }              // obstacle.cpp:193
    }

    RealType error = 0.0;                                                   // obstacle.cpp:202-213
    std::vector<RealType> proj_all;
    if (batched) proj_all = project_function_all(msh, hdi, sol_fun, quadrature_degree_increase);      // one device batch
    for (auto &cl : msh.cells) {
        auto local = take_local_data(msh, cl, hdi, alpha);
        proton_amd::dense_matrix<RealType> proj;
        if (batched) {
            proj = proton_amd::dense_matrix<RealType>(local.rows(), 1);
            std::memcpy(proj.data(), proj_all.data() + offset(msh, cl) * local.rows(), local.rows() * sizeof(RealType));
        } else {
            proj = project_function(msh, cl, hdi, sol_fun, quadrature_degree_increase);
        }
        auto gr = make_hho_laplacian(msh, cl, hdi);
        auto lc = gr.second + make_hho_fancy_stabilization(msh, cl, gr.first, hdi);
        auto diff = local - proj;
        error += diff.dot(lc * diff);
    }
    std::printf("N %zu degree %zu iterations %zu error %.10e converged %d\n", N, degree, iter + (converged ? 1 : 0), std::sqrt(error),
                converged ? 1 : 0);
    return 0;
}
