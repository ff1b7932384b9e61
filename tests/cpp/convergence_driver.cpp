// convergence_driver.cpp -- a driver shaped like the reference's apps/convergence_test
// (convergence_test.cpp:140-329): per-cell make_hho_laplacian / make_hho_fancy_stabilization /
// make_rhs / assembler.assemble, finalize, Jacobi-PCG solve (the reference's own solver,
// src/core/core_bits/solver_cg.hpp:63-144, restated for the CSR type), L2 error and rates.
// Compiled against proton_amd/host/hho.hpp only: no Eigen, no HIP headers.
//   usage: convergence_driver <k> <min_N> <steps> [batched|device]   (device: batched assembly, device CSR, device PCG)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../proton_amd/host/hho.hpp"

using RealType = double;
using mesh_type = quad_mesh<RealType>;

// Jacobi-preconditioned conjugate gradient, semantics of solver_cg.hpp:63-144
static size_t pcg(const proton_amd::sparse_matrix<RealType> &A, const std::vector<RealType> &b, std::vector<RealType> &x,
                  RealType tol, size_t max_iter)
{
    const size_t N = A.rows();
    std::vector<RealType> iD(N, 1.0), r(N), d(N), y(N), z(N);
    for (size_t i = 0; i < N; ++i)
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if ((size_t)A.colind[k] == i) iD[i] = 1.0 / A.values[k];
    x.assign(N, 0.0);
    r = b;
    RealType nr0 = 0.0;
    for (size_t i = 0; i < N; ++i) { d[i] = iD[i] * r[i]; nr0 += r[i] * r[i]; }
    nr0 = std::sqrt(nr0);
    RealType rho = 0.0;
    for (size_t i = 0; i < N; ++i) rho += r[i] * d[i];
    size_t it = 0;
    for (; it < max_iter; ++it) {
        y = A.multiply(d);
        RealType dy = 0.0;
        for (size_t i = 0; i < N; ++i) dy += d[i] * y[i];
        const RealType alpha = rho / dy;
        RealType nr = 0.0;
        for (size_t i = 0; i < N; ++i) { x[i] += alpha * d[i]; r[i] -= alpha * y[i]; nr += r[i] * r[i]; }
        if (std::sqrt(nr) / nr0 < tol) break;
        RealType rho1 = 0.0;
        for (size_t i = 0; i < N; ++i) { z[i] = iD[i] * r[i]; rho1 += r[i] * z[i]; }
        const RealType beta = rho1 / rho;
        for (size_t i = 0; i < N; ++i) d[i] = z[i] + beta * d[i];
        rho = rho1;
    }
    return it;
}

int main(int argc, char **argv)
{
    const size_t k = argc > 1 ? std::atoi(argv[1]) : 1;
    const size_t min_N = argc > 2 ? std::atoi(argv[2]) : 4;
    const size_t steps = argc > 3 ? std::atoi(argv[3]) : 3;
    const bool batched = argc > 4;
    const bool device_solve = argc > 4 && std::string(argv[4]) == "device";

    auto rhs_fun = [](const mesh_type::point_type &pt) -> RealType {      // convergence_test.cpp:100-102
        return 2.0 * M_PI * M_PI * std::sin(M_PI * pt.x()) * std::sin(M_PI * pt.y());
    };
    auto sol_fun = [](const mesh_type::point_type &pt) -> RealType {      // convergence_test.cpp:104-106
        return std::sin(M_PI * pt.x()) * std::sin(M_PI * pt.y());
    };

    hho_degree_info hdi(k + 1, k);                                          // convergence_test.cpp:163
    std::vector<RealType> errors;
    for (size_t i = 0, N = min_N; i < steps; i++, N *= 2) {
        mesh_init_params<RealType> mip;
        mip.Nx = N; mip.Ny = N;
        mesh_type msh(mip);

        const auto t0 = std::chrono::steady_clock::now();
        auto assembler = make_assembler(msh, hdi);
        if (batched) {
            assembler.assemble_all(msh, PA_STAB_FANCY, PA_FN_SIN_SIN_RHS, PA_FN_SIN_SIN_SOL);
        } else {
            for (auto &cl : msh.cells) {                                    // convergence_test.cpp:202-215
                auto gr = make_hho_laplacian(msh, cl, hdi);
                auto stab = make_hho_fancy_stabilization(msh, cl, gr.first, hdi);
                auto lc = gr.second + stab;
                auto f = make_rhs(msh, cl, hdi.cell_degree(), rhs_fun);
                assembler.assemble(msh, cl, lc, f, sol_fun);
            }
        }
        assembler.finalize();
        const double t_asm = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

        std::vector<RealType> sol;
        size_t iters = 0;
        if (device_solve) {                                                 // the reference's conjugated_gradient, on the device
            cg_params<RealType> cgp;
            cgp.convergence_threshold = 1e-12; cgp.max_iter = 3 * assembler.LHS.rows(); cgp.apply_preconditioner = true;
            if (conjugated_gradient(assembler.LHS, assembler.RHS, sol, cgp, &iters) != cg_exit_reason::CONVERGED) return 2;
        } else {
            iters = pcg(assembler.LHS, assembler.RHS, sol, 1e-12, 3 * assembler.LHS.rows());
        }

        // errors_int, convergence_test.cpp:254-268: the device supplies the quadrature points
        RealType err = 0.0;
        {
            auto &cache = proton_amd::batch_cache<mesh_type>::instance();
            int nq = 0;
            const auto &xyw = cache.cell_qpoints(msh, (int)(2 * hdi.cell_degree()), PA_QUAD_TENSOR, nq);
            const size_t cd = hdi.cell_degree(), cbs = (cd + 2) * (cd + 1) / 2;
            for (size_t c = 0; c < msh.cells.size(); ++c) {
                const auto bar = barycenter(msh, msh.cells[c]);
                const auto h = diameter(msh, msh.cells[c]);
                for (int q = 0; q < nq; ++q) {
                    const double x = xyw[(c * nq + q) * 3], y = xyw[(c * nq + q) * 3 + 1], w = xyw[(c * nq + q) * 3 + 2];
                    const double bx = (x - bar.x()) / (0.5 * h), by = (y - bar.y()) / (0.5 * h);
                    double val = 0.0;
                    size_t pos = 0;
                    for (size_t kk = 0; kk <= cd; kk++)                       // bases.hpp:114-128
                        for (size_t ii = 0; ii <= kk; ii++) val += sol[c * cbs + pos++] * std::pow(bx, (double)(kk - ii)) * std::pow(by, (double)ii);
                    const double real_val = sol_fun(mesh_type::point_type(x, y));
                    err += w * (real_val - val) * (real_val - val);
                }
            }
        }
        errors.push_back(std::sqrt(err));
        std::printf("N %zu k %zu system %zu nnz %zu assembly_s %.4f cg_iters %zu l2_error %.10e\n", N, k, assembler.LHS.rows(),
                    assembler.LHS.nonZeros(), t_asm, iters, errors.back());
        if (i > 0) std::printf("rate %.4f\n", std::log2(errors[i - 1] / errors[i]));
    }
    return 0;
}
