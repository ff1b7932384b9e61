// cuthho_driver.cpp -- a driver shaped like the reference's apps/cuthho/cuthho_square.cpp `-f`
// (main :2020-2052, run_cuthho_fictdom :806-1080): preprocessing steps, per-cell cut / uncut
// operators, the generic assembler, Jacobi-PCG (the system is symmetric positive definite; the
// reference uses SparseLU here) and the energy-norm error of :1030-1049.
// Compiled against proton_amd/host/cuthho.hpp only: no Eigen, no HIP headers.
//   usage: cuthho_driver <k> <N> [refsteps]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../proton_amd/host/cuthho.hpp"

using RealType = double;
using mesh_type = cuthho_poly_mesh<RealType>;

static size_t pcg(const proton_amd::sparse_matrix<RealType> &A, const std::vector<RealType> &b, std::vector<RealType> &x,
                  RealType tol, size_t max_iter)
{
    const size_t N = A.rows();
    std::vector<RealType> iD(N, 1.0), r(b), d(N), y(N), z(N);
    for (size_t i = 0; i < N; ++i)
        for (int64_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
            if ((size_t)A.colind[k] == i) iD[i] = 1.0 / A.values[k];
    x.assign(N, 0.0);
    RealType nr0 = 0.0, rho = 0.0;
    for (size_t i = 0; i < N; ++i) { d[i] = iD[i] * r[i]; nr0 += r[i] * r[i]; rho += r[i] * d[i]; }
    nr0 = std::sqrt(nr0);
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
        for (size_t i = 0; i < N; ++i) d[i] = z[i] + (rho1 / rho) * d[i];
        rho = rho1;
    }
    return it;
}

int main(int argc, char **argv)
{
    const size_t degree = argc > 1 ? std::atoi(argv[1]) : 1;
    const size_t N = argc > 2 ? std::atoi(argv[2]) : 10;
    const size_t int_refsteps = argc > 3 ? std::atoi(argv[3]) : 4;

    mesh_init_params<RealType> mip;
    mip.Nx = N; mip.Ny = N;
    mesh_type msh(mip);
    auto level_set_function = circle_level_set<RealType>(0.35, 0.5, 0.5);        // cuthho_square.cpp:2029-2030

    detect_node_position(msh, level_set_function);                                // cuthho_square.cpp:2036-2052
    detect_cut_faces(msh, level_set_function);
    move_nodes(msh, level_set_function);
    detect_cut_faces(msh, level_set_function);
    detect_cut_cells(msh, level_set_function);
    refine_interface(msh, level_set_function, int_refsteps);

    auto rhs_fun = [](const mesh_type::point_type &pt) -> RealType {
        return 2.0 * M_PI * M_PI * std::sin(M_PI * pt.x()) * std::sin(M_PI * pt.y());
    };
    auto sol_fun = [](const mesh_type::point_type &pt) -> RealType { return std::sin(M_PI * pt.x()) * std::sin(M_PI * pt.y()); };
    auto bcs_fun = [&](const mesh_type::point_type &pt) -> RealType { return sol_fun(pt); };

    hho_degree_info hdi(degree + 1, degree);                                      // :871
    const element_location where = element_location::IN_NEGATIVE_SIDE;
    auto assembler = make_assembler(msh, hdi);
    size_t ncut = 0;
    for (auto &cl : msh.cells) {                                                  // :883-900
        auto gr = make_hho_laplacian(msh, cl, level_set_function, hdi, where);
        auto stab = make_hho_cut_stabilization(msh, cl, hdi, where);
        auto lc = gr.second + stab;
        auto f = make_rhs(msh, cl, hdi.cell_degree(), rhs_fun, where, level_set_function, bcs_fun);
        assembler.assemble(msh, cl, lc, f, bcs_fun);
        ncut += is_cut(msh, cl) ? 1 : 0;
    }
    assembler.finalize();

    std::vector<RealType> sol;
    const size_t iters = pcg(assembler.LHS, assembler.RHS, sol, 1e-13, 4 * assembler.LHS.rows());

    RealType H1_error = 0.0;                                                      // :1030-1049
    const size_t cd = hdi.cell_degree(), cbs = (cd + 2) * (cd + 1) / 2;
    for (auto &cl : msh.cells) {
        if (location(msh, cl) == element_location::IN_POSITIVE_SIDE) continue;
        const size_t c = offset(msh, cl);
        const auto bar = barycenter(msh, cl);
        const auto h = diameter(msh, cl);
        for (auto &qp : integrate(msh, cl, 2 * cd, where)) {
            const double bx = (qp.first.x() - bar.x()) / (0.5 * h), by = (qp.first.y() - bar.y()) / (0.5 * h);
            double gx = 0.0, gy = 0.0;
            size_t pos = 0;
            for (size_t kk = 0; kk <= cd; kk++)                                   // bases.hpp:142-184
                for (size_t ii = 0; ii <= kk; ii++, pos++) {
                    if (pos == 0) continue;
                    const double px = (double)(kk - ii), py = (double)ii, u = sol[c * cbs + pos];
                    if (kk - ii > 0) gx += u * px * (2.0 / h) * std::pow(bx, px - 1) * std::pow(by, py);
                    if (ii > 0) gy += u * py * (2.0 / h) * std::pow(bx, px) * std::pow(by, py - 1);
                }
            const double sx = M_PI * std::cos(M_PI * qp.first.x()) * std::sin(M_PI * qp.first.y());
            const double sy = M_PI * std::sin(M_PI * qp.first.x()) * std::cos(M_PI * qp.first.y());
            H1_error += qp.second * ((sx - gx) * (sx - gx) + (sy - gy) * (sy - gy));
        }
    }
    std::printf("N %zu k %zu r %zu cut_cells %zu system %zu cg_iters %zu energy_error %.10e\n", N, degree, int_refsteps, ncut,
                assembler.LHS.rows(), iters, std::sqrt(H1_error));
    return 0;
}
