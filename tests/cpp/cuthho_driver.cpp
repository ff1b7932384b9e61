// cuthho_driver.cpp -- a driver shaped like the reference's apps/cuthho/cuthho_square.cpp
// (main :1940-2068 with its flags -k -M -N -r -f -i; run_cuthho_fictdom :806-1080;
// run_cuthho_interface :1625-1846): preprocessing steps, per-cell cut / uncut operators, the generic
// assembler or the interface_assembler, a conjugate-gradient solve (both systems are symmetric
// positive definite; the reference uses SparseLU for -f and its Jacobi-PCG for -i) and the
// energy-norm errors of :1030-1049 / :1762-1833.
// Compiled against proton_amd/host/cuthho.hpp only: no Eigen, no HIP headers.
//   usage: cuthho_driver -k <degree> -M <Nx> -N <Ny> -r <refsteps> (-f | -i)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

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

static RealType grad_error(const mesh_type &msh, const mesh_type::cell_type &cl, const RealType *dofs, size_t cd, element_location where)
{
    // sum_qp w |grad u_exact - grad u_T|^2 over the `where` part of the cell (cuthho_square.cpp:1036-1046)
    RealType acc = 0.0;
    const auto bar = barycenter(msh, cl);
    const auto h = diameter(msh, cl);
    for (auto &qp : integrate(msh, cl, 2 * cd, where)) {
        const double bx = (qp.first.x() - bar.x()) / (0.5 * h), by = (qp.first.y() - bar.y()) / (0.5 * h);
        double gx = 0.0, gy = 0.0;
        size_t pos = 0;
        for (size_t kk = 0; kk <= cd; kk++)                                       // bases.hpp:142-184
            for (size_t ii = 0; ii <= kk; ii++, pos++) {
                if (pos == 0) continue;
                const double px = (double)(kk - ii), py = (double)ii, u = dofs[pos];
                if (kk - ii > 0) gx += u * px * (2.0 / h) * std::pow(bx, px - 1) * std::pow(by, py);
                if (ii > 0) gy += u * py * (2.0 / h) * std::pow(bx, px) * std::pow(by, py - 1);
            }
        const double sx = M_PI * std::cos(M_PI * qp.first.x()) * std::sin(M_PI * qp.first.y());
        const double sy = M_PI * std::sin(M_PI * qp.first.x()) * std::cos(M_PI * qp.first.y());
        acc += qp.second * ((sx - gx) * (sx - gx) + (sy - gy) * (sy - gy));
    }
    return acc;
}

int main(int argc, char **argv)
{
    size_t degree = 0, int_refsteps = 4;                                          // cuthho_square.cpp:1944-1945
    mesh_init_params<RealType> mip;
    mip.Nx = 5; mip.Ny = 5;                                                       // :1954-1955
    bool solve_interface = false, solve_fictdom = false;
    int ch;
    while ((ch = getopt(argc, argv, "k:M:N:r:if")) != -1) {                      // :1964-2011 (-D -A -d not carried)
        switch (ch) {
        case 'k': degree = std::atoi(optarg); break;
        case 'M': mip.Nx = std::atoi(optarg); break;
        case 'N': mip.Ny = std::atoi(optarg); break;
        case 'r': int_refsteps = std::atoi(optarg); break;
        case 'i': solve_interface = true; break;
        case 'f': solve_fictdom = true; break;
        default: std::printf("wrong arguments\n"); return 1;
        }
    }
    mesh_type msh(mip);
    auto level_set_function = circle_level_set<RealType>(0.35, 0.5, 0.5);        // :2029-2030

    detect_node_position(msh, level_set_function);                                // :2036-2052
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

    hho_degree_info hdi(degree + 1, degree);                                      // :871, :1662
    const size_t cd = hdi.cell_degree(), cbs = (cd + 2) * (cd + 1) / 2, fbs = degree + 1, nfdofs = 4 * fbs;
    size_t ncut = 0;
    for (auto &cl : msh.cells) ncut += is_cut(msh, cl) ? 1 : 0;

    if (solve_interface) {                                                        // run_cuthho_interface :1625-1846
        params<RealType> parms;
        auto assembler = make_interface_assembler(msh, hdi);
        for (auto &cl : msh.cells) {                                              // :1664-1716
            if (location(msh, cl) != element_location::ON_INTERFACE) {
                const RealType kappa = location(msh, cl) == element_location::IN_NEGATIVE_SIDE ? parms.kappa_1 : parms.kappa_2;
                auto gr = make_hho_laplacian(msh, cl, hdi);
                auto lc = gr.second * kappa + make_hho_naive_stabilization(msh, cl, hdi);
                auto f = make_rhs(msh, cl, hdi.cell_degree(), rhs_fun);
                assembler.assemble(msh, cl, lc, f, bcs_fun);
            } else {
                auto gr = make_hho_laplacian_interface(msh, cl, level_set_function, hdi, parms);
                auto lc = gr.second;
                auto stab_n = make_hho_cut_stabilization(msh, cl, hdi, element_location::IN_NEGATIVE_SIDE) * parms.kappa_1;
                auto stab_p = make_hho_cut_stabilization(msh, cl, hdi, element_location::IN_POSITIVE_SIDE) * parms.kappa_2;
                lc.add_to_block(0, 0, stab_n.block(0, 0, cbs, cbs));                                   // :1697-1700
                lc.add_to_block(0, 2 * cbs, stab_n.block(0, cbs, cbs, nfdofs));
                lc.add_to_block(2 * cbs, 0, stab_n.block(cbs, 0, nfdofs, cbs));
                lc.add_to_block(2 * cbs, 2 * cbs, stab_n.block(cbs, cbs, nfdofs, nfdofs));
                lc.add_to_block(cbs, cbs, stab_p.block(0, 0, cbs, cbs));                               // :1702-1705
                lc.add_to_block(cbs, 2 * cbs + nfdofs, stab_p.block(0, cbs, cbs, nfdofs));
                lc.add_to_block(2 * cbs + nfdofs, cbs, stab_p.block(cbs, 0, nfdofs, cbs));
                lc.add_to_block(2 * cbs + nfdofs, 2 * cbs + nfdofs, stab_p.block(cbs, cbs, nfdofs, nfdofs));
                proton_amd::dense_matrix<RealType> f(2 * cbs, 1);
                auto fn = make_rhs(msh, cl, hdi.cell_degree(), element_location::IN_NEGATIVE_SIDE, rhs_fun);
                auto fp = make_rhs(msh, cl, hdi.cell_degree(), element_location::IN_POSITIVE_SIDE, rhs_fun);
                for (size_t i = 0; i < cbs; ++i) { f(i) = fn(i); f(cbs + i) = fp(i); }
                assembler.assemble_cut(msh, cl, lc, f);
            }
        }
        assembler.finalize();
        std::vector<RealType> sol;                                                // :1737-1743: the reference's conjugated_gradient
        cg_params<RealType> cgp;
        cgp.max_iter = assembler.LHS.rows();
        cgp.apply_preconditioner = true;
        size_t iters = 0;
        conjugated_gradient(assembler.LHS, assembler.RHS, sol, cgp, &iters);
        RealType H1_error = 0.0;                                                  // :1762-1833
        for (auto &cl : msh.cells) {
            if (is_cut(msh, cl)) {
                for (auto where : {element_location::IN_NEGATIVE_SIDE, element_location::IN_POSITIVE_SIDE}) {
                    auto locdata = assembler.take_local_data(msh, cl, sol, bcs_fun, where);
                    H1_error += grad_error(msh, cl, locdata.data(), cd, where);
                }
            } else {
                auto locdata = assembler.take_local_data(msh, cl, sol, bcs_fun, element_location::IN_POSITIVE_SIDE);
                H1_error += grad_error(msh, cl, locdata.data(), cd, location(msh, cl));
            }
        }
        std::printf("interface N %zu k %zu r %zu cut_cells %zu system %zu cg_iters %zu energy_error %.10e\n", (size_t)mip.Nx, degree,
                    int_refsteps, ncut, assembler.LHS.rows(), iters, std::sqrt(H1_error));
    }

    if (solve_fictdom) {                                                          // run_cuthho_fictdom :806-1080
        const element_location where = element_location::IN_NEGATIVE_SIDE;
        auto assembler = make_assembler(msh, hdi);
        for (auto &cl : msh.cells) {                                              // :883-900
            auto gr = make_hho_laplacian(msh, cl, level_set_function, hdi, where);
            auto stab = make_hho_cut_stabilization(msh, cl, hdi, where);
            auto lc = gr.second + stab;
            auto f = make_rhs(msh, cl, hdi.cell_degree(), rhs_fun, where, level_set_function, bcs_fun);
            assembler.assemble(msh, cl, lc, f, bcs_fun);
        }
        assembler.finalize();
        std::vector<RealType> sol;
        const size_t iters = pcg(assembler.LHS, assembler.RHS, sol, 1e-13, 4 * assembler.LHS.rows());
        RealType H1_error = 0.0;                                                  // :1030-1049
        for (auto &cl : msh.cells) {
            if (location(msh, cl) == element_location::IN_POSITIVE_SIDE) continue;
            H1_error += grad_error(msh, cl, sol.data() + offset(msh, cl) * cbs, cd, where);
        }
        std::printf("fictdom N %zu k %zu r %zu cut_cells %zu system %zu cg_iters %zu energy_error %.10e\n", (size_t)mip.Nx, degree,
                    int_refsteps, ncut, assembler.LHS.rows(), iters, std::sqrt(H1_error));
    }
    return 0;
}
