// boundary_driver.cpp -- fidelity of the drop-in header proton_amd/host/hho.hpp to the reference's per-cell interface in
// the corners a batched implementation can get wrong:
//   1. project_function with two different TEMPORARY functors in a loop: each call projects the functor it was given;
//   2. a mesh edited in place between two sweeps (one node displaced): the operators of the new geometry are served;
//   3. make_hho_fancy_stabilization honours its `reconstruction` argument: the cell's own operator is accepted, any
//      other matrix is refused (hho.hpp:155-159, 184-190);
//   4. two proton_amd::device objects in one process, selected with device_scope, hold two different meshes side by side;
//   5. the per-cell loop body over 65 536 cells stays linear in the number of cells (time bound);
//   6. two Dirichlet functors that differ on ONE boundary edge only, given to the same assembler in two consecutive sweeps: the
//      second sweep's boundary coefficients are those of the second functor, whichever edge it is (the cache of
//      boundary data was re-checked at 8 probe points only).
// Prints "check <n> ok" per item; exit code 0 iff all pass.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../proton_amd/host/hho.hpp"

using T = double;
using mesh_type = quad_mesh<T>;

static mesh_type make_mesh(size_t N)
{
    mesh_init_params<T> mip;
    mip.Nx = N; mip.Ny = N;
    return mesh_type(mip);
}

static T max_abs_diff(const proton_amd::dense_matrix<T> &a, const proton_amd::dense_matrix<T> &b)
{
    T m = 0;
    for (size_t i = 0; i < a.rows() * a.cols(); ++i) m = std::max(m, std::abs(a.data()[i] - b.data()[i]));
    return m;
}

int main()
{
    int failures = 0;
    auto report = [&](int n, bool ok) { std::printf("check %d %s\n", n, ok ? "ok" : "FAILED"); if (!ok) ++failures; };
    hho_degree_info hdi(2, 1);
    mesh_type msh = make_mesh(6);

    // 1. temporaries: the k-th call gets the projection of x -> k + x, not the first functor's
    {
        bool ok = true;
        const auto &cl = msh.cells[7];
        for (int k = 1; k <= 3; ++k) {
            auto proj = project_function(msh, cl, hdi, [k](const mesh_type::point_type &pt) -> T { return k + pt.x(); });
            // cell dof 0 of the scaled monomial basis is the value at the barycenter for an affine function
            const auto bar = barycenter(msh, cl);
            ok = ok && std::abs(proj(0) - (k + bar.x())) < 1e-12;
        }
        // the batched variant agrees with the per-cell one
        auto f = [](const mesh_type::point_type &pt) -> T { return std::sin(pt.x()) * pt.y(); };
        auto all = project_function_all(msh, hdi, f);
        for (size_t c = 0; c < msh.cells.size(); c += 5) {
            auto one = project_function(msh, msh.cells[c], hdi, f);
            for (size_t i = 0; i < one.rows(); ++i) ok = ok && std::abs(one(i) - all[c * one.rows() + i]) < 1e-14;
        }
        report(1, ok);
    }

    // 2. in-place edit of one node between two sweeps
    {
        const auto &cl = msh.cells[14];
        auto before = make_hho_laplacian(msh, cl, hdi);
        for (auto &c : msh.cells) (void)make_hho_laplacian(msh, c, hdi);            // a full sweep
        const size_t pid = cl.ptids[2];                                               // an interior node of that cell
        msh.points[pid] = mesh_type::point_type(msh.points[pid].x() + 0.02, msh.points[pid].y() - 0.015);
        T changed = 0;
        proton_amd::dense_matrix<T> after_data;
        for (auto &c : msh.cells) {                                                   // the next sweep starts at cell 0
            auto gr = make_hho_laplacian(msh, c, hdi);
            if (&c == &cl) { changed = max_abs_diff(gr.second, before.second); after_data = gr.second; }
        }
        // the answer is that of a mesh built with the displaced node from the start
        mesh_type fresh = make_mesh(6);
        fresh.points[pid] = msh.points[pid];
        auto ref = make_hho_laplacian(fresh, fresh.cells[14], hdi);
        report(2, changed > 1e-3 && max_abs_diff(after_data, ref.second) == 0);
    }

    // 3. the reconstruction argument
    {
        const auto &cl = msh.cells[3];
        auto gr = make_hho_laplacian(msh, cl, hdi);
        bool ok = true;
        try { (void)make_hho_fancy_stabilization(msh, cl, gr.first, hdi); } catch (...) { ok = false; }
        auto other = make_hho_laplacian(msh, msh.cells[14], hdi);                      // a different cell's operator (node 14 was displaced)
        bool refused = false;
        try { (void)make_hho_fancy_stabilization(msh, cl, other.first, hdi); } catch (const std::invalid_argument &) { refused = true; }
        report(3, ok && refused);
    }

    // 4. two devices (two contexts; the same GPU serves both on a one-GPU box)
    {
        proton_amd::device a(0), b(0);
        mesh_type coarse = make_mesh(4), fine = make_mesh(8);
        {   // (square cells of any size have the same local matrices in the scaled bases: make cell 5 of `fine` a general quad)
            const size_t pid = fine.cells[5].ptids[2];
            fine.points[pid] = mesh_type::point_type(fine.points[pid].x() + 0.01, fine.points[pid].y() + 0.02);
        }
        proton_amd::dense_matrix<T> da, db, da2;
        { proton_amd::device_scope on(a); da = make_hho_laplacian(coarse, coarse.cells[5], hdi).second; }
        { proton_amd::device_scope on(b); db = make_hho_laplacian(fine, fine.cells[5], hdi).second; }
        { proton_amd::device_scope on(a); da2 = make_hho_laplacian(coarse, coarse.cells[5], hdi).second; }      // a still holds `coarse`
        // reference answers from the default device
        auto ra = make_hho_laplacian(coarse, coarse.cells[5], hdi).second;
        auto rb = make_hho_laplacian(fine, fine.cells[5], hdi).second;
        report(4, max_abs_diff(da, ra) == 0 && max_abs_diff(db, rb) == 0 && max_abs_diff(da2, ra) == 0 && max_abs_diff(ra, rb) > 1e-6);
    }

    // 5. the reference's loop body (make_hho_laplacian, stabilization, make_rhs on the same cell) over a 256 x 256 mesh through the
    //    per-cell API: the mesh's coordinates are re-hashed once per sweep, not once per call (O(cells x points))
    {
        mesh_type big = make_mesh(256);
        auto rhs_fun = [](const mesh_type::point_type &pt) -> T { return std::sin(M_PI * pt.x()) * std::sin(M_PI * pt.y()); };
        const auto t0 = std::chrono::steady_clock::now();
        T acc = 0;
        for (auto &cl : big.cells) {
            auto gr = make_hho_laplacian(big, cl, hdi);
            auto stab = make_hho_fancy_stabilization(big, cl, gr.first, hdi);
            auto f = make_rhs(big, cl, hdi.cell_degree(), rhs_fun);
            acc += gr.second(0, 0) + stab(1, 1) + f(0);
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("per-cell sweep of %zu cells: %.2f s (checksum %.6e)\n", big.cells.size(), sec, acc);
        report(5, std::isfinite(acc) && sec < 20.0);
    }
    // 6. boundary data follow the functor on every edge
    {
        mesh_type m = make_mesh(6);
        auto g1 = [](const mesh_type::point_type &pt) -> T { return 1.0 + pt.x() + 2.0 * pt.y(); };
        bool ok = true;
        size_t edges = 0;
        for (size_t f = 0; f < m.faces.size(); ++f) {
            if (!m.faces[f].is_boundary) continue;
            ++edges;
            const auto a = m.points[m.faces[f].ptids[0]], b = m.points[m.faces[f].ptids[1]];
            const T x0 = std::min(a.x(), b.x()) - 1e-9, x1 = std::max(a.x(), b.x()) + 1e-9;
            const T y0 = std::min(a.y(), b.y()) - 1e-9, y1 = std::max(a.y(), b.y()) + 1e-9;
            // g2 = g1 except on the open edge f (its end points belong to the neighbouring edges too, so leave them alone)
            auto g2 = [=](const mesh_type::point_type &pt) -> T {
                const bool inside = pt.x() > x0 && pt.x() < x1 && pt.y() > y0 && pt.y() < y1;
                const bool at_end = (std::abs(pt.x() - a.x()) + std::abs(pt.y() - a.y()) < 1e-9) ||
                                    (std::abs(pt.x() - b.x()) + std::abs(pt.y() - b.y()) < 1e-9);
                return 1.0 + pt.x() + 2.0 * pt.y() + ((inside && !at_end) ? 5.0 : 0.0);
            };
            auto assm = make_assembler(m, hdi);
            auto fresh = make_assembler(m, hdi), plain = make_assembler(m, hdi);      // (each of these two sees ONE functor only)
            std::vector<T> sol(assm.RHS.size(), T(0));
            T seen = 0;
            for (auto &cl : m.cells) (void)assm.take_local_data(m, cl, sol, g1);              // sweep 1 fills the cache with g1
            for (auto &cl : m.cells) {                                                         // sweep 2: g2
                auto got = assm.take_local_data(m, cl, sol, g2);
                auto want = fresh.take_local_data(m, cl, sol, g2);
                auto with_g1 = plain.take_local_data(m, cl, sol, g1);
                ok = ok && max_abs_diff(got, want) == 0;
                seen = std::max(seen, max_abs_diff(want, with_g1));
            }
            ok = ok && seen > 1.0;                                                             // the bump reached some cell
        }
        std::printf("boundary data re-checked on %zu edges\n", edges);
        report(6, ok && edges == 24);
    }
    return failures ? 1 : 0;
}
