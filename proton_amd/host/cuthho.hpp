// proton_amd/host/cuthho.hpp -- C++ host side of the cutHHO part of the path, under the reference's
// names, for drivers written like apps/cuthho/cuthho_square.cpp:
//   circle_level_set / line_level_set          apps/cuthho/cuthho_square.cpp:56-124
//   element_location, location(), is_cut()     src/methods/cuthho_bits/cuthho_mesh.hpp:31-100
//   cuthho_poly_mesh                           src/methods/cuthho (alias of the generator mesh + tags)
//   detect_node_position, detect_cut_faces, move_nodes, detect_cut_cells, refine_interface
//                                              src/methods/cuthho_bits/cuthho_geom.hpp:118-161,275-340,466-543,609-673
//   make_hho_laplacian(msh, cl, level_set, hdi, where)          cuthho_square.cpp:308-388
//   make_hho_cut_stabilization(msh, cl, hdi, where)             cuthho_square.cpp:566-621
//   make_rhs(msh, cl, degree, f, where, level_set, bcs)         cuthho_square.cpp:623-666
//   make_hho_laplacian_interface(msh, cl, level_set, hdi, parms) cuthho_square.cpp:390-502
//   integrate(msh, cl, degree, where) for cut cells             cuthho_geom.hpp:798-815
// The preprocessing steps are ONE call of the library (pa_cut_preprocess: host C++ inside
// libproton_amd.so); the reference's step functions are kept so that main()-shaped drivers compile:
// they record what was asked for and refine_interface(), the last of them, runs the fused pass
// with node displacement (the reference's default, -D).  Operators are served from device batches
// like in hho.hpp; there is no CPU fallback.
#pragma once

#include "hho.hpp"

enum class element_location { IN_NEGATIVE_SIDE = PA_LOC_NEGATIVE, IN_POSITIVE_SIDE = PA_LOC_POSITIVE,
                              ON_INTERFACE = PA_LOC_ON_INTERFACE, UNDEF = 3 };

template <typename T>
struct circle_level_set {                               // cuthho_square.cpp:56-89
    T radius, alpha, beta;
    circle_level_set(T r, T a, T b) : radius(r), alpha(a), beta(b) {}
    T operator()(const point<T, 2> &pt) const
    {
        return (pt.x() - alpha) * (pt.x() - alpha) + (pt.y() - beta) * (pt.y() - beta) - radius * radius;
    }
    proton_amd::dense_matrix<T> normal(const point<T, 2> &pt) const
    {
        proton_amd::dense_matrix<T> ret(2, 1);
        ret(0) = 2 * pt.x() - 2 * alpha; ret(1) = 2 * pt.y() - 2 * beta;
        const T n = std::sqrt(ret(0) * ret(0) + ret(1) * ret(1));
        ret(0) /= n; ret(1) /= n;
        return ret;
    }
    pa_level_set c_abi() const { return pa_level_set{0, radius, alpha, beta, 0.0}; }
};

template <typename T>
struct line_level_set {                                 // cuthho_square.cpp:91-124
    T cut_y;
    explicit line_level_set(T cy) : cut_y(cy) {}
    T operator()(const point<T, 2> &pt) const { return pt.y() - cut_y; }
    proton_amd::dense_matrix<T> normal(const point<T, 2> &) const
    {
        proton_amd::dense_matrix<T> ret(2, 1);
        ret(0) = 0; ret(1) = 1;
        return ret;
    }
    pa_level_set c_abi() const { return pa_level_set{1, 0.0, 0.0, 0.0, cut_y}; }
};

template <typename T>
struct params {                                         // cuthho_square.cpp:293-299
    T kappa_1, kappa_2, eta;
    params() : kappa_1(1.0), kappa_2(1.0), eta(5.0) {}
};

// the generator mesh of basic_mesh.hpp:321-403 (same numbering as quad_mesh) plus the cutHHO tags
template <typename T>
struct cuthho_poly_mesh : public quad_mesh<T> {
    static constexpr int pa_quadrature = PA_QUAD_FAN;   // integrate() of a polygonal cell: triangle fan, quadratures.hpp:377-402
    using typename quad_mesh<T>::cell_type;
    using typename quad_mesh<T>::face_type;
    std::vector<element_location> node_tags, face_tags, cell_tags;
    std::vector<int32_t> cut_index;                     // cell -> position among the cut cells, -1 otherwise
    size_t num_cut_cells = 0, interface_refsteps = 0;
    bool preprocessed = false;
    pa_level_set level_set{};
    explicit cuthho_poly_mesh(const mesh_init_params<T> &mip) : quad_mesh<T>(mip) {}
};

namespace proton_amd {

// cuthho_square.cpp:2036-2052 (default path: node displacement) in one library call
template <typename T, typename LevelSet>
void cuthho_preprocess(cuthho_poly_mesh<T> &msh, const LevelSet &level_set_function, size_t refsteps)
{
    auto &dev = device::instance();
    const auto &p = msh.params;
    const pa_level_set ls = level_set_function.c_abi();
    const int st = pa_cut_preprocess(dev.ctx(), p.Nx, p.Ny, p.min_x, p.max_x, p.min_y, p.max_y, &ls, (int)refsteps);
    if (st != PA_OK) throw std::logic_error(std::string("cutHHO preprocessing: ") + pa_last_error(dev.ctx()));   // the reference throws logic_error
    const size_t np = msh.points.size(), nc = msh.cells.size(), nf = msh.faces.size();
    std::vector<int8_t> nl(np), fl(nf), cl(nc);
    std::vector<double> pts(2 * np);
    msh.cut_index.assign(nc, -1);
    dev.check(pa_cut_query(dev.ctx(), &msh.num_cut_cells, cl.data(), msh.cut_index.data()), "pa_cut_query");
    dev.check(pa_cut_query_tags(dev.ctx(), nl.data(), fl.data(), pts.data()), "pa_cut_query_tags");
    msh.node_tags.resize(np); msh.face_tags.resize(nf); msh.cell_tags.resize(nc);
    for (size_t i = 0; i < np; ++i) { msh.node_tags[i] = (element_location)nl[i]; msh.points[i] = point<T, 2>(pts[2 * i], pts[2 * i + 1]); }
    for (size_t i = 0; i < nf; ++i) msh.face_tags[i] = (element_location)fl[i];
    for (size_t i = 0; i < nc; ++i) msh.cell_tags[i] = (element_location)cl[i];
    msh.level_set = ls; msh.interface_refsteps = refsteps; msh.preprocessed = true;
    batch_cache<cuthho_poly_mesh<T>>::instance().adopt(msh);     // the context holds exactly this mesh now
}

// device batches of the cut cells, one per (mesh, face degree, side)
template <typename T>
struct cut_batch {
    const cuthho_poly_mesh<T> *msh = nullptr;
    int face_deg = -1, where = -1;
    pa_sizes sz{};
    std::vector<double> oper, data, stab;               // ncut x (rbs x msize), ncut x msize^2 (column-major)
    std::vector<uint32_t> cell_off, ir_off;             // quadrature lists for caller-sampled functions
    std::vector<double> cell_xyw, ir_xyw;
    const void *rhs_f = nullptr, *rhs_bcs = nullptr;
    std::vector<double> rhs;                            // ncut x cbs for the functors above

    static cut_batch &get(const cuthho_poly_mesh<T> &m, const hho_degree_info &hdi, element_location where)
    {
        static cut_batch slot[2];
        cut_batch &b = slot[(int)where];
        const int fd = (int)hdi.face_degree();
        if (b.msh == &m && b.face_deg == fd) return b;
        if (!m.preprocessed) throw std::logic_error("cutHHO mesh not preprocessed");
        if (hdi.cell_degree() != hdi.face_degree() + 1) throw std::invalid_argument("cut operators need hho_degree_info(k+1, k)");
        auto &dev = device::instance();
        b = cut_batch();
        dev.check(pa_sizes_for(hdi.c_abi(), PA_QUAD_FAN, &b.sz), "pa_sizes_for");
        const size_t n = m.num_cut_cells, mm = (size_t)b.sz.msize * b.sz.msize, om = (size_t)b.sz.rbs * b.sz.msize;
        device_buffer<double> d_oper(n * om + 1), d_data(n * mm + 1), d_stab(n * mm + 1);
        dev.check(pa_cut_local_ops_batch(dev.ctx(), fd, &m.level_set, (int)where, PA_FN_ONE, PA_FN_ONE, d_oper.get(), d_data.get(),
                                         d_stab.get(), nullptr, nullptr, nullptr), "pa_cut_local_ops_batch");
        b.oper.resize(n * om); b.data.resize(n * mm); b.stab.resize(n * mm);
        if (n) { d_oper.download(b.oper.data(), n * om); d_data.download(b.data.data(), n * mm); d_stab.download(b.stab.data(), n * mm); }
        size_t cnt = 0;
        b.cell_off.assign(n + 1, 0); b.ir_off.assign(n + 1, 0);
        dev.check(pa_cut_quadrature_points(dev.ctx(), fd, (int)where, 0, nullptr, nullptr, &cnt), "pa_cut_quadrature_points");
        b.cell_xyw.resize(3 * cnt);
        dev.check(pa_cut_quadrature_points(dev.ctx(), fd, (int)where, 0, b.cell_off.data(), b.cell_xyw.data(), &cnt), "pa_cut_quadrature_points");
        dev.check(pa_cut_quadrature_points(dev.ctx(), fd, (int)where, 2, nullptr, nullptr, &cnt), "pa_cut_quadrature_points");
        b.ir_xyw.resize(3 * cnt);
        dev.check(pa_cut_quadrature_points(dev.ctx(), fd, (int)where, 2, b.ir_off.data(), b.ir_xyw.data(), &cnt), "pa_cut_quadrature_points");
        b.msh = &m; b.face_deg = fd; b.where = (int)where;
        return b;
    }
};

}  // namespace proton_amd

// ---- the reference's preprocessing steps (cuthho_square.cpp:2036-2052) -------------------------
template <typename T, typename Function>
void detect_node_position(cuthho_poly_mesh<T> &, const Function &) {}
template <typename T, typename Function>
void detect_cut_faces(cuthho_poly_mesh<T> &, const Function &) {}
template <typename T, typename Function>
void move_nodes(cuthho_poly_mesh<T> &, const Function &) {}
template <typename T, typename Function>
void detect_cut_cells(cuthho_poly_mesh<T> &, const Function &) {}
template <typename T, typename Function>
void refine_interface(cuthho_poly_mesh<T> &msh, const Function &level_set_function, size_t levels)
{
    proton_amd::cuthho_preprocess(msh, level_set_function, levels);
}

template <typename T>
element_location location(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl)
{
    return msh.cell_tags.at(offset(msh, cl));
}
template <typename T>
element_location location(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::face_type &fc)
{
    return msh.face_tags.at(offset(msh, fc));
}
template <typename T>
bool is_cut(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl)
{
    return location(msh, cl) == element_location::ON_INTERFACE;
}

// integrate(msh, cl, degree, where) cuthho_geom.hpp:798-815.  Cut cells: the library's list at degree
// 2*(k+1) (the only degree the drivers use, with hdi(k+1,k)); uncut cells on the `where` side: the fan rule.
template <typename T>
std::vector<std::pair<point<T, 2>, T>> integrate(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl,
                                                 size_t degree, element_location where)
{
    std::vector<std::pair<point<T, 2>, T>> ret;
    const size_t c = offset(msh, cl);
    if (!is_cut(msh, cl)) {
        if (location(msh, cl) != where) return ret;
        int nq = 0;
        const auto &xyw = proton_amd::batch_cache<cuthho_poly_mesh<T>>::instance().cell_qpoints(msh, (int)degree, PA_QUAD_FAN, nq);
        for (int q = 0; q < nq; ++q) ret.emplace_back(point<T, 2>(xyw[(c * nq + q) * 3], xyw[(c * nq + q) * 3 + 1]), xyw[(c * nq + q) * 3 + 2]);
        return ret;
    }
    if (degree == 0 || degree % 2) throw std::invalid_argument("cut integrate: degree 2*(k+1) expected");
    hho_degree_info hdi(degree / 2, degree / 2 - 1);
    auto &b = proton_amd::cut_batch<T>::get(msh, hdi, where);
    const size_t cc = (size_t)msh.cut_index[c];
    for (uint32_t q = b.cell_off[cc]; q < b.cell_off[cc + 1]; ++q)
        ret.emplace_back(point<T, 2>(b.cell_xyw[3 * q], b.cell_xyw[3 * q + 1]), b.cell_xyw[3 * q + 2]);
    return ret;
}

// cuthho_square.cpp:308-388: rbs x msize for cut cells, (rbs-1) x msize otherwise (:316-317)
template <typename T, typename Function>
std::pair<proton_amd::dense_matrix<T>, proton_amd::dense_matrix<T>>
make_hho_laplacian(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl, const Function & /*level_set_function*/,
                   hho_degree_info di, element_location where)
{
    if (!is_cut(msh, cl)) return make_hho_laplacian(msh, cl, di);
    auto &b = proton_amd::cut_batch<T>::get(msh, di, where);
    const size_t cc = (size_t)msh.cut_index[offset(msh, cl)];
    return std::make_pair(proton_amd::copy_cell<T>(b.oper, cc, b.sz.rbs, b.sz.msize), proton_amd::copy_cell<T>(b.data, cc, b.sz.msize, b.sz.msize));
}

// cuthho_square.cpp:566-621
template <typename T>
proton_amd::dense_matrix<T> make_hho_cut_stabilization(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl,
                                                       const hho_degree_info &di, element_location where)
{
    if (!is_cut(msh, cl)) return make_hho_naive_stabilization(msh, cl, di);
    auto &b = proton_amd::cut_batch<T>::get(msh, di, where);
    return proton_amd::copy_cell<T>(b.stab, (size_t)msh.cut_index[offset(msh, cl)], b.sz.msize, b.sz.msize);
}

// cuthho_square.cpp:623-666.  The functors are sampled on the host at the library's quadrature
// points of ALL cut cells at the first call with a given (f, bcs) pair; the sums run on the device.
template <typename T, typename F1, typename F2, typename F3>
proton_amd::dense_matrix<T> make_rhs(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl, size_t degree,
                                     const F1 &f, element_location where, const F2 & /*level_set_function*/, const F3 &bcs)
{
    const size_t cbs = (degree + 2) * (degree + 1) / 2;
    if (location(msh, cl) == where) return make_rhs(msh, cl, degree, f);                   // :628-629
    if (!is_cut(msh, cl)) return proton_amd::dense_matrix<T>(cbs, 1);                     // :659-664
    hho_degree_info hdi(degree, degree - 1);
    auto &b = proton_amd::cut_batch<T>::get(msh, hdi, where);
    if (b.rhs_f != (const void *)&f || b.rhs_bcs != (const void *)&bcs) {
        auto &dev = proton_amd::device::instance();
        const size_t nq = b.cell_xyw.size() / 3, ni = b.ir_xyw.size() / 3;
        std::vector<double> fv(nq + 1), bv(ni + 1);
        for (size_t q = 0; q < nq; ++q) fv[q] = f(point<T, 2>(b.cell_xyw[3 * q], b.cell_xyw[3 * q + 1]));
        for (size_t q = 0; q < ni; ++q) bv[q] = bcs(point<T, 2>(b.ir_xyw[3 * q], b.ir_xyw[3 * q + 1]));
        proton_amd::device_buffer<double> d_f(nq + 1), d_b(ni + 1), d_r(msh.num_cut_cells * cbs + 1);
        d_f.upload(fv.data(), nq + 1); d_b.upload(bv.data(), ni + 1);
        dev.check(pa_cut_rhs_sampled_batch(dev.ctx(), b.face_deg, &msh.level_set, (int)where, d_f.get(), d_b.get(), d_r.get()),
                  "pa_cut_rhs_sampled_batch");
        b.rhs.resize(msh.num_cut_cells * cbs);
        if (!b.rhs.empty()) d_r.download(b.rhs.data(), b.rhs.size());
        b.rhs_f = (const void *)&f; b.rhs_bcs = (const void *)&bcs;
    }
    return proton_amd::copy_cell<T>(b.rhs, (size_t)msh.cut_index[offset(msh, cl)], cbs, 1);
}

// cuthho_square.cpp:390-502 (built-in source term ids are not involved: operator only)
template <typename T, typename Function>
std::pair<proton_amd::dense_matrix<T>, proton_amd::dense_matrix<T>>
make_hho_laplacian_interface(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl, const Function & /*level_set_function*/,
                             hho_degree_info di, const params<T> &parms = params<T>())
{
    if (!is_cut(msh, cl)) throw std::invalid_argument("The cell is not cut");          // :397-398
    struct cached { const void *msh = nullptr; int fd = -1; double k1 = 0, k2 = 0, eta = 0; std::vector<double> oper, data; pa_sizes sz{}; };
    static cached cache;
    const int fd = (int)di.face_degree();
    if (cache.msh != (const void *)&msh || cache.fd != fd || cache.k1 != parms.kappa_1 || cache.k2 != parms.kappa_2 || cache.eta != parms.eta) {
        auto &dev = proton_amd::device::instance();
        dev.check(pa_sizes_for(di.c_abi(), PA_QUAD_FAN, &cache.sz), "pa_sizes_for");
        const size_t n = msh.num_cut_cells, m2 = 2 * (size_t)cache.sz.msize, n2 = 2 * (size_t)cache.sz.rbs;
        proton_amd::device_buffer<double> d_oper(n * n2 * m2 + 1), d_data(n * m2 * m2 + 1);
        const pa_interface_params ip{parms.kappa_1, parms.kappa_2, parms.eta};
        dev.check(pa_cut_interface_ops_batch(dev.ctx(), fd, &msh.level_set, &ip, PA_FN_ONE, d_oper.get(), d_data.get(), nullptr, nullptr, nullptr),
                  "pa_cut_interface_ops_batch");
        cache.oper.resize(n * n2 * m2); cache.data.resize(n * m2 * m2);
        if (n) { d_oper.download(cache.oper.data(), cache.oper.size()); d_data.download(cache.data.data(), cache.data.size()); }
        cache.msh = (const void *)&msh; cache.fd = fd; cache.k1 = parms.kappa_1; cache.k2 = parms.kappa_2; cache.eta = parms.eta;
    }
    const size_t cc = (size_t)msh.cut_index[offset(msh, cl)], m2 = 2 * (size_t)cache.sz.msize, n2 = 2 * (size_t)cache.sz.rbs;
    return std::make_pair(proton_amd::copy_cell<T>(cache.oper, cc, n2, m2), proton_amd::copy_cell<T>(cache.data, cc, m2, m2));
}

// make_rhs(msh, cl, degree, where, f)  src/methods/cuthho_bits/cuthho_utils.hpp:65-84: the source term
// over the `where` part of a cell (no boundary term).  Cut cells: the library's cut quadrature with
// host-sampled values, sums on the device (the Nitsche part of pa_cut_rhs_sampled_batch gets zeros).
template <typename T, typename Function>
proton_amd::dense_matrix<T> make_rhs(const cuthho_poly_mesh<T> &msh, const typename cuthho_poly_mesh<T>::cell_type &cl, size_t degree,
                                     element_location where, const Function &f)
{
    const size_t cbs = (degree + 2) * (degree + 1) / 2;
    if (!is_cut(msh, cl)) return location(msh, cl) == where ? make_rhs(msh, cl, degree, f) : proton_amd::dense_matrix<T>(cbs, 1);
    struct cached { const void *msh = nullptr, *fn = nullptr; size_t degree = 0; std::vector<double> rhs[2]; };
    static cached cache;
    hho_degree_info hdi(degree, degree - 1);
    if (cache.msh != (const void *)&msh || cache.fn != (const void *)&f || cache.degree != degree) {
        auto &dev = proton_amd::device::instance();
        for (int side = 0; side < 2; ++side) {
            auto &b = proton_amd::cut_batch<T>::get(msh, hdi, (element_location)side);
            const size_t nq = b.cell_xyw.size() / 3, ni = b.ir_xyw.size() / 3;
            std::vector<double> fv(nq + 1), zeros(ni + 1, 0.0);
            for (size_t q = 0; q < nq; ++q) fv[q] = f(point<T, 2>(b.cell_xyw[3 * q], b.cell_xyw[3 * q + 1]));
            proton_amd::device_buffer<double> d_f(nq + 1), d_z(ni + 1), d_r(msh.num_cut_cells * cbs + 1);
            d_f.upload(fv.data(), nq + 1); d_z.upload(zeros.data(), ni + 1);
            dev.check(pa_cut_rhs_sampled_batch(dev.ctx(), b.face_deg, &msh.level_set, side, d_f.get(), d_z.get(), d_r.get()),
                      "pa_cut_rhs_sampled_batch");
            cache.rhs[side].resize(msh.num_cut_cells * cbs);
            if (!cache.rhs[side].empty()) d_r.download(cache.rhs[side].data(), cache.rhs[side].size());
        }
        cache.msh = (const void *)&msh; cache.fn = (const void *)&f; cache.degree = degree;
    }
    return proton_amd::copy_cell<T>(cache.rhs[(int)where], (size_t)msh.cut_index[offset(msh, cl)], cbs, 1);
}

// interface_assembler  apps/cuthho/cuthho_square.cpp:1091-1443: cut cells and cut faces own two
// blocks of unknowns (negative side first).  Cut cells must not touch the Dirichlet boundary.
template <typename Mesh>
class interface_assembler {
    using T = typename Mesh::coordinate_type;
    hho_degree_info di;
    proton_amd::face_numbering<Mesh> numbering;          // Dirichlet data of the boundary functor
    std::vector<int64_t> cell_table, face_table;         // first block of each cell / non-Dirichlet face (-1: Dirichlet)
    size_t num_all_cells = 0, num_other_faces = 0;
    std::vector<std::tuple<int32_t, int32_t, T>> triplets;

    size_t face_block(const Mesh &msh, size_t face_offset, size_t fbs, bool second) const
    {
        const size_t dup = (second && msh.face_tags[face_offset] == element_location::ON_INTERFACE) ? fbs : 0;     // :1319
        return num_all_cells * numbering.cbs + (size_t)face_table[face_offset] * fbs + dup;
    }

  public:
    proton_amd::sparse_matrix<T> LHS;
    std::vector<T> RHS;

    interface_assembler(const Mesh &msh, hho_degree_info hdi) : di(hdi), numbering(msh, hdi)
    {
        cell_table.resize(msh.cells.size());
        for (size_t c = 0; c < msh.cells.size(); ++c) {                                    // :1142-1150
            cell_table[c] = (int64_t)num_all_cells;
            num_all_cells += msh.cell_tags[c] == element_location::ON_INTERFACE ? 2 : 1;
        }
        face_table.assign(msh.faces.size(), -1);
        for (size_t f = 0; f < msh.faces.size(); ++f) {                                    // :1167-1178
            if (numbering.compress[f] < 0) continue;
            face_table[f] = (int64_t)num_other_faces;
            num_other_faces += msh.face_tags[f] == element_location::ON_INTERFACE ? 2 : 1;
        }
        const size_t system_size = numbering.cbs * num_all_cells + numbering.fbs * num_other_faces;      // :1185
        LHS.nrows = LHS.ncols = system_size;
        RHS.assign(system_size, T(0));
    }

    // :1203-1269
    template <typename Function>
    void assemble(const Mesh &msh, const typename Mesh::cell_type &cl, const proton_amd::dense_matrix<T> &lhs,
                  const proton_amd::dense_matrix<T> &rhs, const Function &dirichlet_bf)
    {
        if (location(msh, cl) == element_location::ON_INTERFACE) throw std::invalid_argument("UNcut cell expected.");
        const size_t cbs = numbering.cbs, fbs = numbering.fbs, ms = cbs + 4 * fbs, c = offset(msh, cl);
        const auto fids = proton_amd::face_offsets(msh, cl);
        std::vector<int64_t> gidx(ms, -1);
        std::vector<T> dir(ms, T(0));
        for (size_t i = 0; i < cbs; ++i) gidx[i] = cell_table[c] * (int64_t)cbs + (int64_t)i;
        for (size_t lf = 0; lf < 4; ++lf)
            for (size_t k = 0; k < fbs; ++k) {
                const size_t l = cbs + lf * fbs + k;
                if (face_table[fids[lf]] >= 0) gidx[l] = (int64_t)(face_block(msh, fids[lf], fbs, false) + k);
                else dir[l] = numbering.dirichlet_data(msh, dirichlet_bf)[fids[lf] * fbs + k];
            }
        for (size_t i = 0; i < ms; ++i) {
            if (gidx[i] < 0) continue;
            for (size_t j = 0; j < ms; ++j) {
                if (gidx[j] >= 0) triplets.emplace_back((int32_t)gidx[i], (int32_t)gidx[j], lhs(i, j));
                else RHS[gidx[i]] -= lhs(i, j) * dir[j];      // term by term, the reference's order
            }
        }
        for (size_t i = 0; i < cbs; ++i) RHS[gidx[i]] += rhs(i);
    }

    // :1271-1354: unknowns [cell-, cell+, faces-, faces+]
    void assemble_cut(const Mesh &msh, const typename Mesh::cell_type &cl, const proton_amd::dense_matrix<T> &lhs,
                      const proton_amd::dense_matrix<T> &rhs)
    {
        if (location(msh, cl) != element_location::ON_INTERFACE) throw std::invalid_argument("Cut cell expected.");
        const size_t cbs = numbering.cbs, fbs = numbering.fbs, ms2 = 2 * (cbs + 4 * fbs), c = offset(msh, cl);
        const auto fids = proton_amd::face_offsets(msh, cl);
        std::vector<int64_t> gidx(ms2);
        for (size_t i = 0; i < 2 * cbs; ++i) gidx[i] = cell_table[c] * (int64_t)cbs + (int64_t)i;
        for (size_t pass = 0; pass < 2; ++pass)
            for (size_t lf = 0; lf < 4; ++lf) {
                if (face_table[fids[lf]] < 0) throw std::invalid_argument("Dirichlet boundary on cut cell not supported.");
                for (size_t k = 0; k < fbs; ++k)
                    gidx[2 * cbs + pass * 4 * fbs + lf * fbs + k] = (int64_t)(face_block(msh, fids[lf], fbs, pass == 1) + k);
            }
        if (lhs.rows() != ms2 || lhs.cols() != ms2) throw std::invalid_argument("interface_assembler::assemble_cut: local matrix size");
        for (size_t i = 0; i < ms2; ++i)
            for (size_t j = 0; j < ms2; ++j) triplets.emplace_back((int32_t)gidx[i], (int32_t)gidx[j], lhs(i, j));
        for (size_t i = 0; i < 2 * cbs; ++i) RHS[gidx[i]] += rhs(i);
    }

    // :1356-1430.  Cell unknowns of the `where` side, then the four faces' unknowns of that side.  (The
    // reference overwrites its face offsets with a formula that ignores the duplicated unknowns --
    // quirk 11 of the survey; its driver only reads the cell part.  Here the face part is the intended one.)
    template <typename Function>
    proton_amd::dense_matrix<T> take_local_data(const Mesh &msh, const typename Mesh::cell_type &cl, const std::vector<T> &solution,
                                                 const Function &dirichlet_bf, element_location where)
    {
        const size_t cbs = numbering.cbs, fbs = numbering.fbs, c = offset(msh, cl);
        const bool second = where == element_location::IN_POSITIVE_SIDE;
        if (location(msh, cl) == element_location::ON_INTERFACE && where != element_location::IN_NEGATIVE_SIDE && !second)
            throw std::invalid_argument("Invalid location");
        const size_t cell_SOL_offset = (size_t)cell_table[c] * cbs + ((second && location(msh, cl) == element_location::ON_INTERFACE) ? cbs : 0);
        const auto fids = proton_amd::face_offsets(msh, cl);
        proton_amd::dense_matrix<T> ret(cbs + 4 * fbs, 1);
        for (size_t i = 0; i < cbs; ++i) ret(i) = solution[cell_SOL_offset + i];
        for (size_t lf = 0; lf < 4; ++lf)
            for (size_t k = 0; k < fbs; ++k)
                ret(cbs + lf * fbs + k) = face_table[fids[lf]] < 0 ? numbering.dirichlet_data(msh, dirichlet_bf)[fids[lf] * fbs + k]
                                                                   : solution[face_block(msh, fids[lf], fbs, second) + k];
        return ret;
    }

    void finalize(void)
    {
        LHS.set_from_triplets(RHS.size(), triplets);
        triplets.clear();
    }
};

template <typename Mesh>
auto make_interface_assembler(const Mesh &msh, hho_degree_info hdi)
{
    return interface_assembler<Mesh>(msh, hdi);
}
