// proton_amd/host/hho.hpp -- C++ host side of the MI355X-native HHO assembly path.
//
// Mirrors the reference's interface for this path (names, argument order, meaning), so a driver
// written like apps/convergence_test or apps/obstacle compiles against it:
//   mesh_init_params / quad_mesh            src/core/core_bits/basic_mesh.hpp:178-197, 210-299
//   offset / faces / barycenter / diameter  src/core/core_bits/basic_geom.hpp:30-61, 183-212, 247-315
//   hho_degree_info                         src/core/core_bits/utils.hpp:62-111
//   make_rhs                                src/core/core_bits/utils.hpp:153-174
//   make_hho_laplacian                      src/methods/hho_bits/hho.hpp:32-96
//   make_hho_naive_stabilization            src/methods/hho_bits/hho.hpp:99-148
//   make_hho_fancy_stabilization            src/methods/hho_bits/hho.hpp:155-237
//   assembler / make_assembler              src/methods/hho_bits/hho.hpp:252-463
//   obstacle_assembler / take_local_data    src/methods/hho_bits/hho.hpp:471-789
//   project_function                        src/core/core_bits/utils.hpp:199-227
//   cg_params / conjugated_gradient         src/core/core_bits/solver_cg.hpp:38-144
// Everything numerical is computed on the GPU through the C ABI of include/proton_amd.h
// (libproton_amd.so); there is no CPU fallback -- a missing library or GPU throws.
//
// The per-cell calls of the reference are served from a batch: the first call for a given
// (mesh, degrees, quadrature, stabilization) computes ALL cells in one launch and caches the
// result; later calls copy one cell out of it.  Large meshes should use the batched entry
// points (proton_amd::local_operators / assemble_all) and keep the data on the device.
//
// Matrices are returned as proton_amd::dense_matrix<T>: column-major like Eigen's default,
// with rows(), cols(), operator()(i,j), data(); if <Eigen/Dense> is available, map() gives an
// Eigen::Map over the same storage.
#pragma once

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <typeindex>
#include <typeinfo>
#include <utility>
#include <vector>

#include "../../include/proton_amd.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define PROTON_AMD_HAS_EIGEN 1
#endif
#endif

namespace proton_amd {

struct error : std::runtime_error {
    int status;
    error(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};

inline const char *status_name(int st)
{
    switch (st) {
    case PA_OK: return "PA_OK";
    case PA_ERR_INVALID_ARG: return "PA_ERR_INVALID_ARG";
    case PA_ERR_INVALID_DEGREE: return "PA_ERR_INVALID_DEGREE";
    case PA_ERR_QUADRATURE: return "PA_ERR_QUADRATURE (Quadrature order too high)";
    case PA_ERR_HIP: return "PA_ERR_HIP";
    case PA_ERR_NO_MESH: return "PA_ERR_NO_MESH";
    case PA_ERR_NOT_SPD: return "PA_ERR_NOT_SPD";
    default: return "unknown status";
    }
}

// One pa_context (one GPU, a library-owned stream) and everything the per-cell API caches for it.  The reference's
// functions take no context argument, so the per-cell API runs on the CURRENT device of the calling thread: the default
// instance (device 0 unless PROTON_AMD_DEVICE is set), or the one a device_scope has made current --
//     proton_amd::device gpu1(1);
//     { proton_amd::device_scope on(gpu1);  auto gr = make_hho_laplacian(msh, cl, hdi); ... }
// Devices are independent: each owns its mesh copy and its batches (no state is shared between them).
class device {
    pa_context *ctx_ = nullptr;
    int index_ = 0;
    std::map<std::type_index, std::shared_ptr<void>> slots_;      // per-device caches, by type (batch_cache<Mesh>)

    static int default_index()
    {
        if (const char *e = std::getenv("PROTON_AMD_DEVICE")) return std::atoi(e);
        return 0;
    }
    static device *&current_slot()
    {
        static thread_local device *cur = nullptr;
        return cur;
    }
    friend class device_scope;

  public:
    device() : device(default_index()) {}
    explicit device(int dev) : index_(dev)
    {
        const int st = pa_context_create(dev, nullptr, 1, &ctx_);
        if (st != PA_OK) throw error(st, std::string("pa_context_create: ") + status_name(st) + " (no GPU? there is no CPU fallback)");
    }
    ~device()
    {
        slots_.clear();
        if (ctx_) pa_context_destroy(ctx_);
    }
    device(const device &) = delete;
    device &operator=(const device &) = delete;
    pa_context *ctx() const { return ctx_; }
    int index() const { return index_; }
    void check(int st, const char *where) const
    {
        if (st != PA_OK) throw error(st, std::string(where) + ": " + status_name(st) + " " + pa_last_error(ctx_));
    }
    // the cache object of type T this device owns (created on first use)
    template <typename T>
    T &slot()
    {
        auto &p = slots_[std::type_index(typeid(T))];
        if (!p) p = std::make_shared<T>();
        return *static_cast<T *>(p.get());
    }
    // the device the per-cell API of the calling thread runs on
    static device &instance()
    {
        if (device *c = current_slot()) return *c;
        static device d;
        return d;
    }
};

// makes `d` the current device of the calling thread for the scope's lifetime
class device_scope {
    device *prev_;

  public:
    explicit device_scope(device &d) : prev_(device::current_slot()) { device::current_slot() = &d; }
    ~device_scope() { device::current_slot() = prev_; }
    device_scope(const device_scope &) = delete;
    device_scope &operator=(const device_scope &) = delete;
};

// RAII device buffer
template <typename T>
class device_buffer {
    T *p_ = nullptr;
    size_t n_ = 0;
    device *dev_ = nullptr;       // the device it was allocated on (freed there, whatever is current then)

  public:
    device_buffer() = default;
    explicit device_buffer(size_t n) { resize(n); }
    ~device_buffer() { release(); }
    device_buffer(const device_buffer &) = delete;
    device_buffer &operator=(const device_buffer &) = delete;
    device_buffer(device_buffer &&o) noexcept : p_(o.p_), n_(o.n_), dev_(o.dev_) { o.p_ = nullptr; o.n_ = 0; }
    void release()
    {
        if (p_) pa_free(dev_->ctx(), p_);
        p_ = nullptr; n_ = 0;
    }
    void resize(size_t n)
    {
        release();
        dev_ = &device::instance();
        void *q = nullptr;
        dev_->check(pa_malloc(dev_->ctx(), n * sizeof(T), &q), "pa_malloc");
        p_ = static_cast<T *>(q); n_ = n;
    }
    T *get() const { return p_; }
    size_t size() const { return n_; }
    void upload(const T *src, size_t n) { dev_->check(pa_memcpy_h2d(dev_->ctx(), p_, src, n * sizeof(T)), "pa_memcpy_h2d"); }
    void download(T *dst, size_t n, size_t offset = 0) const
    {
        dev_->check(pa_memcpy_d2h(dev_->ctx(), dst, p_ + offset, n * sizeof(T)), "pa_memcpy_d2h");
    }
};

// column-major dense matrix (vectors are n x 1)
template <typename T>
class dense_matrix {
    size_t r_ = 0, c_ = 0;
    std::vector<T> v_;

  public:
    dense_matrix() = default;
    dense_matrix(size_t r, size_t c, T init = T(0)) : r_(r), c_(c), v_(r * c, init) {}
    static dense_matrix Zero(size_t r, size_t c = 1) { return dense_matrix(r, c); }
    size_t rows() const { return r_; }
    size_t cols() const { return c_; }
    size_t size() const { return v_.size(); }
    T *data() { return v_.data(); }
    const T *data() const { return v_.data(); }
    T &operator()(size_t i, size_t j) { return v_[i + j * r_]; }
    const T &operator()(size_t i, size_t j) const { return v_[i + j * r_]; }
    T &operator()(size_t i) { return v_[i]; }
    const T &operator()(size_t i) const { return v_[i]; }
    dense_matrix operator+(const dense_matrix &o) const
    {
        assert(r_ == o.r_ && c_ == o.c_);
        dense_matrix m(r_, c_);
        for (size_t i = 0; i < v_.size(); ++i) m.v_[i] = v_[i] + o.v_[i];
        return m;
    }
    dense_matrix operator-(const dense_matrix &o) const
    {
        assert(r_ == o.r_ && c_ == o.c_);
        dense_matrix m(r_, c_);
        for (size_t i = 0; i < v_.size(); ++i) m.v_[i] = v_[i] - o.v_[i];
        return m;
    }
    dense_matrix operator*(const dense_matrix &o) const
    {
        assert(c_ == o.r_);
        dense_matrix m(r_, o.c_);
        for (size_t j = 0; j < o.c_; ++j)
            for (size_t k = 0; k < c_; ++k) {
                const T b = o(k, j);
                for (size_t i = 0; i < r_; ++i) m(i, j) += (*this)(i, k) * b;
            }
        return m;
    }
    dense_matrix operator*(T s) const
    {
        dense_matrix m(*this);
        for (auto &x : m.v_) x *= s;
        return m;
    }
    dense_matrix transpose() const
    {
        dense_matrix m(c_, r_);
        for (size_t j = 0; j < c_; ++j)
            for (size_t i = 0; i < r_; ++i) m(j, i) = (*this)(i, j);
        return m;
    }
    // this.block(i0, j0, m.rows(), m.cols()) += m   (Eigen: lc.block(...) += ...)
    void add_to_block(size_t i0, size_t j0, const dense_matrix &m)
    {
        assert(i0 + m.r_ <= r_ && j0 + m.c_ <= c_);
        for (size_t j = 0; j < m.c_; ++j)
            for (size_t i = 0; i < m.r_; ++i) (*this)(i0 + i, j0 + j) += m(i, j);
    }
    dense_matrix block(size_t i0, size_t j0, size_t nr, size_t nc) const
    {
        dense_matrix m(nr, nc);
        for (size_t j = 0; j < nc; ++j)
            for (size_t i = 0; i < nr; ++i) m(i, j) = (*this)(i0 + i, j0 + j);
        return m;
    }
    T dot(const dense_matrix &o) const
    {
        assert(v_.size() == o.v_.size());
        T s = 0;
        for (size_t i = 0; i < v_.size(); ++i) s += v_[i] * o.v_[i];
        return s;
    }
#ifdef PROTON_AMD_HAS_EIGEN
    Eigen::Map<Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic>> map() { return {v_.data(), (Eigen::Index)r_, (Eigen::Index)c_}; }
#endif
};

// CSR matrix produced by assembler::finalize (setFromTriplets semantics: duplicates are summed)
template <typename T>
struct sparse_matrix {
    size_t nrows = 0, ncols = 0;
    std::vector<int64_t> rowptr;
    std::vector<int32_t> colind;
    std::vector<T> values;
    size_t rows() const { return nrows; }
    size_t cols() const { return ncols; }
    size_t nonZeros() const { return values.size(); }
    std::vector<T> multiply(const std::vector<T> &x) const
    {
        std::vector<T> y(nrows, T(0));
        for (size_t i = 0; i < nrows; ++i) {
            T s = 0;
            for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) s += values[k] * x[colind[k]];
            y[i] = s;
        }
        return y;
    }
    void set_from_triplets(size_t n, std::vector<std::tuple<int32_t, int32_t, T>> &trip)
    {
        nrows = ncols = n;
        std::sort(trip.begin(), trip.end(), [](const auto &a, const auto &b) {
            return std::get<0>(a) != std::get<0>(b) ? std::get<0>(a) < std::get<0>(b) : std::get<1>(a) < std::get<1>(b);
        });
        rowptr.assign(n + 1, 0);
        colind.clear(); values.clear();
        for (size_t k = 0; k < trip.size();) {
            const int32_t r = std::get<0>(trip[k]), c = std::get<1>(trip[k]);
            T s = 0;
            while (k < trip.size() && std::get<0>(trip[k]) == r && std::get<1>(trip[k]) == c) s += std::get<2>(trip[k++]);
            colind.push_back(c); values.push_back(s); rowptr[r + 1]++;
        }
        for (size_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    }
};

}  // namespace proton_amd

// ---------------------------------------------------------------------------------------------
// point, mesh (basic_mesh.hpp:49-299, point.hpp)
// ---------------------------------------------------------------------------------------------
template <typename T, size_t DIM>
class point {
    std::array<T, DIM> m_coords{};

  public:
    typedef T value_type;
    point() = default;
    point(T x, T y) { m_coords[0] = x; m_coords[1] = y; }
    T x() const { return m_coords[0]; }
    T y() const { return m_coords[1]; }
    T &x() { return m_coords[0]; }
    T &y() { return m_coords[1]; }
    T operator[](size_t i) const { return m_coords[i]; }
    friend point operator+(const point &a, const point &b) { return point(a.x() + b.x(), a.y() + b.y()); }
    friend point operator-(const point &a, const point &b) { return point(a.x() - b.x(), a.y() - b.y()); }
    friend point operator*(const point &a, T s) { return point(a.x() * s, a.y() * s); }
    friend point operator*(T s, const point &a) { return a * s; }
    friend point operator/(const point &a, T s) { return point(a.x() / s, a.y() / s); }
};

enum class boundary { NONE, DIRICHLET, NEUMANN, ROBIN };

template <typename T>
struct mesh_init_params {
    T min_x, max_x, min_y, max_y;
    size_t Nx, Ny;
    mesh_init_params() : min_x(0.0), max_x(1.0), min_y(0.0), max_y(1.0), Nx(4), Ny(4) {}
    T hx() const { return (max_x - min_x) / Nx; }
    T hy() const { return (max_y - min_y) / Ny; }
};

template <typename T>
struct quad_mesh {
    typedef point<T, 2> point_type;
    typedef T coordinate_type;
    // integrate(msh, cl, degree) of this mesh type: tensor Gauss on the bilinear map
    // (quadratures.hpp:311-375); cuthho_poly_mesh uses the triangle fan (:377-402)
    static constexpr int pa_quadrature = PA_QUAD_TENSOR;
    struct cell_type {
        std::array<size_t, 4> ptids;
        bool operator<(const cell_type &o) const { return ptids < o.ptids; }
        bool operator==(const cell_type &o) const { return ptids == o.ptids; }
    };
    struct face_type {
        std::array<size_t, 2> ptids;
        bool is_boundary = false;
        boundary bndtype = boundary::NONE;
        bool operator<(const face_type &o) const { return ptids < o.ptids; }
        bool operator==(const face_type &o) const { return ptids == o.ptids; }
    };
    struct node_type { size_t ptid; };

    std::vector<point_type> points;
    std::vector<node_type> nodes;
    std::vector<face_type> faces;
    std::vector<cell_type> cells;
    mesh_init_params<T> params;

    quad_mesh() : quad_mesh(mesh_init_params<T>()) {}
    // the structured generator, basic_mesh.hpp:230-298
    explicit quad_mesh(const mesh_init_params<T> &parms) : params(parms)
    {
        const auto hx = parms.hx(), hy = parms.hy();
        points.reserve((parms.Nx + 1) * (parms.Ny + 1));
        size_t point_num = 0;
        for (size_t j = 0; j < parms.Ny + 1; j++)
            for (size_t i = 0; i < parms.Nx + 1; i++) {
                points.push_back(point_type(parms.min_x + i * hx, parms.min_y + j * hy));
                nodes.push_back(node_type{point_num++});
            }
        for (size_t j = 0; j < parms.Ny; j++)
            for (size_t i = 0; i < parms.Nx; i++) {
                const size_t p0 = j * (parms.Nx + 1) + i, p1 = p0 + 1, p2 = p0 + parms.Nx + 2, p3 = p0 + parms.Nx + 1;
                cells.push_back(cell_type{{{p0, p1, p2, p3}}});
                face_type f0; f0.ptids = {p0, p1}; f0.is_boundary = (j == 0);
                face_type f1; f1.ptids = {p1, p2}; f1.is_boundary = (i == parms.Nx - 1);
                face_type f2; f2.ptids = {p3, p2}; f2.is_boundary = (j == parms.Ny - 1);
                face_type f3; f3.ptids = {p0, p3}; f3.is_boundary = (i == 0);
                faces.push_back(f0); faces.push_back(f1); faces.push_back(f2); faces.push_back(f3);
            }
        std::sort(cells.begin(), cells.end());
        std::sort(faces.begin(), faces.end());
        faces.erase(std::unique(faces.begin(), faces.end()), faces.end());
        for (auto &fc : faces)
            if (fc.is_boundary) fc.bndtype = boundary::DIRICHLET;
    }
};

// basic_geom.hpp:30-61
template <typename Mesh>
size_t offset(const Mesh &msh, const typename Mesh::cell_type &cl)
{
    auto itor = std::lower_bound(msh.cells.begin(), msh.cells.end(), cl);
    if (itor == msh.cells.end()) throw std::logic_error("Cell not found: this is likely a bug.");
    return std::distance(msh.cells.begin(), itor);
}
template <typename Mesh>
size_t offset(const Mesh &msh, const typename Mesh::face_type &fc)
{
    auto itor = std::lower_bound(msh.faces.begin(), msh.faces.end(), fc);
    if (itor == msh.faces.end()) throw std::logic_error("Face not found: this is likely a bug.");
    return std::distance(msh.faces.begin(), itor);
}
// basic_geom.hpp:183-212
template <typename Mesh>
std::array<typename Mesh::face_type, 4> faces(const Mesh &msh, const typename Mesh::cell_type &cl)
{
    std::array<typename Mesh::face_type, 4> ret;
    for (size_t i = 0; i < 4; i++) {
        typename Mesh::face_type f;
        f.ptids[0] = cl.ptids[i];
        f.ptids[1] = cl.ptids[(i + 1) % 4];
        if (f.ptids[0] > f.ptids[1]) std::swap(f.ptids[0], f.ptids[1]);
        auto itor = std::lower_bound(msh.faces.begin(), msh.faces.end(), f);
        if (itor == msh.faces.end()) throw std::logic_error("Face not found, this is likely a bug.");
        ret[i] = *itor;
    }
    return ret;
}
template <typename Mesh>
std::array<typename Mesh::point_type, 4> points(const Mesh &msh, const typename Mesh::cell_type &cl)
{
    std::array<typename Mesh::point_type, 4> ret;
    for (size_t i = 0; i < 4; i++) ret[i] = msh.points.at(cl.ptids[i]);
    return ret;
}
// basic_geom.hpp:247-305
template <typename Mesh>
typename Mesh::point_type barycenter(const Mesh &msh, const typename Mesh::cell_type &cl)
{
    typedef typename Mesh::coordinate_type T;
    auto pts = points(msh, cl);
    typename Mesh::point_type ret;
    T den = 0.0;
    for (size_t i = 2; i < 4; i++) {
        auto pprev = pts[i - 1] - pts[0], pcur = pts[i] - pts[0];
        auto d = (pprev.x() * pcur.y() - pprev.y() * pcur.x()) / 2.0;
        ret = ret + (pprev + pcur) * d;
        den += d;
    }
    return pts[0] + ret / (den * 3);
}
template <typename Mesh>
typename Mesh::coordinate_type diameter(const Mesh &msh, const typename Mesh::cell_type &cl)
{
    typename Mesh::coordinate_type diam = 0.0;
    for (size_t i = 0; i < 4; i++)
        for (size_t j = i + 1; j < 4; j++) {
            auto d = msh.points.at(cl.ptids[j]) - msh.points.at(cl.ptids[i]);
            diam = std::max(diam, std::sqrt(d.x() * d.x() + d.y() * d.y()));
        }
    return diam;
}

// utils.hpp:62-111
class hho_degree_info {
    size_t cell_deg, face_deg, reconstruction_deg;

  public:
    hho_degree_info() : cell_deg(1), face_deg(1), reconstruction_deg(2) {}
    explicit hho_degree_info(size_t degree) : cell_deg(degree), face_deg(degree), reconstruction_deg(degree + 1) {}
    hho_degree_info(size_t cd, size_t fd)
    {
        int fell_back = 0;
        const pa_degree_info d = pa_degree_info_make((int)cd, (int)fd, &fell_back);
        if (fell_back) std::cout << "Invalid cell degree. Reverting to equal-order" << std::endl;      // utils.hpp:88
        cell_deg = d.cell_deg; face_deg = d.face_deg; reconstruction_deg = d.rec_deg;
    }
    size_t cell_degree() const { return cell_deg; }
    size_t face_degree() const { return face_deg; }
    size_t reconstruction_degree() const { return reconstruction_deg; }
    pa_degree_info c_abi() const { return pa_degree_info{(int32_t)cell_deg, (int32_t)face_deg, (int32_t)reconstruction_deg}; }
};

namespace proton_amd {

// ---------------------------------------------------------------------------------------------
// The batch behind the per-cell API
// ---------------------------------------------------------------------------------------------
template <typename Mesh>
class mesh_on_device {
  public:
    const Mesh *msh = nullptr;
    size_t npoints = 0, ncells = 0;
    uint64_t probe = 0;      // change detector: hash of EVERY point coordinate (FNV-1a over the bit patterns)

    static uint64_t make_probe(const Mesh &m)
    {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < m.points.size(); ++i) {
            const double xy[2] = {m.points[i].x(), m.points[i].y()};
            uint64_t b[2];
            std::memcpy(b, xy, sizeof(b));
            h = (h ^ b[0]) * 1099511628211ull;
            h = (h ^ b[1]) * 1099511628211ull;
        }
        return h;
    }
    // same object and sizes: the cheap test of every per-cell call
    bool same_object(const Mesh &m) const { return msh == &m && npoints == m.points.size() && ncells == m.cells.size(); }
    // ... and the same coordinates, every one of them (an O(points) pass)
    bool matches(const Mesh &m) const { return same_object(m) && probe == make_probe(m); }
    // msh.points, cell.ptids and msh.faces as the reference holds them
    void upload(const Mesh &m)
    {
        auto &dev = device::instance();
        std::vector<double> pts(2 * m.points.size());
        for (size_t i = 0; i < m.points.size(); ++i) { pts[2 * i] = m.points[i].x(); pts[2 * i + 1] = m.points[i].y(); }
        std::vector<uint32_t> ids(4 * m.cells.size()), cf(4 * m.cells.size());
        for (size_t c = 0; c < m.cells.size(); ++c) {
            auto fcs = faces(m, m.cells[c]);
            for (int v = 0; v < 4; ++v) {
                ids[4 * c + v] = (uint32_t)m.cells[c].ptids[v];
                cf[4 * c + v] = (uint32_t)offset(m, fcs[v]);
            }
        }
        std::vector<uint32_t> fp(2 * m.faces.size());
        std::vector<uint8_t> fd(m.faces.size());
        for (size_t f = 0; f < m.faces.size(); ++f) {
            fp[2 * f] = (uint32_t)m.faces[f].ptids[0]; fp[2 * f + 1] = (uint32_t)m.faces[f].ptids[1];
            fd[f] = (m.faces[f].is_boundary && m.faces[f].bndtype == boundary::DIRICHLET) ? 1 : 0;
        }
        dev.check(pa_mesh_upload(dev.ctx(), pts.data(), m.points.size(), ids.data(), m.cells.size()), "pa_mesh_upload");
        dev.check(pa_mesh_set_faces(dev.ctx(), cf.data(), fp.data(), fd.data(), m.faces.size()), "pa_mesh_set_faces");
        msh = &m; npoints = m.points.size(); ncells = m.cells.size(); probe = make_probe(m);
    }
};

struct local_batch {
    pa_sizes sz{};
    std::vector<double> oper, data, stab;     // host copies, cell-major, column-major per cell
    std::vector<int32_t> info;
};

template <typename Mesh>
class batch_cache {
    mesh_on_device<Mesh> dev_mesh_;
    std::map<std::tuple<int, int, int, int>, std::shared_ptr<local_batch>> batches_;
    std::map<std::tuple<int, int, int>, std::vector<double>> qpoints_;      // (degree, quad) -> n x nq x 3
    size_t last_pos_ = 0;
    std::map<int, std::vector<double>> face_qpoints_;                       // Gauss points per face -> nfaces x nfq x 3
    std::map<int, std::shared_ptr<device_buffer<double>>> face_samples_;    // ... -> device array nfaces x nfq of samples

  public:
    // the cache of the calling thread's current device (device::instance(), device_scope)
    static batch_cache &instance() { return device::instance().template slot<batch_cache>(); }
    // The cached mesh copy and batches serve `m` only while m is the same object with the same coordinates.  The
    // coordinates are re-hashed in full whenever a per-cell call does not continue a forward sweep over the cells
    // (pos = the cell's offset; the reference's drivers edit a mesh between their loops over the cells, not inside
    // one), and by every call that is not tied to a cell.  Editing a mesh in the middle of a sweep needs
    // proton_amd::invalidate(msh).
    void ensure_mesh(const Mesh &m, size_t pos = 0)
    {
        // (>=: the reference's loop body calls make_hho_laplacian, the stabilization and make_rhs on the SAME cell -- with >
        // the second and third call of every cell re-hashed every point: O(cells x points) per sweep; pos 0 -- the first cell, and
        // every call that is not tied to a cell -- always re-hashes)
        const bool sweep_continues = pos > 0 && pos >= last_pos_;
        last_pos_ = pos;
        if (sweep_continues ? !dev_mesh_.same_object(m) : !dev_mesh_.matches(m)) {
            batches_.clear(); qpoints_.clear(); face_qpoints_.clear(); face_samples_.clear();
            dev_mesh_.upload(m);
        }
    }
    // points of integrate(msh, fc, 2 (nfq - 1)) for every face (x, y, w), and a device array for per-face samples
    const std::vector<double> &face_qpoints(const Mesh &m, int nfq, size_t pos = 0)
    {
        ensure_mesh(m, pos);
        auto it = face_qpoints_.find(nfq);
        if (it != face_qpoints_.end()) return it->second;
        auto &dev = device::instance();
        const size_t nf = m.faces.size();
        device_buffer<double> d(nf * nfq * 3);
        dev.check(pa_face_quadrature_points(dev.ctx(), nfq - 1, d.get()), "pa_face_quadrature_points");
        std::vector<double> h(nf * nfq * 3);
        d.download(h.data(), h.size());
        return face_qpoints_[nfq] = std::move(h);
    }
    device_buffer<double> &face_samples(const Mesh &m, int nfq)
    {
        auto &p = face_samples_[nfq];
        if (!p || p->size() != m.faces.size() * nfq) {
            p = std::make_shared<device_buffer<double>>(m.faces.size() * nfq);
            auto &dev = device::instance();
            dev.check(pa_memset(dev.ctx(), p->get(), 0, p->size() * sizeof(double)), "pa_memset");
        }
        return *p;
    }
    void invalidate() { dev_mesh_ = mesh_on_device<Mesh>(); batches_.clear(); qpoints_.clear(); face_qpoints_.clear(); face_samples_.clear(); }
    // the context already holds this mesh (pa_cut_preprocess built it): no upload
    void adopt(const Mesh &m)
    {
        batches_.clear(); qpoints_.clear(); face_qpoints_.clear(); face_samples_.clear();
        dev_mesh_.msh = &m; dev_mesh_.npoints = m.points.size(); dev_mesh_.ncells = m.cells.size();
        dev_mesh_.probe = mesh_on_device<Mesh>::make_probe(m);
    }

    std::shared_ptr<local_batch> get(const Mesh &m, const hho_degree_info &hdi, int quad, int stab, size_t pos = 0)
    {
        ensure_mesh(m, pos);
        const auto key = std::make_tuple((int)hdi.cell_degree(), (int)hdi.face_degree(), quad, stab);
        auto it = batches_.find(key);
        if (it != batches_.end()) return it->second;
        auto &dev = device::instance();
        auto b = std::make_shared<local_batch>();
        dev.check(pa_sizes_for(hdi.c_abi(), quad, &b->sz), "pa_sizes_for");
        const size_t n = m.cells.size(), mm = (size_t)b->sz.msize * b->sz.msize, om = (size_t)b->sz.oper_rows * b->sz.msize;
        device_buffer<double> d_oper(n * om), d_data(n * mm), d_stab(n * mm);
        device_buffer<int32_t> d_info(n);
        dev.check(pa_local_ops_batch(dev.ctx(), hdi.c_abi(), quad, stab, 0, n, d_oper.get(), d_data.get(), d_stab.get(),
                                     nullptr, d_info.get()), "pa_local_ops_batch");
        b->oper.resize(n * om); b->data.resize(n * mm); b->stab.resize(n * mm); b->info.resize(n);
        d_oper.download(b->oper.data(), n * om);
        d_data.download(b->data.data(), n * mm);
        d_stab.download(b->stab.data(), n * mm);
        d_info.download(b->info.data(), n);
        batches_[key] = b;
        return b;
    }

    // quadrature points of integrate(msh, cl, degree) for every cell (x, y, w)
    const std::vector<double> &cell_qpoints(const Mesh &m, int degree, int quad, int &nq, size_t pos = 0)
    {
        ensure_mesh(m, pos);
        auto &dev = device::instance();
        int32_t nqp = 0;
        dev.check(pa_cell_quadrature_points(dev.ctx(), degree, quad, 0, 0, nullptr, &nqp), "pa_cell_quadrature_points");
        nq = nqp;
        const auto key = std::make_tuple(degree, quad, 0);
        auto it = qpoints_.find(key);
        if (it != qpoints_.end()) return it->second;
        const size_t n = m.cells.size();
        device_buffer<double> d(n * nqp * 3);
        dev.check(pa_cell_quadrature_points(dev.ctx(), degree, quad, 0, n, d.get(), &nqp), "pa_cell_quadrature_points");
        std::vector<double> h(n * nqp * 3);
        d.download(h.data(), h.size());
        return qpoints_[key] = std::move(h);
    }
};

template <typename Mesh>
inline void invalidate(const Mesh &) { batch_cache<Mesh>::instance().invalidate(); }

template <typename T>
inline dense_matrix<T> copy_cell(const std::vector<double> &src, size_t cell, size_t rows, size_t cols)
{
    dense_matrix<T> m(rows, cols);
    std::memcpy(m.data(), src.data() + cell * rows * cols, rows * cols * sizeof(double));
    return m;
}

}  // namespace proton_amd

// ---------------------------------------------------------------------------------------------
// The reference's per-cell interface
// ---------------------------------------------------------------------------------------------
// hho.hpp:32-35  -> pair(oper (rbs-1) x msize, data msize x msize)
template <typename Mesh>
std::pair<proton_amd::dense_matrix<typename Mesh::coordinate_type>, proton_amd::dense_matrix<typename Mesh::coordinate_type>>
make_hho_laplacian(const Mesh &msh, const typename Mesh::cell_type &cl, const hho_degree_info &di)
{
    using T = typename Mesh::coordinate_type;
    const size_t c = offset(msh, cl);
    auto b = proton_amd::batch_cache<Mesh>::instance().get(msh, di, Mesh::pa_quadrature, PA_STAB_FANCY, c);
    return std::make_pair(proton_amd::copy_cell<T>(b->oper, c, b->sz.oper_rows, b->sz.msize),
                          proton_amd::copy_cell<T>(b->data, c, b->sz.msize, b->sz.msize));
}

// hho.hpp:99-101
template <typename Mesh>
proton_amd::dense_matrix<typename Mesh::coordinate_type>
make_hho_naive_stabilization(const Mesh &msh, const typename Mesh::cell_type &cl, const hho_degree_info &di)
{
    using T = typename Mesh::coordinate_type;
    const size_t c = offset(msh, cl);
    auto b = proton_amd::batch_cache<Mesh>::instance().get(msh, di, Mesh::pa_quadrature, PA_STAB_NAIVE, c);
    return proton_amd::copy_cell<T>(b->stab, c, b->sz.msize, b->sz.msize);
}

// hho.hpp:155-159.  The reference builds the stabilization from the `reconstruction` it is handed (hho.hpp:184-190,
// 222-226).  The batch computes it from the reconstruction operator of the same cell, so the argument must BE that
// operator -- make_hho_laplacian(msh, cl, di).first, what every driver of the reference passes -- and anything else is
// refused rather than silently answered with the stabilization of a different matrix.
template <typename Mesh>
proton_amd::dense_matrix<typename Mesh::coordinate_type>
make_hho_fancy_stabilization(const Mesh &msh, const typename Mesh::cell_type &cl,
                             const proton_amd::dense_matrix<typename Mesh::coordinate_type> &reconstruction,
                             const hho_degree_info &di)
{
    using T = typename Mesh::coordinate_type;
    const size_t c = offset(msh, cl);
    auto b = proton_amd::batch_cache<Mesh>::instance().get(msh, di, Mesh::pa_quadrature, PA_STAB_FANCY, c);
    const size_t om = (size_t)b->sz.oper_rows * b->sz.msize;
    if (reconstruction.rows() != (size_t)b->sz.oper_rows || reconstruction.cols() != (size_t)b->sz.msize ||
        std::memcmp(reconstruction.data(), b->oper.data() + c * om, om * sizeof(double)) != 0)
        throw std::invalid_argument("make_hho_fancy_stabilization: `reconstruction` is not make_hho_laplacian(msh, cl, di).first of this "
                                    "cell; the device batch computes the stabilization of the cell's own reconstruction operator only");
    return proton_amd::copy_cell<T>(b->stab, c, b->sz.msize, b->sz.msize);
}

// utils.hpp:153-156.  The functor runs on the host at the quadrature points of
// integrate(msh, cl, 2*(degree+di)); the weighted sums run on the device (PA_FN_SAMPLED).
template <typename Mesh, typename Function>
proton_amd::dense_matrix<typename Mesh::coordinate_type>
make_rhs(const Mesh &msh, const typename Mesh::cell_type &cl, size_t degree, const Function &f, size_t di = 0)
{
    using T = typename Mesh::coordinate_type;
    auto &cache = proton_amd::batch_cache<Mesh>::instance();
    auto &dev = proton_amd::device::instance();
    int nq = 0;
    const size_t c = offset(msh, cl);
    const auto &xyw = cache.cell_qpoints(msh, (int)(2 * (degree + di)), Mesh::pa_quadrature, nq, c);
    std::vector<double> fv(nq);
    for (int q = 0; q < nq; ++q) fv[q] = f(typename Mesh::point_type(xyw[(c * nq + q) * 3], xyw[(c * nq + q) * 3 + 1]));
    const size_t cbs = (degree + 2) * (degree + 1) / 2;
    proton_amd::device_buffer<double> d_f(nq), d_r(cbs);
    d_f.upload(fv.data(), nq);
    dev.check(pa_cell_rhs_batch(dev.ctx(), (int)degree, (int)di, Mesh::pa_quadrature, PA_FN_SAMPLED, d_f.get(), c, 1, d_r.get()),
              "pa_cell_rhs_batch");
    proton_amd::dense_matrix<T> ret(cbs, 1);
    d_r.download(ret.data(), cbs);
    return ret;
}

// utils.hpp:199-227.  The functor is sampled on the host at the points of integrate(msh, cl, 2*(celdeg+di)) and, for
// the cell's four faces, integrate(msh, fc, 2*(facdeg+di)); the mass matrices, right-hand sides and LLT solves of THIS
// cell run on the device.  Nothing of the result is cached: the functor is evaluated afresh at every call (a cache
// keyed on the functor's address would hand a re-created temporary the previous functor's projection).  Drivers
// that project one function on every cell use project_function_all below (one device batch).
template <typename Mesh, typename Function>
proton_amd::dense_matrix<typename Mesh::coordinate_type>
project_function(const Mesh &msh, const typename Mesh::cell_type &cl, hho_degree_info hdi, const Function &f, size_t di = 0)
{
    using T = typename Mesh::coordinate_type;
    auto &dev = proton_amd::device::instance();
    auto &bc = proton_amd::batch_cache<Mesh>::instance();
    const size_t cd = hdi.cell_degree(), fd = hdi.face_degree();
    const size_t cbs = (cd + 2) * (cd + 1) / 2, fbs = fd + 1, ms = cbs + 4 * fbs, c = offset(msh, cl);
    int nq = 0;
    const auto &xyw = bc.cell_qpoints(msh, (int)(2 * (cd + di)), Mesh::pa_quadrature, nq, c);
    std::vector<double> cv(nq);
    for (int q = 0; q < nq; ++q) cv[q] = f(typename Mesh::point_type(xyw[(c * nq + q) * 3], xyw[(c * nq + q) * 3 + 1]));
    const int nfq = (int)(fd + di + 1);
    const auto &fx = bc.face_qpoints(msh, nfq, c);
    auto &d_fv = bc.face_samples(msh, nfq);
    const auto fcs = faces(msh, cl);
    for (size_t lf = 0; lf < 4; ++lf) {
        const size_t fo = offset(msh, fcs[lf]);
        std::vector<double> fv(nfq);
        for (int q = 0; q < nfq; ++q) fv[q] = f(typename Mesh::point_type(fx[(fo * nfq + q) * 3], fx[(fo * nfq + q) * 3 + 1]));
        dev.check(pa_memcpy_h2d(dev.ctx(), d_fv.get() + fo * nfq, fv.data(), nfq * sizeof(double)), "pa_memcpy_h2d");
    }
    proton_amd::device_buffer<double> d_cv(nq), d_out(ms);
    d_cv.upload(cv.data(), nq);
    dev.check(pa_project_function_batch(dev.ctx(), hdi.c_abi(), Mesh::pa_quadrature, (int)di, PA_FN_SAMPLED, d_cv.get(), d_fv.get(), c, 1,
                                        d_out.get(), nullptr), "pa_project_function_batch");
    proton_amd::dense_matrix<T> ret(ms, 1);
    d_out.download(ret.data(), ms);
    return ret;
}

// The same for every cell in one device batch: ncells x msize coefficients, cell-major (cell c at [c * msize, ...)).
template <typename Mesh, typename Function>
std::vector<typename Mesh::coordinate_type>
project_function_all(const Mesh &msh, hho_degree_info hdi, const Function &f, size_t di = 0)
{
    auto &dev = proton_amd::device::instance();
    auto &bc = proton_amd::batch_cache<Mesh>::instance();
    const size_t cd = hdi.cell_degree(), fd = hdi.face_degree();
    const size_t cbs = (cd + 2) * (cd + 1) / 2, fbs = fd + 1, ms = cbs + 4 * fbs, n = msh.cells.size(), nf = msh.faces.size();
    int nq = 0;
    const auto &xyw = bc.cell_qpoints(msh, (int)(2 * (cd + di)), Mesh::pa_quadrature, nq);
    std::vector<double> cv(n * nq);
    for (size_t k = 0; k < n * (size_t)nq; ++k) cv[k] = f(typename Mesh::point_type(xyw[3 * k], xyw[3 * k + 1]));
    const int nfq = (int)(fd + di + 1);
    const auto &fx = bc.face_qpoints(msh, nfq);
    std::vector<double> fv(nf * nfq);
    for (size_t k = 0; k < nf * (size_t)nfq; ++k) fv[k] = f(typename Mesh::point_type(fx[3 * k], fx[3 * k + 1]));
    proton_amd::device_buffer<double> d_cv(cv.size()), d_fv(fv.size()), d_out(n * ms);
    d_cv.upload(cv.data(), cv.size());
    d_fv.upload(fv.data(), fv.size());
    dev.check(pa_project_function_batch(dev.ctx(), hdi.c_abi(), Mesh::pa_quadrature, (int)di, PA_FN_SAMPLED, d_cv.get(), d_fv.get(), 0, n,
                                        d_out.get(), nullptr), "pa_project_function_batch");
    std::vector<typename Mesh::coordinate_type> all(n * ms);
    d_out.download(all.data(), all.size());
    return all;
}

namespace proton_amd {

// What both assemblers share: the numbering of the non-Dirichlet faces (the compress table of
// hho.hpp:305-323 / :551-563, -1 for Dirichlet faces) and the Dirichlet data of a boundary
// functor, computed once per functor on the device (mass.llt().solve(rhs), hho.hpp:383-385).
template <typename Mesh>
class face_numbering {
    using T = typename Mesh::coordinate_type;
    std::vector<double> face_xyw_, g_, samples_;
    std::vector<size_t> probe_idx_;      // sample positions on Dirichlet faces at which a cached result is re-checked inside a sweep
    std::vector<size_t> boundary_idx_;   // every sample position on a Dirichlet face: re-checked when a sweep starts
    size_t last_pos_ = 0;

  public:
    std::vector<int64_t> compress;
    size_t num_other_faces = 0, cbs = 0, fbs = 0, face_degree = 0;

    face_numbering(const Mesh &msh, const hho_degree_info &di)
        : cbs((di.cell_degree() + 2) * (di.cell_degree() + 1) / 2), fbs(di.face_degree() + 1), face_degree(di.face_degree())
    {
        compress.assign(msh.faces.size(), -1);
        for (size_t f = 0; f < msh.faces.size(); ++f)
            if (!dirichlet(msh.faces[f])) compress[f] = (int64_t)num_other_faces++;
    }
    static bool dirichlet(const typename Mesh::face_type &fc) { return fc.is_boundary && fc.bndtype == boundary::DIRICHLET; }

    // nfaces x fbs coefficients, zeros on faces that are not Dirichlet
    // `pos`: the offset of the cell an assemble() call is working on (0: a call that is not tied to a cell).
    template <typename Function>
    const std::vector<double> &dirichlet_data(const Mesh &msh, const Function &bf, size_t pos = 0)
    {
        // The data of one boundary function serve every cell of an assembly loop.  They are NOT tied to the functor's address (a
        // re-created temporary may live where the previous functor did).  The reference evaluates the boundary function for every
        // cell at every assemble (hho.hpp:381-386); here a cached result is reused only while the functor reproduces the cached
        // samples: at EVERY quadrature point of every Dirichlet face whenever a call starts a new sweep over the cells (pos not
        // beyond the previous call's: O(boundary faces) functor calls on the host, no device work when they agree) -- two functors
        // that differ anywhere on the boundary get their own data --, and at a handful of probe points spread over the Dirichlet
        // faces for the calls that continue a sweep (the per-cell cost of the check must not grow with the boundary).
        const bool sweep_continues = pos > 0 && pos >= last_pos_;
        last_pos_ = pos;
        if (!g_.empty()) {
            bool same = true;
            if (sweep_continues) {
                for (size_t k : probe_idx_)
                    if (bf(typename Mesh::point_type(face_xyw_[3 * k], face_xyw_[3 * k + 1])) != samples_[k]) { same = false; break; }
            } else {
                for (size_t k : boundary_idx_)
                    if (bf(typename Mesh::point_type(face_xyw_[3 * k], face_xyw_[3 * k + 1])) != samples_[k]) { same = false; break; }
            }
            if (same) return g_;
        }
        auto &dev = device::instance();
        batch_cache<Mesh>::instance().ensure_mesh(msh);
        const size_t nf = msh.faces.size(), nq = fbs;       // integrate(msh, fc, 2*facdeg): facdeg+1 Gauss points
        if (face_xyw_.empty()) {
            device_buffer<double> d(nf * nq * 3);
            dev.check(pa_face_quadrature_points(dev.ctx(), (int)face_degree, d.get()), "pa_face_quadrature_points");
            face_xyw_.resize(nf * nq * 3);
            d.download(face_xyw_.data(), face_xyw_.size());
        }
        std::vector<double> &samples = samples_;
        samples.resize(nf * nq);
        for (size_t k = 0; k < nf * nq; ++k) samples[k] = bf(typename Mesh::point_type(face_xyw_[3 * k], face_xyw_[3 * k + 1]));
        if (probe_idx_.empty()) {
            std::vector<size_t> &on_boundary = boundary_idx_;
            on_boundary.clear();
            for (size_t f = 0; f < nf; ++f)
                if (compress[f] < 0) for (size_t q = 0; q < nq; ++q) on_boundary.push_back(f * nq + q);
            const size_t want = std::min<size_t>(8, on_boundary.size());
            for (size_t i = 0; i < want; ++i) probe_idx_.push_back(on_boundary[i * on_boundary.size() / want]);
        }
        device_buffer<double> d_f(nf * nq), d_g(nf * fbs);
        d_f.upload(samples.data(), samples.size());
        dev.check(pa_dirichlet_data_batch(dev.ctx(), (int)face_degree, PA_FN_SAMPLED, d_f.get(), d_g.get()), "pa_dirichlet_data_batch");
        g_.resize(nf * fbs);
        d_g.download(g_.data(), g_.size());
        return g_;
    }
};

// global offsets of the four faces of a cell, in the cell's local face order
template <typename Mesh>
inline std::array<size_t, 4> face_offsets(const Mesh &msh, const typename Mesh::cell_type &cl)
{
    std::array<size_t, 4> ids;
    auto fcs = faces(msh, cl);
    for (size_t lf = 0; lf < 4; ++lf) ids[lf] = offset(msh, fcs[lf]);
    return ids;
}

}  // namespace proton_amd

// hho.hpp:252-463
template <typename Mesh>
class assembler {
    using T = typename Mesh::coordinate_type;
    hho_degree_info di;
    proton_amd::face_numbering<Mesh> numbering;
    size_t ncells;
    std::vector<std::tuple<int32_t, int32_t, T>> triplets;

    // global index of every local dof, -1 where the reference's assembly_index says "do not
    // assemble" (hho.hpp:362-379), and the Dirichlet coefficient of those dofs
    template <typename Function>
    void local_map(const Mesh &msh, const typename Mesh::cell_type &cl, const Function &bf, std::vector<int64_t> &gidx,
                   std::vector<T> &dir)
    {
        const size_t cbs = numbering.cbs, fbs = numbering.fbs, c = offset(msh, cl);
        const auto fids = proton_amd::face_offsets(msh, cl);
        gidx.assign(cbs + 4 * fbs, -1);
        dir.assign(cbs + 4 * fbs, T(0));
        for (size_t i = 0; i < cbs; ++i) gidx[i] = (int64_t)(c * cbs + i);
        for (size_t lf = 0; lf < 4; ++lf) {
            const int64_t comp = numbering.compress[fids[lf]];
            for (size_t k = 0; k < fbs; ++k) {
                const size_t l = cbs + lf * fbs + k;
                if (comp >= 0) gidx[l] = (int64_t)(cbs * ncells + (size_t)comp * fbs + k);
                else dir[l] = numbering.dirichlet_data(msh, bf, c)[fids[lf] * fbs + k];
            }
        }
    }

  public:
    proton_amd::sparse_matrix<T> LHS;
    std::vector<T> RHS;

    assembler(const Mesh &msh, hho_degree_info hdi) : di(hdi), numbering(msh, hdi), ncells(msh.cells.size())
    {
        const size_t system_size = numbering.cbs * ncells + numbering.fbs * numbering.num_other_faces;      // hho.hpp:331
        LHS.nrows = LHS.ncols = system_size;
        RHS.assign(system_size, T(0));
    }

    // hho.hpp:344-406: triplets of the assembled (row, column) pairs in local row-major order,
    // Dirichlet columns moved to the right-hand side, cell part of rhs added
    template <typename Function>
    void assemble(const Mesh &msh, const typename Mesh::cell_type &cl, const proton_amd::dense_matrix<T> &lhs,
                  const proton_amd::dense_matrix<T> &rhs, const Function &dirichlet_bf)
    {
        std::vector<int64_t> gidx;
        std::vector<T> dir;
        local_map(msh, cl, dirichlet_bf, gidx, dir);
        const size_t ms = gidx.size();
        if (lhs.rows() != ms || lhs.cols() != ms) throw std::invalid_argument("assembler::assemble: local matrix size");
        for (size_t i = 0; i < ms; ++i) {
            if (gidx[i] < 0) continue;
            for (size_t j = 0; j < ms; ++j) {
                if (gidx[j] >= 0) triplets.emplace_back((int32_t)gidx[i], (int32_t)gidx[j], lhs(i, j));
                else RHS[gidx[i]] -= lhs(i, j) * dir[j];             // term by term, the reference's order (hho.hpp:401)
            }
        }
        for (size_t i = 0; i < numbering.cbs; ++i) RHS[gidx[i]] += rhs(i);
    }

    // hho.hpp:408-449
    template <typename Function>
    proton_amd::dense_matrix<T> take_local_data(const Mesh &msh, const typename Mesh::cell_type &cl,
                                                 const std::vector<T> &solution, const Function &dirichlet_bf)
    {
        std::vector<int64_t> gidx;
        std::vector<T> dir;
        local_map(msh, cl, dirichlet_bf, gidx, dir);
        proton_amd::dense_matrix<T> ret(gidx.size(), 1);
        for (size_t l = 0; l < gidx.size(); ++l) ret(l) = gidx[l] >= 0 ? solution[gidx[l]] : dir[l];
        return ret;
    }

    // hho.hpp:451-455: LHS.setFromTriplets (duplicates summed).  After assemble_all() the CSR was
    // already built on the device.
    void finalize(void)
    {
        if (device_csr_) { device_csr_ = false; return; }
        LHS.set_from_triplets(RHS.size(), triplets);
        triplets.clear();
    }

  private:
    bool device_csr_ = false;

  public:
    // Batched equivalent of the whole loop of convergence_test.cpp:202-215 with built-in source
    // terms: local operators, right-hand sides, Dirichlet data and triplets on the device.
    void assemble_all(const Mesh &msh, int stab_kind, int rhs_fn, int dirichlet_fn)
    {
        auto &dev = proton_amd::device::instance();
        proton_amd::batch_cache<Mesh>::instance().ensure_mesh(msh);
        pa_sizes sz;
        dev.check(pa_sizes_for(di.c_abi(), Mesh::pa_quadrature, &sz), "pa_sizes_for");
        const size_t n = msh.cells.size(), ms = sz.msize, mm = ms * ms;
        proton_amd::device_buffer<double> d_lc(n * mm), d_rhs(n * sz.cbs), d_g(msh.faces.size() * sz.fbs), d_vals(n * mm), d_rv(n * ms);
        proton_amd::device_buffer<int32_t> d_rows(n * mm), d_cols(n * mm), d_rr(n * ms);
        dev.check(pa_local_ops_batch(dev.ctx(), di.c_abi(), Mesh::pa_quadrature, stab_kind, 0, n, nullptr, nullptr, nullptr, d_lc.get(), nullptr), "pa_local_ops_batch");
        dev.check(pa_cell_rhs_batch(dev.ctx(), (int)di.cell_degree(), 0, Mesh::pa_quadrature, rhs_fn, nullptr, 0, n, d_rhs.get()), "pa_cell_rhs_batch");
        dev.check(pa_dirichlet_data_batch(dev.ctx(), (int)di.face_degree(), dirichlet_fn, nullptr, d_g.get()), "pa_dirichlet_data_batch");
        dev.check(pa_triplets_batch(dev.ctx(), di.c_abi(), 0, n, d_lc.get(), d_rhs.get(), d_g.get(), d_rows.get(), d_cols.get(),
                                    d_vals.get(), d_rr.get(), d_rv.get()), "pa_triplets_batch");
        // setFromTriplets on the device (pa_csr_from_triplets): only the CSR arrays come back
        proton_amd::device_buffer<int64_t> d_rowptr(RHS.size() + 1);
        proton_amd::device_buffer<int32_t> d_colind(n * mm);
        proton_amd::device_buffer<double> d_values(n * mm);
        size_t nnz = 0;
        dev.check(pa_csr_from_triplets(dev.ctx(), n * mm, d_rows.get(), d_cols.get(), d_vals.get(), RHS.size(), d_rowptr.get(),
                                       d_colind.get(), d_values.get(), &nnz), "pa_csr_from_triplets");
        LHS.nrows = LHS.ncols = RHS.size();
        LHS.rowptr.resize(RHS.size() + 1); LHS.colind.resize(nnz); LHS.values.resize(nnz);
        d_rowptr.download(LHS.rowptr.data(), LHS.rowptr.size());
        if (nnz) { d_colind.download(LHS.colind.data(), nnz); d_values.download(LHS.values.data(), nnz); }
        device_csr_ = true;
        std::vector<int32_t> rr(n * ms);
        std::vector<double> rv(n * ms);
        d_rr.download(rr.data(), rr.size()); d_rv.download(rv.data(), rv.size());
        for (size_t k = 0; k < rr.size(); ++k)
            if (rr[k] >= 0) RHS[rr[k]] += rv[k];
    }
};

template <typename Mesh>
auto make_assembler(const Mesh &msh, hho_degree_info hdi)
{
    return assembler<Mesh>(msh, hdi);
}

// solver_cg.hpp:38-144: the reference's (optionally Jacobi-preconditioned) conjugate gradient,
// run on the device over the CSR matrix (pa_conjugated_gradient)
enum class cg_exit_reason { CONVERGED, DIVERGED, MAX_ITER_REACHED };

template <typename T>
struct cg_params {
    T convergence_threshold, divergence_threshold;
    size_t max_iter;
    bool verbose, apply_preconditioner;
    std::string histfile;
    cg_params() : convergence_threshold(1e-9), divergence_threshold(100), max_iter(1000), verbose(false), apply_preconditioner(false) {}
};

template <typename T>
cg_exit_reason conjugated_gradient(const proton_amd::sparse_matrix<T> &A, const std::vector<T> &b, std::vector<T> &x,
                                   const cg_params<T> &parms = cg_params<T>(), size_t *iterations = nullptr)
{
    auto &dev = proton_amd::device::instance();
    const size_t n = A.rows(), nnz = A.nonZeros();
    proton_amd::device_buffer<int64_t> d_rowptr(n + 1);
    proton_amd::device_buffer<int32_t> d_colind(nnz + 1);
    proton_amd::device_buffer<double> d_values(nnz + 1), d_b(n + 1), d_x(n + 1);
    d_rowptr.upload(A.rowptr.data(), n + 1);
    if (nnz) { d_colind.upload(A.colind.data(), nnz); d_values.upload(A.values.data(), nnz); }
    if (n) d_b.upload(b.data(), n);
    int32_t reason = 0;
    size_t iters = 0;
    double rr = 0.0;
    dev.check(pa_conjugated_gradient(dev.ctx(), n, d_rowptr.get(), d_colind.get(), d_values.get(), d_b.get(), d_x.get(),
                                     parms.convergence_threshold, parms.divergence_threshold, parms.max_iter,
                                     parms.apply_preconditioner ? 1 : 0, &reason, &iters, &rr), "pa_conjugated_gradient");
    x.resize(n);
    if (n) d_x.download(x.data(), n);
    if (parms.verbose) std::cout << " -> Iteration " << iters << ", rr = " << rr << std::endl;
    if (iterations) *iterations = iters;
    return reason == 0 ? cg_exit_reason::CONVERGED : reason == 1 ? cg_exit_reason::DIVERGED : cg_exit_reason::MAX_ITER_REACHED;
}

// hho.hpp:471-751.  Unknowns: cells outside the active set, then the non-Dirichlet faces, then one
// multiplier per active cell; equations: one per cell (row = cell id), then the faces.  As in the
// reference the row numbering does not multiply by cbs (hho.hpp:631, 686): it is meant for cbs = 1,
// which is what apps/obstacle uses (obstacle.cpp:51).
template <typename Mesh>
class obstacle_assembler {
    using T = typename Mesh::coordinate_type;
    hho_degree_info di;
    proton_amd::face_numbering<Mesh> numbering;
    std::vector<bool> is_in_set_A;
    std::vector<int64_t> kept_pos, active_pos;          // A_ct / B_ct of the reference, -1 where undefined
    size_t num_all_cells, num_all_faces, num_A_cells = 0, num_I_cells = 0;
    std::vector<std::tuple<int32_t, int32_t, T>> triplets;

    size_t multiplier_base() const { return numbering.cbs * num_I_cells + numbering.fbs * numbering.num_other_faces; }

  public:
    proton_amd::sparse_matrix<T> LHS;
    std::vector<T> RHS;

    obstacle_assembler(const Mesh &msh, const std::vector<bool> &in_A, hho_degree_info hdi)
        : di(hdi), numbering(msh, hdi), is_in_set_A(in_A), num_all_cells(msh.cells.size()), num_all_faces(msh.faces.size())
    {
        if (is_in_set_A.size() != num_all_cells) throw std::invalid_argument("obstacle_assembler: in_A size");
        kept_pos.assign(num_all_cells, -1);
        active_pos.assign(num_all_cells, -1);
        for (size_t c = 0; c < num_all_cells; ++c) {                       // hho.hpp:538-578
            if (is_in_set_A[c]) active_pos[c] = (int64_t)num_A_cells++;
            else kept_pos[c] = (int64_t)num_I_cells++;
        }
        const size_t system_size = numbering.cbs * (num_I_cells + num_A_cells) + numbering.fbs * numbering.num_other_faces;   // :586
        LHS.nrows = LHS.ncols = system_size;
        RHS.assign(system_size, T(0));
    }

    // hho.hpp:609-695
    template <typename Function>
    void assemble(const Mesh &msh, const typename Mesh::cell_type &cl, const proton_amd::dense_matrix<T> &lhs,
                  const proton_amd::dense_matrix<T> &rhs, const std::vector<T> &gamma, const Function &dirichlet_bf)
    {
        const size_t cbs = numbering.cbs, fbs = numbering.fbs, ms = cbs + 4 * fbs, c = offset(msh, cl);
        const bool active = is_in_set_A[c];
        const auto fids = proton_amd::face_offsets(msh, cl);
        std::vector<int64_t> row(ms, -1), col(ms, -1);
        std::vector<T> known(ms, T(0));                 // value a dropped column is multiplied with
        for (size_t i = 0; i < cbs; ++i) {
            row[i] = (int64_t)(c + i);
            if (!active) col[i] = kept_pos[c] * (int64_t)cbs + (int64_t)i;
            else known[i] = gamma[c];                                       // :677
        }
        for (size_t lf = 0; lf < 4; ++lf) {
            const int64_t comp = numbering.compress[fids[lf]];
            for (size_t k = 0; k < fbs; ++k) {
                const size_t l = cbs + lf * fbs + k;
                if (comp >= 0) {
                    row[l] = (int64_t)(cbs * num_all_cells + (size_t)comp * fbs + k);     // :644
                    col[l] = (int64_t)(cbs * num_I_cells + (size_t)comp * fbs + k);       // :645
                } else {
                    known[l] = numbering.dirichlet_data(msh, dirichlet_bf, c)[fids[lf] * fbs + k];
                }
            }
        }
        if (lhs.rows() != ms || lhs.cols() != ms) throw std::invalid_argument("obstacle_assembler::assemble: local matrix size");
        for (size_t i = 0; i < ms; ++i) {
            if (row[i] < 0) continue;
            for (size_t j = 0; j < ms; ++j) {
                if (col[j] >= 0) triplets.emplace_back((int32_t)row[i], (int32_t)col[j], lhs(i, j));
                else RHS[row[i]] -= lhs(i, j) * known[j];      // term by term, the reference's order
            }
        }
        for (size_t i = 0; i < cbs; ++i) RHS[c + i] += rhs(i);                           // :686
        if (active)                                                                       // :688-693
            triplets.emplace_back((int32_t)(c * cbs), (int32_t)(multiplier_base() + (size_t)active_pos[c]), T(1));
    }

    // hho.hpp:698-744: alpha = (cell dofs, then ALL face dofs), beta = multipliers per cell
    template <typename Function>
    void expand_solution(const Mesh &msh, const std::vector<T> &solution, const Function &dirichlet_bf,
                         const std::vector<T> &gamma, std::vector<T> &alpha, std::vector<T> &beta)
    {
        const size_t cbs = numbering.cbs, fbs = numbering.fbs;
        alpha.resize(num_all_cells * cbs + num_all_faces * fbs);
        beta.assign(num_all_cells * cbs, T(0));
        for (size_t c = 0; c < num_all_cells; ++c)
            for (size_t k = 0; k < cbs; ++k) {
                if (is_in_set_A[c]) {
                    alpha[c * cbs + k] = gamma[c * cbs + k];
                    beta[c * cbs + k] = solution[multiplier_base() + (size_t)active_pos[c] * cbs + k];
                } else {
                    alpha[c * cbs + k] = solution[(size_t)kept_pos[c] * cbs + k];
                }
            }
        const auto &g = numbering.dirichlet_data(msh, dirichlet_bf);
        for (size_t f = 0; f < num_all_faces; ++f)
            for (size_t k = 0; k < fbs; ++k)
                alpha[num_all_cells * cbs + f * fbs + k] =
                    numbering.compress[f] < 0 ? g[f * fbs + k] : solution[cbs * num_I_cells + (size_t)numbering.compress[f] * fbs + k];
    }

    void finalize(void)
    {
        if (device_csr_) { device_csr_ = false; return; }       // assemble_all() built the CSR on the device
        LHS.set_from_triplets(RHS.size(), triplets);
        triplets.clear();
    }

  private:
    bool device_csr_ = false;

  public:
    // The whole cell loop of obstacle.cpp:148-156 on the device (pa_obstacle_tables +
    // pa_obstacle_triplets_batch): d_lc / d_rhs / d_g are device arrays for all cells / faces.
    void assemble_all(const Mesh &msh, const double *d_lc, const double *d_rhs, const double *d_g, const std::vector<T> &gamma)
    {
        auto &dev = proton_amd::device::instance();
        proton_amd::batch_cache<Mesh>::instance().ensure_mesh(msh);
        const size_t n = num_all_cells, ms = numbering.cbs + 4 * numbering.fbs, slots = ms * ms + 1;
        std::vector<uint8_t> flags(n);
        for (size_t c = 0; c < n; ++c) flags[c] = is_in_set_A[c] ? 1 : 0;
        proton_amd::device_buffer<uint8_t> d_in(n);
        proton_amd::device_buffer<int32_t> d_a(n), d_b(n), d_rows(n * slots), d_cols(n * slots), d_rr(n * ms);
        proton_amd::device_buffer<double> d_gamma(n), d_vals(n * slots), d_rv(n * ms);
        d_in.upload(flags.data(), n);
        d_gamma.upload(gamma.data(), n);
        size_t ni = 0, na = 0;
        dev.check(pa_obstacle_tables(dev.ctx(), d_in.get(), d_a.get(), d_b.get(), &ni, &na), "pa_obstacle_tables");
        dev.check(pa_obstacle_triplets_batch(dev.ctx(), di.c_abi(), 0, n, d_lc, d_rhs, d_g, d_gamma.get(), d_in.get(), d_a.get(),
                                             d_b.get(), ni, d_rows.get(), d_cols.get(), d_vals.get(), d_rr.get(), d_rv.get()),
                  "pa_obstacle_triplets_batch");
        proton_amd::device_buffer<int64_t> d_rowptr(RHS.size() + 1);          // setFromTriplets on the device
        proton_amd::device_buffer<int32_t> d_colind(n * slots);
        proton_amd::device_buffer<double> d_values(n * slots);
        size_t nnz = 0;
        dev.check(pa_csr_from_triplets(dev.ctx(), n * slots, d_rows.get(), d_cols.get(), d_vals.get(), RHS.size(), d_rowptr.get(),
                                       d_colind.get(), d_values.get(), &nnz), "pa_csr_from_triplets");
        LHS.nrows = LHS.ncols = RHS.size();
        LHS.rowptr.resize(RHS.size() + 1); LHS.colind.resize(nnz); LHS.values.resize(nnz);
        d_rowptr.download(LHS.rowptr.data(), LHS.rowptr.size());
        if (nnz) { d_colind.download(LHS.colind.data(), nnz); d_values.download(LHS.values.data(), nnz); }
        device_csr_ = true;
        std::vector<int32_t> rr(n * ms);
        std::vector<double> rv(n * ms);
        d_rr.download(rr.data(), rr.size()); d_rv.download(rv.data(), rv.size());
        for (size_t k = 0; k < rr.size(); ++k)
            if (rr[k] >= 0) RHS[rr[k]] += rv[k];
    }
};

// hho.hpp:753-782
template <typename T, typename Mesh>
proton_amd::dense_matrix<T> take_local_data(const Mesh &msh, const typename Mesh::cell_type &cl, hho_degree_info di,
                                            const std::vector<T> &expanded_solution)
{
    const size_t cbs = (di.cell_degree() + 2) * (di.cell_degree() + 1) / 2, fbs = di.face_degree() + 1;
    const size_t c = offset(msh, cl);
    const auto fids = proton_amd::face_offsets(msh, cl);
    proton_amd::dense_matrix<T> ret(cbs + 4 * fbs, 1);
    for (size_t i = 0; i < cbs; ++i) ret(i) = expanded_solution[c * cbs + i];
    for (size_t lf = 0; lf < 4; ++lf)
        for (size_t k = 0; k < fbs; ++k) ret(cbs + lf * fbs + k) = expanded_solution[cbs * msh.cells.size() + fids[lf] * fbs + k];
    return ret;
}

template <typename Mesh>
auto make_obstacle_assembler(const Mesh &msh, const std::vector<bool> &in_A, hho_degree_info hdi)
{
    return obstacle_assembler<Mesh>(msh, in_A, hdi);
}
