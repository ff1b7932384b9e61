// cut_host.hpp -- host-side cutHHO preprocessing of the product path (C++).
//
// What the reference does on the host before the assembly loop of `cuthho_square -f`
// (apps/cuthho/cuthho_square.cpp:2036-2052) and inside the cut integrate() overloads:
//   detect_node_position   src/methods/cuthho_bits/cuthho_geom.hpp:118-130
//   detect_cut_faces       :132-161   (find_zero_crossing :68-116)
//   move_nodes             :466-543   (the default -D path)
//   detect_cut_cells       :275-340
//   refine_interface       :609-673
//   collect_triangulation_points / triangulate / integrate(cell|face) / integrate_interface  :675-895
// Here it produces, per cut cell, flat quadrature lists that the cut-cell kernel consumes.
// Its own data model (struct-of-arrays over the generator mesh), not the reference's classes.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <thread>
#include <vector>

#include "hho_assembly.hpp"      // StructuredMesh closed forms
#include "hho_device.hpp"        // QuadTables

namespace pa {

enum : int8_t { LOC_NEG = 0, LOC_POS = 1, LOC_CUT = 2, LOC_UNDEF = 3 };

struct LevelSet {
    int kind;                    // 0 circle (x-a)^2 + (y-b)^2 - r^2, 1 line y - cut_y   cuthho_square.cpp:56-124
    double radius, alpha, beta, cut_y;
    __host__ __device__ double operator()(double x, double y) const
    {
        return kind == 0 ? (x - alpha) * (x - alpha) + (y - beta) * (y - beta) - radius * radius : y - cut_y;
    }
    __host__ __device__ void normal(double x, double y, double &nx, double &ny) const
    {
        double gx = kind == 0 ? 2 * x - 2 * alpha : 0.0, gy = kind == 0 ? 2 * y - 2 * beta : 1.0;
        const double nrm = sqrt(gx * gx + gy * gy);
        nx = gx / nrm; ny = gy / nrm;
    }
};

// The O(cells) sweeps of the preprocessing (level-set evaluation at every node, tags of every face
// and cell) are independent per element: contiguous index ranges on the host's hardware threads.
// Every element sees the same arithmetic as in a serial sweep, so the results are bit-identical.
template <typename Fn>
inline void parallel_ranges(size_t n, Fn fn)
{
    const size_t hw = std::max<size_t>(1, std::min<size_t>(16, std::thread::hardware_concurrency()));
    const size_t nt = n < 65536 ? 1 : hw;
    if (nt == 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(nt);
    for (size_t t = 0; t < nt; ++t)
        pool.emplace_back([&, t]() {
            try { fn(n * t / nt, n * (t + 1) / nt); } catch (...) { errs[t] = std::current_exception(); }
        });
    for (auto &th : pool) th.join();
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
}

struct P2d { double x, y; };
inline P2d operator+(P2d a, P2d b) { return {a.x + b.x, a.y + b.y}; }
inline P2d operator-(P2d a, P2d b) { return {a.x - b.x, a.y - b.y}; }
inline P2d operator*(P2d a, double s) { return {a.x * s, a.y * s}; }
inline P2d operator/(P2d a, double s) { return {a.x / s, a.y / s}; }
inline double norm(P2d a) { return std::sqrt(a.x * a.x + a.y * a.y); }

// bisection of cuthho_geom.hpp:68-116 (30 halvings at most, stops when the midpoint moves less than `threshold`)
inline P2d zero_crossing(P2d pa, P2d pb, const LevelSet &ls, double threshold)
{
    P2d pm = (pa + pb) / 2.0, prev = pm;
    size_t budget = 30;
    bool again;
    do {
        const double lb = ls(pb.x, pb.y), lm = ls(pm.x, pm.y);
        prev = pm;
        if ((lb >= 0 && lm >= 0) || (lb < 0 && lm < 0)) pb = pm; else pa = pm;
        pm = (pa + pb) / 2.0;
        const double dx = prev.x - pm.x, dy = prev.y - pm.y;
        again = std::sqrt(dx * dx + dy * dy) > threshold && (budget-- != 0);
    } while (again);
    return pm;
}

// Tagged, displaced mesh + interface polylines of the cut cells
struct CutMeshHost {
    StructuredMesh sm{};
    std::vector<double> pts;                 // np x 2 (displaced)
    std::vector<int8_t> node_loc, face_loc, cell_loc;
    std::vector<P2d> face_ip;
    std::vector<uint32_t> cut_cells;         // ids of the cut cells, ascending
    std::vector<int32_t> cut_index;          // cell -> position in cut_cells or -1
    std::vector<P2d> iface;                  // cut_cells.size() x nif
    size_t nif = 2;

    size_t npoints() const { return (size_t)(sm.Nx + 1) * (sm.Ny + 1); }
    size_t ncells() const { return (size_t)sm.Nx * sm.Ny; }
    size_t nfaces() const { return (size_t)sm.Nx * (sm.Ny + 1) + (size_t)sm.Ny * (sm.Nx + 1); }
    P2d point(uint32_t id) const { return {pts[2 * (size_t)id], pts[2 * (size_t)id + 1]}; }
    void cell_ids(uint32_t c, uint32_t ids[4]) const
    {
        const uint32_t i = c % sm.Nx, j = c / sm.Nx, p0 = j * (sm.Nx + 1) + i;
        ids[0] = p0; ids[1] = p0 + 1; ids[2] = p0 + sm.Nx + 2; ids[3] = p0 + sm.Nx + 1;     // basic_mesh.hpp:362-368
    }
    void cell_face_ids(uint32_t c, uint32_t f[4]) const
    {
        const uint32_t i = c % sm.Nx, j = c / sm.Nx;
        f[0] = sm_hface(sm, i, j); f[1] = sm_vface(sm, i + 1, j); f[2] = sm_hface(sm, i, j + 1); f[3] = sm_vface(sm, i, j);
    }
    void face_ends(uint32_t f, uint32_t &lo, uint32_t &hi) const
    {
        bool d; int32_t comp;
        sm_face_decode(sm, f, lo, hi, d, comp);
    }
};

inline void tag_faces(CutMeshHost &m, const LevelSet &ls)                      // detect_cut_faces
{
    const size_t nf = m.nfaces();
    m.face_loc.assign(nf, LOC_UNDEF);
    m.face_ip.assign(nf, P2d{0, 0});
    parallel_ranges(nf, [&](size_t f0, size_t f1) {
        for (uint32_t f = (uint32_t)f0; f < f1; ++f) {
            uint32_t lo, hi;
            m.face_ends(f, lo, hi);
            const P2d p0 = m.point(lo), p1 = m.point(hi);
            const double l0 = ls(p0.x, p0.y), l1 = ls(p1.x, p1.y);
            if (l0 >= 0 && l1 >= 0) { m.face_loc[f] = LOC_POS; continue; }
            if (l0 < 0 && l1 < 0) { m.face_loc[f] = LOC_NEG; continue; }
            m.face_ip[f] = zero_crossing(p0, p1, ls, norm(p1 - p0) / 1e4);
            m.face_loc[f] = LOC_CUT;
        }
    });
}

// Runs the preprocessing of cuthho_square.cpp:2036-2052: with node displacement (-D, the default) or,
// displace = false, the agglomeration branch (-A: no displacement; detect_cell_agglo_set is
// classify_agglomeration below).  Throws std::logic_error like the reference on bad cuts.
inline void cut_preprocess(CutMeshHost &m, uint32_t Nx, uint32_t Ny, double min_x, double max_x, double min_y, double max_y,
                           const LevelSet &ls, int refsteps, bool displace = true)
{
    m.sm = StructuredMesh{Nx, Ny, 0, Ny};
    const size_t np = m.npoints(), nc = m.ncells(), nf = m.nfaces();
    const double hx = (max_x - min_x) / Nx, hy = (max_y - min_y) / Ny;
    m.pts.resize(2 * np);
    m.node_loc.resize(np);
    parallel_ranges(np, [&](size_t n0, size_t n1) {
        for (size_t n = n0; n < n1; ++n) {
            const size_t i = n % (Nx + 1), j = n / (Nx + 1);
            m.pts[2 * n] = min_x + i * hx;
            m.pts[2 * n + 1] = min_y + j * hy;
            m.node_loc[n] = ls(m.pts[2 * n], m.pts[2 * n + 1]) < 0 ? LOC_NEG : LOC_POS;      // detect_node_position
        }
    });
    tag_faces(m, ls);
    if (displace) {                                                               // move_nodes
        std::vector<P2d> disp(np, P2d{0, 0});
        std::vector<uint8_t> moved(np, 0);
        for (uint32_t f = 0; f < nf; ++f) {
            if (m.face_loc[f] != LOC_CUT) continue;
            uint32_t lo, hi;
            m.face_ends(f, lo, hi);
            const P2d p0 = m.point(lo), p1 = m.point(hi), bar = (p1 + p0) / 2.0;
            const double closeness = norm(m.face_ip[f] - p0) / norm(p1 - p0);
            uint32_t ntc;
            if (closeness < 0.4) ntc = lo; else if (closeness > 1.0 - 0.4) ntc = hi; else continue;
            disp[ntc] = disp[ntc] - (bar - m.face_ip[f]) / 2;
            moved[ntc] = 1;
        }
        for (size_t n = 0; n < np; ++n)
            if (moved[n]) { m.pts[2 * n] += disp[n].x; m.pts[2 * n + 1] += disp[n].y; }
        parallel_ranges(nc, [&](size_t c0, size_t c1) {
            for (uint32_t c = (uint32_t)c0; c < c1; ++c) {
                uint32_t ids[4];
                m.cell_ids(c, ids);
                if (!(moved[ids[0]] | moved[ids[1]] | moved[ids[2]] | moved[ids[3]])) continue;
                for (int i = 0; i < 4; ++i) {
                    const P2d v1 = m.point(ids[(i + 1) % 4]) - m.point(ids[i]), v2 = m.point(ids[(i + 2) % 4]) - m.point(ids[(i + 1) % 4]);
                    if (v1.x * v2.y - v2.x * v1.y < 0) throw std::logic_error("concave poly");
                }
            }
        });
    }
    if (displace) tag_faces(m, ls);                                               // again: updated intersection points
    m.cell_loc.assign(nc, LOC_UNDEF);                                             // detect_cut_cells
    m.cut_index.assign(nc, -1);
    m.cut_cells.clear();
    parallel_ranges(nc, [&](size_t c0, size_t c1) {                               // tags: independent per cell
        for (uint32_t c = (uint32_t)c0; c < c1; ++c) {
            uint32_t fcs[4], ids[4];
            m.cell_face_ids(c, fcs);
            m.cell_ids(c, ids);
            int k = 0;
            for (int i = 0; i < 4; ++i) k += m.face_loc[fcs[i]] == LOC_CUT;
            if (k == 0) {
                bool all_pos = true;
                for (int v = 0; v < 4; ++v) { const P2d p = m.point(ids[v]); all_pos = all_pos && ls(p.x, p.y) > 0; }
                m.cell_loc[c] = all_pos ? LOC_POS : LOC_NEG;
            } else if (k == 2) m.cell_loc[c] = LOC_CUT;
            else throw std::logic_error("invalid number of cuts in cell");
        }
    });
    std::vector<P2d> p0p1;
    for (uint32_t c = 0; c < nc; ++c) {                                           // cut cells in ascending order, interface endpoints
        if (m.cell_loc[c] != LOC_CUT) continue;
        uint32_t fcs[4];
        m.cell_face_ids(c, fcs);
        int k = 0;
        P2d cut[2] = {{0, 0}, {0, 0}};
        for (int i = 0; i < 4; ++i)
            if (m.face_loc[fcs[i]] == LOC_CUT) cut[k++] = m.face_ip[fcs[i]];
        const P2d pt = cut[1] - cut[0], pn = cut[0] + P2d{-pt.y, pt.x};
        const bool swap = ls(pn.x, pn.y) >= 0;
        m.cut_index[c] = (int32_t)m.cut_cells.size();
        m.cut_cells.push_back(c);
        p0p1.push_back(swap ? cut[1] : cut[0]);
        p0p1.push_back(swap ? cut[0] : cut[1]);
    }
    // refine_interface: 2^refsteps segments per cut cell, midpoints pushed onto the interface by bisection
    const size_t nseg = (size_t)1 << refsteps;
    m.nif = nseg + 1;
    m.iface.assign(m.cut_cells.size() * m.nif, P2d{0, 0});
    for (size_t cc = 0; cc < m.cut_cells.size(); ++cc) {
        P2d *ifc = &m.iface[cc * m.nif];
        ifc[0] = p0p1[2 * cc]; ifc[nseg] = p0p1[2 * cc + 1];
        uint32_t ids[4];
        m.cell_ids(m.cut_cells[cc], ids);
        double diam = 0.0;
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b) diam = std::max(diam, norm(m.point(ids[b]) - m.point(ids[a])));
        for (size_t span = nseg; span >= 2; span /= 2)                 // same midpoints as the recursion of :609-651
            for (size_t lo = 0; lo + span <= nseg; lo += span) {
                const P2d a = ifc[lo], b = ifc[lo + span], pm = (a + b) / 2.0, pt = b - a, pn{-pt.y, pt.x};
                const P2d s1 = pm + pn, s2 = pm - pn;
                const double lm = ls(pm.x, pm.y), l1 = ls(s1.x, s1.y), l2 = ls(s2.x, s2.y);
                P2d ip;
                if (!((lm >= 0 && l1 >= 0) || (lm < 0 && l1 < 0))) ip = zero_crossing(pm, s1, ls, diam / 10000.0);
                else if (!((lm >= 0 && l2 >= 0) || (lm < 0 && l2 < 0))) ip = zero_crossing(pm, s2, ls, diam / 10000.0);
                else throw std::logic_error("interface not found in search range");
                ifc[lo + span / 2] = ip;
            }
    }
}

// detect_cell_agglo_set (cuthho_geom.hpp:163-273): for every cut cell, how close the interface
// passes to the cell's nodes along the two cut faces (threshold 0.3 of the face length).
// 0 UNDEF (not cut), 1 T_OK, 2 T_KO_NEG, 3 T_KO_POS -- the reference's cell_agglo_set order.
enum : int8_t { AGGLO_UNDEF = 0, AGGLO_T_OK = 1, AGGLO_T_KO_NEG = 2, AGGLO_T_KO_POS = 3 };

inline void classify_agglomeration(const CutMeshHost &m, std::vector<int8_t> &agglo)
{
    const double threshold = 0.3;                                                 // :170
    const size_t nc = m.ncells();
    agglo.assign(nc, AGGLO_UNDEF);
    for (uint32_t c = 0; c < nc; ++c) {
        uint32_t fcs[4], ids[4];
        m.cell_face_ids(c, fcs);
        m.cell_ids(c, ids);
        bool cut[4];
        for (int i = 0; i < 4; ++i) cut[i] = m.face_loc[fcs[i]] == LOC_CUT;
        auto rel_dist = [&](int node, int face) {                                 // |p_node - ip(face)| / |face|
            uint32_t lo, hi;
            m.face_ends(fcs[face], lo, hi);
            return norm(m.point(ids[node]) - m.face_ip[fcs[face]]) / norm(m.point(hi) - m.point(lo));
        };
        for (int i = 0; i < 4; ++i) {                                             // two consecutive cut faces: one node cut off (:182-206, :243-252)
            const int f1 = i, f2 = (i + 1) % 4, n = (i + 1) % 4;
            if (!(cut[f1] && cut[f2])) continue;
            const double da = rel_dist(n, f1), db = rel_dist(n, f2);             // faces n-1 and n of node n
            if (std::min(da, db) > threshold) agglo[c] = AGGLO_T_OK;
            else agglo[c] = m.node_loc[ids[n]] == LOC_NEG ? AGGLO_T_KO_NEG : AGGLO_T_KO_POS;
        }
        for (int f1 = 0; f1 < 2; ++f1) {                                          // two opposite cut faces (:208-241, :254-258)
            const int f2 = f1 + 2;
            if (!(cut[f1] && cut[f2])) continue;
            const int n1 = f1, n2 = (f2 + 1) % 4;
            const double da = rel_dist(n1, f1), db = rel_dist(n2, f2);
            const double m1 = std::max(da, db), m2 = std::max(1 - da, 1 - db);
            if (std::min(m1, m2) > threshold) { agglo[c] = AGGLO_T_OK; continue; }
            if (m.node_loc[ids[n1]] == LOC_NEG) agglo[c] = (m1 <= threshold) ? AGGLO_T_KO_NEG : AGGLO_T_KO_POS;
            else agglo[c] = (m2 <= threshold) ? AGGLO_T_KO_NEG : AGGLO_T_KO_POS;
        }
    }
}

// make_neighbors_info (cuthho_geom.hpp:343-370: cells sharing at least one point, found there by an
// O(cells^2) search) in closed form on the generator mesh: the up-to-8 surrounding cells, ascending,
// -1 padded.
inline void structured_neighbors(const StructuredMesh &sm, std::vector<int32_t> &nb)
{
    nb.assign((size_t)sm.Nx * sm.Ny * 8, -1);
    for (uint32_t j = 0; j < sm.Ny; ++j)
        for (uint32_t i = 0; i < sm.Nx; ++i) {
            int k = 0;
            for (int dj = -1; dj <= 1; ++dj)
                for (int di = -1; di <= 1; ++di) {
                    if (!di && !dj) continue;
                    const int64_t ii = (int64_t)i + di, jj = (int64_t)j + dj;
                    if (ii < 0 || jj < 0 || ii >= sm.Nx || jj >= sm.Ny) continue;
                    nb[((size_t)j * sm.Nx + i) * 8 + k++] = (int32_t)(jj * sm.Nx + ii);
                }
        }
}

// ---- quadrature lists ---------------------------------------------------------------------
// Per cut cell (record r): cell points [cell_off[r], cell_off[r+1]), interface points (laplacian
// degree) [il_off..], interface points (rhs degree) [ir_off..]; face points: fixed slots,
// FACE_SLOTS per face, counts in fl_cnt / fs_cnt (laplacian degree 2 recdeg / stabilization degree 2 facdeg).
constexpr int FACE_SLOTS = 5;

struct CutLists {
    std::vector<uint32_t> cell_off, il_off, ir_off;
    std::vector<double> cell_xyw, il_xyw, ir_xyw;             // 3 doubles per point
    std::vector<double> fl_xyw, fs_xyw;                       // ncut x 4 x FACE_SLOTS x 3
    std::vector<int32_t> fl_cnt, fs_cnt;                      // ncut x 4
};

inline void polygon_where(const CutMeshHost &m, size_t cc, int where, std::vector<P2d> &tp)     // collect_triangulation_points
{
    const uint32_t c = m.cut_cells[cc];
    uint32_t ids[4];
    m.cell_ids(c, ids);
    const P2d *ifc = &m.iface[cc * m.nif];
    bool in[4];
    for (int v = 0; v < 4; ++v) in[v] = m.node_loc[ids[v]] == where;
    tp.clear();
    auto insert_interface = [&]() {
        if (where == LOC_NEG) for (size_t i = 0; i < m.nif; ++i) tp.push_back(ifc[i]);
        else for (size_t i = m.nif; i-- > 0;) tp.push_back(ifc[i]);
    };
    if (!(in[0] && in[3])) {                                   // cases 1-3 of :706-717
        for (int v = 0; v < 4; ++v) if (in[v]) tp.push_back(m.point(ids[v]));
        insert_interface();
    } else {                                                   // case 4: the in-side run wraps around
        int i = 0;
        while (i < 4 && in[i]) tp.push_back(m.point(ids[i++]));
        insert_interface();
        while (i < 4 && !in[i]) ++i;
        while (i < 4 && in[i]) tp.push_back(m.point(ids[i++]));
    }
}

inline P2d polygon_centroid(const std::vector<P2d> &tp)        // barycenter(begin, end) basic_geom.hpp:247-270
{
    P2d acc{0, 0};
    double den = 0.0;
    for (size_t i = 2; i < tp.size(); ++i) {
        const P2d a = tp[i - 1] - tp[0], b = tp[i] - tp[0];
        const double d = (a.x * b.y - a.y * b.x) / 2.0;
        acc = acc + (a + b) * d;
        den += d;
    }
    return tp[0] + acc / (den * 3);
}

inline void gauss_segment(const QuadTables &t, int degree, P2d a, P2d b, double sign, std::vector<double> &out)
{
    const int n = gauss_nodes(degree);
    const double meas = norm(b - a);
    for (int q = 0; q < n; ++q) {
        const double tq = t.gauss_x[n][q];
        out.push_back(0.5 * (1 - tq) * a.x + 0.5 * (1 + tq) * b.x);              // quadratures.hpp:420-428 / cuthho_geom.hpp:838-845
        out.push_back(0.5 * (1 - tq) * a.y + 0.5 * (1 + tq) * b.y);
        out.push_back(sign * t.gauss_w[n][q] * meas * 0.5);
    }
}

// recdeg = facdeg + 1 = celdeg is what the cut operators assume (cuthho_square.cpp:871, :381)
inline void build_cut_lists(const CutMeshHost &m, const QuadTables &t, int facdeg, int where, CutLists &L)
{
    const int recdeg = facdeg + 1, celdeg = recdeg;
    const size_t ncut = m.cut_cells.size();
    if (2 * recdeg > 8 || t.dun_n[2 * recdeg] == 0) throw std::invalid_argument("Quadrature order too high");
    if (gauss_nodes(2 * recdeg) > FACE_SLOTS) throw std::invalid_argument("Quadrature order too high");
    L = CutLists();
    L.cell_off.assign(1, 0); L.il_off.assign(1, 0); L.ir_off.assign(1, 0);
    L.fl_xyw.assign(ncut * 4 * FACE_SLOTS * 3, 0.0); L.fs_xyw.assign(ncut * 4 * FACE_SLOTS * 3, 0.0);
    L.fl_cnt.assign(ncut * 4, 0); L.fs_cnt.assign(ncut * 4, 0);
    std::vector<P2d> tp;
    const int R = 2 * recdeg;                                                     // rules[deg]: the off-by-one of quadratures.hpp:257
    for (size_t cc = 0; cc < ncut; ++cc) {
        polygon_where(m, cc, where, tp);
        const P2d bar = polygon_centroid(tp);
        for (size_t i = 0; i < tp.size(); ++i) {                                  // triangulate + triangle_quadrature(bar, tp[i], tp[i+1])
            const P2d p1 = tp[i], p2 = tp[(i + 1) % tp.size()];
            const P2d v0 = p1 - bar, v1 = p2 - bar;
            const double area = std::fabs((v0.x * v1.y - v0.y * v1.x) / 2.0);
            for (int q = 0; q < t.dun_n[R]; ++q) {
                const double *row = t.dun[R][q];
                L.cell_xyw.push_back(bar.x * row[0] + p1.x * row[1] + p2.x * row[2]);
                L.cell_xyw.push_back(bar.y * row[0] + p1.y * row[1] + p2.y * row[2]);
                L.cell_xyw.push_back(area * row[3]);
            }
        }
        L.cell_off.push_back((uint32_t)(L.cell_xyw.size() / 3));
        // integrate_interface: orientation sign from the first segment (cuthho_geom.hpp:863-870)
        const P2d *ifc = &m.iface[cc * m.nif];
        const P2d va = ifc[0] - bar, vt = ifc[1] - ifc[0];
        const double sign = (va.x * vt.y + va.y * (-vt.x)) < 0 ? -1.0 : +1.0;
        for (size_t i = 1; i < m.nif; ++i) gauss_segment(t, 2 * recdeg, ifc[i - 1], ifc[i], sign, L.il_xyw);
        L.il_off.push_back((uint32_t)(L.il_xyw.size() / 3));
        for (size_t i = 1; i < m.nif; ++i) gauss_segment(t, celdeg, ifc[i - 1], ifc[i], sign, L.ir_xyw);     // quirk: degree, not 2*degree (:647)
        L.ir_off.push_back((uint32_t)(L.ir_xyw.size() / 3));
        // faces: the `where` part of each face (cuthho_geom.hpp:546-569, 817-849)
        uint32_t fcs[4];
        m.cell_face_ids(m.cut_cells[cc], fcs);
        for (int lf = 0; lf < 4; ++lf) {
            const uint32_t f = fcs[lf];
            if (m.face_loc[f] != where && m.face_loc[f] != LOC_CUT) continue;     // no points on this face
            uint32_t lo, hi;
            m.face_ends(f, lo, hi);
            P2d a = m.point(lo), b = m.point(hi);
            if (m.face_loc[f] == LOC_CUT) {
                const bool in0 = m.node_loc[lo] == where, in1 = m.node_loc[hi] == where;
                if (in0 && !in1) b = m.face_ip[f];
                else if (!in0 && in1) a = m.face_ip[f];
                else throw std::logic_error("Invalid point configuration");
            }
            std::vector<double> tmp;
            gauss_segment(t, 2 * recdeg, a, b, 1.0, tmp);
            std::copy(tmp.begin(), tmp.end(), L.fl_xyw.begin() + ((cc * 4 + lf) * FACE_SLOTS) * 3);
            L.fl_cnt[cc * 4 + lf] = (int32_t)(tmp.size() / 3);
            tmp.clear();
            gauss_segment(t, 2 * facdeg, a, b, 1.0, tmp);
            std::copy(tmp.begin(), tmp.end(), L.fs_xyw.begin() + ((cc * 4 + lf) * FACE_SLOTS) * 3);
            L.fs_cnt[cc * 4 + lf] = (int32_t)(tmp.size() / 3);
        }
    }
}

}  // namespace pa
