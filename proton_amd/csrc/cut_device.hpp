// cut_device.hpp -- the cut-cell local operators of the fictitious-domain driver:
//   make_hho_laplacian(msh, cl, level_set, di, where)   apps/cuthho/cuthho_square.cpp:308-388
//   make_hho_cut_stabilization                          apps/cuthho/cuthho_square.cpp:566-621
//   make_rhs(msh, cl, degree, f, where, level_set, bcs) apps/cuthho/cuthho_square.cpp:623-666
// for the cells cut by the interface (about 0.5 % of a 512 x 512 mesh).  One wavefront per cut
// cell; the quadrature lists come from the host preprocessing (cut_host.hpp).  The uncut cells
// of the same mesh go through hho_local_ops_kernel (fan quadrature, naive stabilization).
// Assumes celdeg == recdeg == facdeg + 1, as the reference's cut operators do (:381, :871).
//
// DD (the default): stages A-E -- everything between the scaled coordinates of the quadrature points and `oper` / `data` -- run in
// double-double arithmetic (dd_arith.hpp).  The Nitsche-penalised rbs x rbs system of a sliver cut is badly conditioned (1-norm
// condition numbers to 1.9e9 on the 512 x 512 mesh): in double, ROUNDING gr_lhs and gr_rhs once and solving exactly already
// costs 1e-11 in `data` (tests/test_oracle_cut_truth.py), and the reference's own operation order in double sits at 1e-10 -- no
// double evaluation is within 1e-12 of another there.  What may stay double without that amplification is what perturbs the
// bilinear forms CONSISTENTLY (the scaled coordinates bx, by of a point, 2/h, the normals, the face coordinate, eta / h_T: a
// slightly different quadrature point or penalty, the same on both sides of the solve); what may not are the sums, the products of
// the powers, the factorization, the substitutions and the final product.  With those in double-double every cut cell is within
// 1e-12 of the 50-digit / binary128 evaluation of the reference's formulas (measured: data <= 4e-14, oper <= 1.3e-13 on all 1 436
// cut cells of config 3).  Stabilization and right-hand side involve no such solve and stay in double (1e-13 everywhere).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "cut_host.hpp"
#include "dd_arith.hpp"
#include "hho_aux.hpp"
#include "hho_device.hpp"

namespace pa {

struct CutArgs {
    const QuadTables *tab;
    const double *points;
    const uint32_t *ptids;
    const uint32_t *cut_cells;
    uint32_t ncut;
    const uint32_t *cell_off, *il_off, *ir_off;
    const double *cell_xyw, *il_xyw, *ir_xyw, *fl_xyw, *fs_xyw;
    const int32_t *fl_cnt, *fs_cnt;
    LevelSet ls;
    int rhs_fn, bcs_fn;             // FN_SAMPLED: values per point of the cell list / of the rhs interface list
    const double *rhs_vals, *bcs_vals;
    double eta;                     // cell_eta, cuthho_square.cpp:301-306
    double *oper, *data, *stab, *lc, *rhs;
    int32_t *info;
    long long *dbg;                 // tuning builds (PA_CUT_CLOCK): shader-clock stamps at the stage boundaries of block 0
};
#define PA_CUT_TICK(i) do { if (a.dbg != nullptr && blockIdx.x == 0 && l == 0) a.dbg[i] = clock64(); } while (0)

__device__ __forceinline__ double ipow(double x, int n)
{
    double v = 1.0;
    for (int e = 0; e < n; ++e) v *= x;
    return v;
}

// 1: the quadrature points a stage starts from are loaded under the stage before it.  Measured (A/B of tagged builds in one call,
// config 3): 0.0975 ms with, 0.0955 ms without -- in a stream of launches the lists sit in L2 and the registers the early loads
// hold cost more than their latency.  Kept as a switch, off.
#ifndef PA_CUT_PREFETCH
#define PA_CUT_PREFETCH 0
#endif

template <int FD, bool DD = true>
// (a block is one wavefront: wave_sync() orders its LDS traffic without the wait for outstanding loads / stores that
// __syncthreads() adds -- the quadrature lists of the next chunk stay in flight)
__global__ __launch_bounds__(64, DD ? 2 : 4) void cut_local_ops_kernel(CutArgs a)
{
    constexpr int RD = FD + 1, RBS = P2(RD), CBS = RBS, FBS = FD + 1, NF = 4 * FBS, MS = CBS + NF;
    constexpr int NMOM = P2(2 * RD), LD = (RBS + 1) & ~1, NFPT = 4 * FACE_SLOTS, CH = 64;
    // LDS map (doubles).  DD: moments, stiffness (factored in place), gr_rhs and oper hold (hi, lo) pairs.
    constexpr int W = DD ? 2 : 1;
    constexpr int oMOM = 0, oST = (oMOM + W * NMOM + 1) & ~1, oLL = oST + (DD ? 0 : LD * RBS), oGR = oLL + W * LD * RBS, oOP = oGR + W * RBS * MS;
    constexpr int PW = 2 * (2 * RD + 1) + 1;                   // per-point scratch row: w bx^e, by^e, f
    constexpr int ROWW = imax(2 * RBS, PW);
    constexpr int oTPHI = oOP + W * RBS * MS, oTDN = oTPHI + CH * RBS, oTW = oTPHI + CH * ROWW;
    constexpr int oFB = oTW + CH, oMF = oFB + NFPT * FBS, oTR = oMF + 4 * FBS * FBS, oPT = oTR + NF * CBS, oDATA = oPT + NF * CBS, oEND = oDATA + MS * MS;
    // DD tables on the point table [oTPHI, oTPHI + CH ROWW): stage B: the interface moments; stage C:
    // phi and w dn of the NFPT face points, then their face-basis values
    constexpr int oCPH = oTPHI, oCDN = oTPHI + 2 * NFPT * RBS, oCFB = oCDN + 2 * NFPT * RBS;
    static_assert(!DD || oCFB + 2 * NFPT * FBS <= oTPHI + CH * ROWW, "the double-double tables fit the point table");
    __shared__ __attribute__((aligned(16))) double S[oEND];
    const int l = threadIdx.x;

    for (uint32_t cc = blockIdx.x; cc < a.ncut; cc += gridDim.x) {
        const uint32_t cell = a.cut_cells[cc];
        // ---- geometry of the WHOLE cell (cell_basis, normals, measure: bases.hpp:85-91,
        // basic_geom.hpp:349-372, cuthho_square.cpp:344)
        const uint4 idv = *reinterpret_cast<const uint4 *>(a.ptids + 4 * (size_t)cell);
        const uint32_t ids[4] = {idv.x, idv.y, idv.z, idv.w};
        double px[4], py[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double2 p = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)ids[v]);
            px[v] = p.x; py[v] = p.y;
        }
        double barx, bary;
        {
            double rx = 0.0, ry = 0.0, den = 0.0;
#pragma unroll
            for (int i = 2; i < 4; ++i) {
                const double ax = px[i - 1] - px[0], ay = py[i - 1] - py[0], bx = px[i] - px[0], by = py[i] - py[0];
                const double d = (ax * by - ay * bx) / 2.0;
                rx += (ax + bx) * d; ry += (ay + by) * d; den += d;
            }
            barx = px[0] + rx / (den * 3); bary = py[0] + ry / (den * 3);
        }
        double hd = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) hd = fmax(hd, sqrt((px[j] - px[i]) * (px[j] - px[i]) + (py[j] - py[i]) * (py[j] - py[i])));
        const double ihalf = 1.0 / (0.5 * hd), ih = 2.0 / hd;
        double hT = 0.0;
#pragma unroll
        for (int i = 1; i < 3; ++i)
            hT += fabs((px[i] - px[0]) * (py[i + 1] - py[0]) - (py[i] - py[0]) * (px[i + 1] - px[0])) * 0.5;
        const double eta_h = a.eta / hT;

        // basis helpers (scaled monomials, bases.hpp:93-184), runtime index m -> exponents
        auto phi_m = [&](double bx, double by, int m) {
            int p, r; mono_exps(m, p, r);
            return ipow(bx, p) * ipow(by, r);
        };
        auto grad_m = [&](double bx, double by, int m, double &gx, double &gy) {
            int p, r; mono_exps(m, p, r);
            gx = p == 0 ? 0.0 : p * ih * ipow(bx, p - 1) * ipow(by, r);
            gy = r == 0 ? 0.0 : r * ih * ipow(bx, p) * ipow(by, r - 1);
        };

        // ---- A: cell moments over the cut quadrature (cuthho_square.cpp:336-341) and, from the same
        // points, the volume part of the right-hand side (:639-644; degree == recdeg: same list).
        // Chunks of 64 points: one lane per point stages w*bx^e, by^e and f(x) in LDS, then one lane
        // per moment (and one per rhs mode) accumulates.
        const uint32_t c0 = a.cell_off[cc], c1 = a.cell_off[cc + 1];
        constexpr int NPW = 2 * RD + 1;
        double mom_acc = 0.0, rhs_acc = 0.0;
        int mp = 0, mr = 0, rp = 0, rr = 0;
        if (l < NMOM) mono_exps(l, mp, mr);
        if (l < CBS) mono_exps(l, rp, rr);
        if constexpr (!DD) {
            for (uint32_t base = c0; base < c1; base += CH) {
                const uint32_t q = base + l;
                if (q < c1) {
                    const double x = a.cell_xyw[3 * q], y = a.cell_xyw[3 * q + 1], w = a.cell_xyw[3 * q + 2];
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    double vx = w, vy = 1.0;
                    for (int e = 0; e < NPW; ++e) {
                        S[oTPHI + l * ROWW + e] = vx;
                        S[oTPHI + l * ROWW + NPW + e] = vy;
                        vx *= bx; vy *= by;
                    }
                    S[oTPHI + l * ROWW + 2 * NPW] = a.rhs == nullptr ? 0.0 : (a.rhs_fn == FN_SAMPLED ? a.rhs_vals[q] : builtin_fn(a.rhs_fn, x, y));
                }
                wave_sync();
                const int nq = (int)((c1 - base) < (uint32_t)CH ? (c1 - base) : (uint32_t)CH);
                // (unrolled by 8: the LDS reads of eight points are in flight together -- one at a time, each iteration waited for its
                // own round trip and the kernel's time was the sum of those latencies; the sums keep their order)
                // (one loop for the moments and the right-hand-side modes: their reads travel together; lanes without a moment / a mode
                // accumulate into registers nobody reads)
                {
                    const bool hm = l < NMOM, hr_ = l < CBS;
                    const int mpo = hm ? mp : 0, mro = NPW + (hm ? mr : 0), rpo = hr_ ? rp : 0, rro = NPW + (hr_ ? rr : 0);
    #pragma unroll 8
                    for (int t = 0; t < nq; ++t) {
                        const double *row = S + oTPHI + t * ROWW;
                        mom_acc += row[mpo] * row[mro];
                        rhs_acc += (row[rpo] * row[rro]) * row[2 * NPW];
                    }
                }
                wave_sync();
            }
        }
        int bad = 0;
        // loads of stages F and H issued early by the double-double path (see stage E)
        double f_x = 0.0, f_y = 0.0, f_w = 0.0, h_x = 0.0, h_y = 0.0, h_w = 0.0;
        int f_n = 0;
        // Only the stabilization wanted (the interface problem takes make_hho_cut_stabilization of both sides from this kernel,
        // cuthho_square.cpp:1694-1705): stages A-E -- the whole reconstruction -- are skipped.
        const bool only_stab = a.oper == nullptr && a.data == nullptr && a.lc == nullptr && a.rhs == nullptr;
        if (!only_stab) {
        if constexpr (DD) {
        // =========== stages A-E in double-double (see the head of the file) ===========
        static_assert(RBS * (RBS + 1) / 2 <= 64 && MS <= 64 && NFPT <= 64, "one lane per pair / column / point");
        // exponents of the monomials as compile-time tables (graded ordering, bases.hpp:114-128)
        // ---- A: moments  sum_q w_q bx^p by^r  over the cut cell's quadrature (cuthho_square.cpp:336-341): one lane per POINT (the
        // points of the list dealt out round robin), 28 double-double accumulators per lane, then a butterfly over the lanes
        // (the volume part of the right-hand side, :639-644, rides along in double: per point f(x) w phi_m, m < cbs)
        PA_CUT_TICK(0);
        {
            dd macc[NMOM];
            double racc[CBS];
#pragma unroll
            for (int m = 0; m < NMOM; ++m) macc[m] = dd_from(0.0);
#pragma unroll
            for (int m = 0; m < CBS; ++m) racc[m] = 0.0;
            // (the next point of a lane is loaded while the current one is worked on: a wavefront alone on its SIMD has nothing else
            // to cover a load's round trip to L2 / HBM with, and every stage of this kernel used to start with one)
            double nxt_x = 0.0, nxt_y = 0.0, nxt_w = 0.0;
            if (PA_CUT_PREFETCH && c0 + l < c1) { nxt_x = a.cell_xyw[3 * (c0 + l)]; nxt_y = a.cell_xyw[3 * (c0 + l) + 1]; nxt_w = a.cell_xyw[3 * (c0 + l) + 2]; }
            for (uint32_t q = c0 + l; q < c1; q += 64) {
                const double x = PA_CUT_PREFETCH ? nxt_x : a.cell_xyw[3 * q], y = PA_CUT_PREFETCH ? nxt_y : a.cell_xyw[3 * q + 1];
                const double w = PA_CUT_PREFETCH ? nxt_w : a.cell_xyw[3 * q + 2];
                if (PA_CUT_PREFETCH && q + 64 < c1) { nxt_x = a.cell_xyw[3 * (q + 64)]; nxt_y = a.cell_xyw[3 * (q + 64) + 1]; nxt_w = a.cell_xyw[3 * (q + 64) + 2]; }
                const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                dd pbx[NPW], pby[NPW];                  // w bx^e, by^e
                pbx[0] = dd_from(w); pby[0] = dd_from(1.0);
#pragma unroll
                for (int e = 1; e < NPW; ++e) { pbx[e] = dd_mul_d(pbx[e - 1], bx); pby[e] = dd_mul_d(pby[e - 1], by); }
#pragma unroll
                for (int k = 0; k < NPW; ++k)
#pragma unroll
                    for (int r = 0; r <= k; ++r)
                        macc[k * (k + 1) / 2 + r] = dd_add_fast(macc[k * (k + 1) / 2 + r], dd_mul(pbx[k - r], pby[r]));
                if (a.rhs != nullptr) {
                    const double fv = a.rhs_fn == FN_SAMPLED ? a.rhs_vals[q] : builtin_fn(a.rhs_fn, x, y);
#pragma unroll
                    for (int k = 0; k <= RD; ++k)
#pragma unroll
                        for (int r = 0; r <= k; ++r) racc[k * (k + 1) / 2 + r] += fv * (pbx[k - r].hi * pby[r].hi);
                }
            }
            // sums over the lanes: a butterfly that halves the values a lane carries at every level (dd_arith.hpp): 29 exchanges
            // for the 28 moments where the plain butterfly makes 168
            {
                dd tot; bool ok = true;
                const int m = lanes_transpose_reduce<NMOM, 32>(macc, l, dd_from(0.0), [](dd x, dd y) { return dd_add_fast(x, y); },
                                                               [](dd x, int off) { return dd_shfl_xor(x, off); }, tot, ok);
                if (ok) dd_store(S + oMOM + 2 * m, tot);      // (the lanes that end with the same entry hold the same sum)
            }
            if (a.rhs != nullptr) {
                double tot; bool ok = true;
                const int m = lanes_transpose_reduce<CBS, 32>(racc, l, 0.0, [](double x, double y) { return x + y; },
                                                              [](double x, int off) { return __shfl_xor(x, off); }, tot, ok);
                // lane m of the wavefront keeps the volume part of mode m for stage H
                S[oTW + (ok ? m : CH - 1)] = tot;       // (scratch: the point weights' row is dead in the double-double path until H)
            }
        }
        wave_sync();
        if (a.rhs != nullptr && l < CBS) rhs_acc = S[oTW + l];
        wave_sync();
        PA_CUT_TICK(1);
        const dd ih2 = two_prod(ih, ih);
        for (int e = l; e < RBS * RBS; e += 64) {             // stiffness from the moments  bases.hpp:170-176
            int ai, bi, aj, bj;
            mono_exps(e % RBS, ai, bi);
            mono_exps(e / RBS, aj, bj);
            dd v = dd_from(0.0);
            if (ai * aj) v = dd_add(v, dd_mul_d(dd_load(S + oMOM + 2 * mono_index(ai + aj - 2, bi + bj)), (double)(ai * aj)));
            if (bi * bj) v = dd_add(v, dd_mul_d(dd_load(S + oMOM + 2 * mono_index(ai + aj, bi + bj - 2)), (double)(bi * bj)));
            dd_store(S + oST + 2 * ((e % RBS) + (e / RBS) * LD), dd_mul(v, ih2));
        }
        wave_sync();
        // the scaled monomials and their gradients at a point, double-double from the double coordinates:
        // one monomial of the same basis with RUN-TIME exponents (a lane's share of a point's monomials): the powers are picked from
        // the registers by selects, never by a dynamic index (which would send the arrays to scratch memory)
        auto pick_dd = [&](const dd (&arr)[RD + 1], int i) {
            dd v = arr[0];
#pragma unroll
            for (int e = 1; e <= RD; ++e) { v.hi = i == e ? arr[e].hi : v.hi; v.lo = i == e ? arr[e].lo : v.lo; }
            return v;
        };
        auto powers_dd = [&](double bx, double by, dd (&pbx)[RD + 1], dd (&pby)[RD + 1]) {
            pbx[0] = dd_from(1.0); pby[0] = dd_from(1.0);
#pragma unroll
            for (int e = 1; e <= RD; ++e) { pbx[e] = dd_mul_d(pbx[e - 1], bx); pby[e] = dd_mul_d(pby[e - 1], by); }
        };
        auto basis_dd_one = [&](const dd (&pbx)[RD + 1], const dd (&pby)[RD + 1], int m, dd &phi, dd &gx, dd &gy) {
            int p_, r_;
            mono_exps(m, p_, r_);
            const dd xp = pick_dd(pbx, p_), yr = pick_dd(pby, r_);
            const dd xp1 = pick_dd(pbx, p_ > 0 ? p_ - 1 : 0), yr1 = pick_dd(pby, r_ > 0 ? r_ - 1 : 0);
            phi = dd_mul(xp, yr);
            gx = dd_mul(dd_mul(xp1, yr), two_prod((double)p_, ih));           // (p = 0: the factor is zero)
            gy = dd_mul(dd_mul(xp, yr1), two_prod((double)r_, ih));
        };
        PA_CUT_TICK(2);
        constexpr int LPC = 64 / NFPT;                     // stage C: lanes per face point (3 with the 20 slots)
        static_assert(LPC >= 1, "a lane per face point");
        const int c_pt = imin(l / LPC, NFPT - 1);
        const double *c_src;
        double c_x, c_y, c_w;
        int c_n;
        // ---- B: Nitsche terms on the interface (cuthho_square.cpp:347-360):  sum_q w (eta/h_T phi_i phi_j - phi_i dn_j - dn_i phi_j).
        // The basis functions are monomials bx^p by^r, so with P = p_i + p_j, R = r_i + r_j the entry is
        //     eta/h_T M(P, R) - ih ( P Mx(P - 1, R) + R My(P, R - 1) ),
        // M, Mx, My the moments of the interface measure w, w n_x, w n_y: 28 + 21 + 21 sums at k = 2 instead of the 55 pairs' two
        // products per point, formed like the cell moments of stage A (a lane per point, the transposed butterfly), three passes over
        // the points so that one pass's accumulators fit the registers.  (Staging phi, w dn and w (eta/h phi - dn) per point and
        // summing the pairs' terms point by point was 41 k of the kernel's 157 k clocks.)
        {
            const uint32_t i0 = a.il_off[cc], i1 = a.il_off[cc + 1];
            constexpr int DM = 2 * RD, NM1 = P2(DM - 1);
            constexpr int oIM = oTPHI, oIX = oIM + 2 * NMOM, oIY = oIX + 2 * NM1;
            static_assert(2 * (NMOM + 2 * NM1) <= CH * ROWW, "the interface moments fit the point table");
            // (... and the face point stage C will want, so that its round trip runs under this stage: the slots of a face without
            // points exist and hold zeros, the loads do not wait for the count)
            auto load_c = [&]() {
                c_src = a.fl_xyw + (((size_t)cc * 4 + c_pt / FACE_SLOTS) * FACE_SLOTS + c_pt % FACE_SLOTS) * 3;
                c_x = c_src[0]; c_y = c_src[1]; c_w = c_src[2];
                c_n = a.fl_cnt[cc * 4 + c_pt / FACE_SLOTS];
            };
            if (PA_CUT_PREFETCH) load_c();
            // (the lane's point of the first 64 -- normally all there are -- and its normal: loaded and formed once for the passes)
            const bool in0 = i0 + l < i1;
            const double x0 = in0 ? a.il_xyw[3 * (i0 + l)] : barx, y0 = in0 ? a.il_xyw[3 * (i0 + l) + 1] : bary;
            const double w0 = in0 ? a.il_xyw[3 * (i0 + l) + 2] : 0.0;
            double nx0, ny0;
            a.ls.normal(x0, y0, nx0, ny0);                                  // not flipped for the positive side (:352-355)
            // one pass: the moments of total degree K0 .. K1 of measure KIND (0: w, 1: w n_x, 2: w n_y)
            auto iface_moments = [&](auto kind_c, auto k0_c, auto k1_c, double *dst) {
                constexpr int KIND = decltype(kind_c)::value, K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
                constexpr int B0 = K0 * (K0 + 1) / 2, NM = P2(K1) - B0;
                dd acc[NM];
                bool first = true;
                for (uint32_t q0 = i0; q0 < i1; q0 += 64) {
                    const uint32_t q = q0 + l;
                    const bool in = q < i1;
                    double x = x0, y = y0, w = w0, nx = nx0, ny = ny0;
                    if (q0 != i0) {
                        x = in ? a.il_xyw[3 * q] : barx; y = in ? a.il_xyw[3 * q + 1] : bary; w = in ? a.il_xyw[3 * q + 2] : 0.0;
                        if (KIND != 0) a.ls.normal(x, y, nx, ny);
                    }
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    dd pbx[K1 + 1], pby[K1 + 1];
                    pbx[0] = KIND == 0 ? dd_from(w) : two_prod(w, KIND == 1 ? nx : ny);
                    pby[0] = dd_from(1.0);
#pragma unroll
                    for (int e = 1; e <= K1; ++e) { pbx[e] = dd_mul_d(pbx[e - 1], bx); pby[e] = dd_mul_d(pby[e - 1], by); }
                    if (first) {                                             // (the usual case is one point per lane: no addition at all)
#pragma unroll
                        for (int k = K0; k <= K1; ++k)
#pragma unroll
                            for (int r = 0; r <= k; ++r) acc[k * (k + 1) / 2 + r - B0] = dd_mul(pbx[k - r], pby[r]);
                        first = false;
                    } else {
#pragma unroll
                        for (int k = K0; k <= K1; ++k)
#pragma unroll
                            for (int r = 0; r <= k; ++r)
                                acc[k * (k + 1) / 2 + r - B0] = dd_add_fast(acc[k * (k + 1) / 2 + r - B0], dd_mul(pbx[k - r], pby[r]));
                    }
                }
                if (first) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) acc[m] = dd_from(0.0);
                }
                dd tot; bool ok = true;
                const int m = lanes_transpose_reduce<NM, 32>(acc, l, dd_from(0.0), [](dd u, dd v_) { return dd_add_fast(u, v_); },
                                                             [](dd u, int off) { return dd_shfl_xor(u, off); }, tot, ok);
                if (ok) dd_store(dst + 2 * (B0 + m), tot);
            };
            {
                using I0 = std::integral_constant<int, 0>;
                constexpr int KS = DM >= 4 ? DM - 2 : 0;              // (k = 2: degrees 0..4 and 5..6 apart: 15 + 13 accumulators, not 28)
                iface_moments(I0(), I0(), std::integral_constant<int, KS>(), S + oIM);
                if constexpr (KS < DM) iface_moments(I0(), std::integral_constant<int, KS + 1>(), std::integral_constant<int, DM>(), S + oIM);
                iface_moments(std::integral_constant<int, 1>(), I0(), std::integral_constant<int, DM - 1>(), S + oIX);
                iface_moments(std::integral_constant<int, 2>(), I0(), std::integral_constant<int, DM - 1>(), S + oIY);
            }
            wave_sync();
            if (l < RBS * (RBS + 1) / 2) {                                              // this lane's pair, i <= j
                int pj_ = 0;
                while ((pj_ + 1) * (pj_ + 2) / 2 <= l && pj_ + 1 < RBS) ++pj_;
                const int pi_ = l - pj_ * (pj_ + 1) / 2;
                int p1, r1, p2, r2;
                mono_exps(pi_, p1, r1);
                mono_exps(pj_, p2, r2);
                const int P = p1 + p2, R = r1 + r2;
                dd nacc = dd_mul_d(dd_load(S + oIM + 2 * mono_index(P, R)), eta_h);
                dd g = dd_from(0.0);
                if (P > 0) g = dd_mul_d(dd_load(S + oIX + 2 * mono_index(P - 1, R)), (double)P);
                if (R > 0) g = dd_add(g, dd_mul_d(dd_load(S + oIY + 2 * mono_index(P, R - 1)), (double)R));
                nacc = dd_sub(nacc, dd_mul_d(g, ih));
                const dd v = dd_add(dd_load(S + oST + 2 * (pi_ + pj_ * LD)), nacc);
                dd_store(S + oST + 2 * (pi_ + pj_ * LD), v);
                dd_store(S + oST + 2 * (pj_ + pi_ * LD), v);
            }
            wave_sync();
        }
        PA_CUT_TICK(3);
        // ---- C: gr_rhs (cuthho_square.cpp:362-383); face points of the `where` part
        {
            const int pt_ = l / LPC, sub = l % LPC;
            if (!PA_CUT_PREFETCH) {
                c_src = a.fl_xyw + (((size_t)cc * 4 + c_pt / FACE_SLOTS) * FACE_SLOTS + c_pt % FACE_SLOTS) * 3;
                c_x = c_src[0]; c_y = c_src[1]; c_w = c_src[2];
                c_n = a.fl_cnt[cc * 4 + c_pt / FACE_SLOTS];
            }
            if (pt_ < NFPT) {
                const int f = pt_ / FACE_SLOTS, qq = pt_ % FACE_SLOTS;
                const bool ok = qq < c_n;
                const double x = ok ? c_x : barx, y = ok ? c_y : bary, w = ok ? c_w : 0.0;
                const int f1 = (f + 1) & 3;
                const double ex = px[f1] - px[f], ey = py[f1] - py[f];
                const double len = sqrt(ex * ex + ey * ey);
                const double nx = ey / len, ny = -ex / len;                      // basic_geom.hpp:361-369
                const bool flip = ids[f] > ids[f1];                              // face basis from the lower-id endpoint (bases.hpp:253-280)
                const double ax = flip ? px[f1] : px[f], ay = flip ? py[f1] : py[f];
                const double bxx = flip ? px[f] : px[f1], byy = flip ? py[f] : py[f1];
                const double fbx = 0.5 * (ax + bxx), fby = 0.5 * (ay + byy);
                const double ep = 4.0 * ((fbx - ax) * (x - fbx) + (fby - ay) * (y - fby)) / (len * len);
                const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                dd pbx[RD + 1], pby[RD + 1];
                powers_dd(bx, by, pbx, pby);
#pragma unroll
                for (int it = 0; it < (RBS + LPC - 1) / LPC; ++it) {
                    const int m = sub + it * LPC;
                    if (m < RBS) {
                        dd phi, gx, gy;
                        basis_dd_one(pbx, pby, m, phi, gx, gy);
                        dd_store(S + oCPH + 2 * (pt_ * RBS + m), phi);
                        dd_store(S + oCDN + 2 * (pt_ * RBS + m), dd_mul_d(dd_add(dd_mul_d(gx, nx), dd_mul_d(gy, ny)), w));
                    }
                }
                if (sub == LPC - 1) {
                    dd pe = dd_from(1.0);
#pragma unroll
                    for (int k = 0; k < FBS; ++k) { dd_store(S + oCFB + 2 * (pt_ * FBS + k), pe); pe = dd_mul_d(pe, ep); }
                }
            }
            wave_sync();
            for (int e = l; e < RBS * MS; e += 64) {
                const int i = e % RBS, j = e / RBS;
                dd sacc;
                if (j < CBS) {
                    // (four running sums over the face points: one sum is a chain of NFPT dependent additions)
                    dd s4[4] = {dd_from(0.0), dd_from(0.0), dd_from(0.0), dd_from(0.0)};
                    static_assert(NFPT % 4 == 0, "four running sums");
#pragma unroll
                    for (int p_ = 0; p_ < NFPT; ++p_)
                        s4[p_ & 3] = dd_add_fast(s4[p_ & 3], dd_mul(dd_load(S + oCDN + 2 * (p_ * RBS + i)), dd_load(S + oCPH + 2 * (p_ * RBS + j))));
                    sacc = dd_sub_fast(dd_load(S + oST + 2 * (i + j * LD)), dd_add_fast(dd_add_fast(s4[0], s4[1]), dd_add_fast(s4[2], s4[3])));
                } else {
                    const int f = (j - CBS) / FBS, k = (j - CBS) % FBS;
                    sacc = dd_from(0.0);
                    for (int qq = 0; qq < FACE_SLOTS; ++qq)
                        sacc = dd_add_fast(sacc, dd_mul(dd_load(S + oCDN + 2 * ((f * FACE_SLOTS + qq) * RBS + i)), dd_load(S + oCFB + 2 * ((f * FACE_SLOTS + qq) * FBS + k))));
                }
                dd_store(S + oGR + 2 * (i + j * RBS), sacc);
            }
            wave_sync();
        }
        PA_CUT_TICK(4);
        // ---- D: oper = llt(gr_lhs).solve(gr_rhs) (cuthho_square.cpp:385) in registers, one lane per COLUMN of [gr_lhs | gr_rhs]
        // (rbs + msize lanes).  Right-looking: at pivot j every lane forms y = v_j / sqrt(d) -- a column of gr_lhs its entry L_kj (the
        // matrix is kept whole and symmetric), a column of gr_rhs its forward-substituted entry -- and takes  v_i -= L_ij y  for the
        // rows below, L_ij read from lane j's registers (v_readlane: no LDS image of the factor, no barrier per pivot).  A pivot's
        // dependent path is rsqrt -> y -> ONE update of the next pivot column; the updates of a step are independent of each other.
        // The left-looking form on an LDS image it replaces had the j-term sum in front of every pivot, and its substitutions kept
        // msize lanes busy with a chain of rbs (rbs - 1) / 2 dependent subtractions each way: 51 k of the kernel's 230 k clocks.
        {
            static_assert(RBS + MS <= 64, "a lane per column of [gr_lhs | gr_rhs]");
            const bool isA = l < RBS, isG = l >= RBS && l < RBS + MS;
            const int gc = isG ? l - RBS : 0;
            dd v[RBS], rsv[RBS];
#pragma unroll
            for (int i = 0; i < RBS; ++i)
                v[i] = isA ? dd_load(S + oST + 2 * (i + l * LD)) : isG ? dd_load(S + oGR + 2 * (i + gc * RBS)) : dd_from(1.0);
#pragma unroll
            for (int j = 0; j < RBS; ++j) {
                const dd piv = dd_readlane(v[j], j);
                if (!(piv.hi > 0.0) && !bad) bad = j + 1;
                const dd rs = dd_rsqrt_1(piv);
                rsv[j] = rs;
                const dd y = dd_mul(v[j], rs);
#pragma unroll
                for (int i = j + 1; i < RBS; ++i) {
                    const dd scaled = dd_mul(v[i], rs);                  // (lane j: L_ij)
                    const dd lij = dd_readlane(scaled, j);
                    const dd upd = dd_sub_fast(v[i], dd_mul(lij, y));
                    v[i].hi = l == j ? scaled.hi : l > j ? upd.hi : v[i].hi;
                    v[i].lo = l == j ? scaled.lo : l > j ? upd.lo : v[i].lo;
                }
                v[j].hi = l >= j ? y.hi : v[j].hi;
                v[j].lo = l >= j ? y.lo : v[j].lo;
            }
            PA_CUT_TICK(5);
            // backward substitution, column-oriented as well: x_i = v_i / L_ii, then v_k -= L_ik x_i for the rows above
#pragma unroll
            for (int i = RBS - 1; i >= 0; --i) {
                const dd x = dd_mul(v[i], rsv[i]);
                if (isG) v[i] = x;
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    const dd lik = dd_readlane(v[i], k);                 // entry i of column k of L (lanes < rbs are not touched here)
                    const dd upd = dd_sub_fast(v[k], dd_mul(lik, x));
                    if (isG) v[k] = upd;
                }
            }
            if (isG) {
#pragma unroll
                for (int k = 0; k < RBS; ++k) dd_store(S + oOP + 2 * (k + gc * RBS), v[k]);
            }
            wave_sync();
            if (a.oper != nullptr)
                for (int e = l; e < RBS * MS; e += 64) a.oper[(size_t)cc * (RBS * MS) + e] = dd_round(dd_load(S + oOP + 2 * e));
        }
        PA_CUT_TICK(6);
        // (the face point of stage F and the interface point of stage H: their round trips run under stage E)
        if (PA_CUT_PREFETCH) {
            const int fp = imin(l, NFPT - 1);
            const double *src = a.fs_xyw + (((size_t)cc * 4 + fp / FACE_SLOTS) * FACE_SLOTS + fp % FACE_SLOTS) * 3;
            f_x = src[0]; f_y = src[1]; f_w = src[2];
            f_n = a.fs_cnt[cc * 4 + fp / FACE_SLOTS];
            if (a.rhs != nullptr) {
                const uint32_t r0 = a.ir_off[cc], r1 = a.ir_off[cc + 1];
                if (r0 + l < r1) { h_x = a.ir_xyw[3 * (r0 + l)]; h_y = a.ir_xyw[3 * (r0 + l) + 1]; h_w = a.ir_xyw[3 * (r0 + l) + 2]; }
            }
        }
        // ---- E: data = gr_rhs^T oper (cuthho_square.cpp:386) = G^T A^-1 G, symmetric: the entries i <= j, two running sums each,
        // rounded once, mirrored into the LDS image
#pragma unroll 1
        for (int e = l; e < MS * (MS + 1) / 2; e += 64) {
            int j = (int)((__fsqrt_rn((float)(8 * e + 1)) - 1.0f) * 0.5f);      // column of packed entry e: j (j + 1) / 2 <= e
            j += (j + 1) * (j + 2) / 2 <= e ? 1 : 0;
            j -= j * (j + 1) / 2 > e ? 1 : 0;
            const int i = e - j * (j + 1) / 2;
            dd s0 = dd_from(0.0), s1 = dd_from(0.0);
#pragma unroll
            for (int k = 0; k < RBS; ++k) {
                const dd t = dd_mul(dd_load(S + oGR + 2 * (k + i * RBS)), dd_load(S + oOP + 2 * (k + j * RBS)));
                if (k & 1) s1 = dd_add_fast(s1, t); else s0 = dd_add_fast(s0, t);
            }
            const double r = dd_round(dd_add_fast(s0, s1));
            S[oDATA + i + j * MS] = r;
            S[oDATA + j + i * MS] = r;
        }
        wave_sync();
        } else {
            static_assert(NMOM <= 64, "one lane per moment");
            if (l < NMOM) S[oMOM + l] = mom_acc;
            wave_sync();
            for (int e = l; e < RBS * RBS; e += 64) {
                int ai, bi, aj, bj;
                mono_exps(e % RBS, ai, bi);
                mono_exps(e / RBS, aj, bj);
                double v = 0.0;
                if (ai * aj) v += (double)(ai * aj) * S[oMOM + mono_index(ai + aj - 2, bi + bj)];
                if (bi * bj) v += (double)(bi * bj) * S[oMOM + mono_index(ai + aj, bi + bj - 2)];
                S[oST + (e % RBS) + (e / RBS) * LD] = ih * ih * v;
            }
            wave_sync();

            // ---- B: Nitsche terms on the interface (cuthho_square.cpp:347-360), chunks of 64 points
            const uint32_t i0 = a.il_off[cc], i1 = a.il_off[cc + 1];
            for (uint32_t base = i0; base < i1; base += CH) {
                const uint32_t q = base + l;
                if (q < i1) {
                    const double x = a.il_xyw[3 * q], y = a.il_xyw[3 * q + 1];
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    double nx, ny;
                    a.ls.normal(x, y, nx, ny);                    // not flipped for the positive side (:352-355)
    #pragma unroll
                    for (int m = 0; m < RBS; ++m) {      // (unrolled: the exponents of monomial m are constants, not a search per point)
                        double gx, gy;
                        grad_m(bx, by, m, gx, gy);
                        S[oTPHI + l * RBS + m] = phi_m(bx, by, m);
                        S[oTDN + l * RBS + m] = gx * nx + gy * ny;
                    }
                    S[oTW + l] = a.il_xyw[3 * q + 2];
                } else S[oTW + l] = 0.0;
                wave_sync();
                const int nq = (int)((i1 - base) < (uint32_t)CH ? (i1 - base) : (uint32_t)CH);
                for (int e = l; e < RBS * RBS; e += 64) {
                    const int i = e % RBS, j = e / RBS;
                    double s = 0.0;
    #pragma unroll 4
                    for (int t = 0; t < nq; ++t) {
                        const double w = S[oTW + t], pi_ = S[oTPHI + t * RBS + i], pj = S[oTPHI + t * RBS + j];
                        const double di_ = S[oTDN + t * RBS + i], dj = S[oTDN + t * RBS + j];
                        s += w * (-pi_ * dj - di_ * pj + eta_h * pi_ * pj);
                    }
                    S[oST + i + j * LD] += s;
                }
                wave_sync();
            }

            // ---- C: gr_lhs = stiff, gr_rhs (cuthho_square.cpp:362-383); face points of the `where` part
            {
                if (l < NFPT) {
                    const int f = l / FACE_SLOTS, qq = l % FACE_SLOTS;
                    const bool ok = qq < a.fl_cnt[cc * 4 + f];
                    const double *src = a.fl_xyw + (((size_t)cc * 4 + f) * FACE_SLOTS + qq) * 3;
                    const double x = ok ? src[0] : barx, y = ok ? src[1] : bary, w = ok ? src[2] : 0.0;
                    const int f1 = (f + 1) & 3;
                    const double ex = px[f1] - px[f], ey = py[f1] - py[f];          // dynamic index: 4 entries, fine here
                    const double len = sqrt(ex * ex + ey * ey);
                    const double nx = ey / len, ny = -ex / len;                      // basic_geom.hpp:361-369
                    // face basis of the WHOLE face from its lower-id endpoint (bases.hpp:253-280)
                    const bool flip = ids[f] > ids[f1];
                    const double ax = flip ? px[f1] : px[f], ay = flip ? py[f1] : py[f];
                    const double bxx = flip ? px[f] : px[f1], byy = flip ? py[f] : py[f1];
                    const double fbx = 0.5 * (ax + bxx), fby = 0.5 * (ay + byy);
                    const double ep = 4.0 * ((fbx - ax) * (x - fbx) + (fby - ay) * (y - fby)) / (len * len);
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
    #pragma unroll
                    for (int m = 0; m < RBS; ++m) {      // (unrolled: the exponents of monomial m are constants, not a search per point)
                        double gx, gy;
                        grad_m(bx, by, m, gx, gy);
                        S[oTPHI + l * RBS + m] = phi_m(bx, by, m);
                        S[oTDN + l * RBS + m] = w * (gx * nx + gy * ny);
                    }
                    for (int k = 0; k < FBS; ++k) S[oFB + l * FBS + k] = ipow(ep, k);
                }
                wave_sync();
                for (int e = l; e < RBS * MS; e += 64) {
                    const int i = e % RBS, j = e / RBS;
                    double s;
                    if (j < CBS) {
                        s = S[oST + i + j * LD];
                        for (int p = 0; p < NFPT; ++p) s -= S[oTDN + p * RBS + i] * S[oTPHI + p * RBS + j];
                    } else {
                        const int f = (j - CBS) / FBS, k = (j - CBS) % FBS;
                        s = 0.0;
                        for (int qq = 0; qq < FACE_SLOTS; ++qq) s += S[oTDN + (f * FACE_SLOTS + qq) * RBS + i] * S[oFB + (f * FACE_SLOTS + qq) * FBS + k];
                    }
                    S[oGR + i + j * RBS] = s;
                }
                for (int e = l; e < RBS * LD; e += 64) S[oLL + e] = S[oST + e];
                wave_sync();
            }

            // ---- D: oper = llt(gr_lhs).solve(gr_rhs) (cuthho_square.cpp:385), full rbs x rbs system
            bad = lds_cholesky<RBS, LD, 64, 2>(S + oLL, l);     // sliver cuts are badly conditioned: full-accuracy pivots
            {
                double x[RBS];
                const int c = l < MS ? l : 0;
    #pragma unroll
                for (int k = 0; k < RBS; ++k) x[k] = S[oGR + k + c * RBS];
                lds_forward<RBS, LD>(S + oLL, x);
                lds_backward<RBS, LD>(S + oLL, x);
                if (l < MS) {
    #pragma unroll
                    for (int k = 0; k < RBS; ++k) S[oOP + k + c * RBS] = x[k];
                }
            }
            wave_sync();
            if (a.oper != nullptr)
                for (int e = l; e < RBS * MS; e += 64) a.oper[(size_t)cc * (RBS * MS) + e] = S[oOP + e];

            // ---- E: data = gr_rhs^T oper (cuthho_square.cpp:386), kept in an LDS image
    #pragma unroll 1
            for (int e = l; e < MS * MS; e += 64) {
                const int i = e % MS, j = e / MS;
                double s = 0.0;
    #pragma unroll
                for (int k = 0; k < RBS; ++k) s += S[oGR + k + i * RBS] * S[oOP + k + j * RBS];
                S[oDATA + e] = s;
            }
            wave_sync();

        }
        }   // !only_stab
        PA_CUT_TICK(7);
        // ---- F: cut stabilization (cuthho_square.cpp:566-621): faces without points are skipped
        {
            if (l < NFPT) {
                const int f = l / FACE_SLOTS, qq = l % FACE_SLOTS;
                if (!DD || !PA_CUT_PREFETCH || only_stab) {
                    const double *src = a.fs_xyw + (((size_t)cc * 4 + f) * FACE_SLOTS + qq) * 3;
                    f_x = src[0]; f_y = src[1]; f_w = src[2];
                    f_n = a.fs_cnt[cc * 4 + f];
                }
                const bool ok = qq < f_n;
                const double x = ok ? f_x : barx, y = ok ? f_y : bary, w = ok ? f_w : 0.0;
                const int f1 = (f + 1) & 3;
                const double ex = px[f1] - px[f], ey = py[f1] - py[f];
                const double len2 = ex * ex + ey * ey;
                const bool flip = ids[f] > ids[f1];
                const double ax = flip ? px[f1] : px[f], ay = flip ? py[f1] : py[f];
                const double bxx = flip ? px[f] : px[f1], byy = flip ? py[f] : py[f1];
                const double fbx = 0.5 * (ax + bxx), fby = 0.5 * (ay + byy);
                const double ep = 4.0 * ((fbx - ax) * (x - fbx) + (fby - ay) * (y - fby)) / len2;
                const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                for (int m = 0; m < CBS; ++m) S[oTPHI + l * RBS + m] = phi_m(bx, by, m);
                for (int k = 0; k < FBS; ++k) S[oFB + l * FBS + k] = ipow(ep, k);
                S[oTW + l] = w;
            }
            wave_sync();
            for (int e = l; e < 4 * FBS * FBS; e += 64) {              // mass_F  (:608)
                const int f = e / (FBS * FBS), k = e % FBS, k2 = (e / FBS) % FBS;
                double s = 0.0;
                for (int qq = 0; qq < FACE_SLOTS; ++qq) {
                    const int p = f * FACE_SLOTS + qq;
                    s += (S[oTW + p] * S[oFB + p * FBS + k]) * S[oFB + p * FBS + k2];
                }
                S[oMF + f * FBS * FBS + k + k2 * FBS] = s;
            }
            for (int e = l; e < NF * CBS; e += 64) {                   // trace_F (:609)
                const int fk = e % NF, i = e / NF, f = fk / FBS, k = fk % FBS;
                double s = 0.0;
                for (int qq = 0; qq < FACE_SLOTS; ++qq) {
                    const int p = f * FACE_SLOTS + qq;
                    s += (S[oTW + p] * S[oFB + p * FBS + k]) * S[oTPHI + p * RBS + i];
                }
                S[oTR + fk + i * NF] = s;
            }
            wave_sync();
            static_assert(4 * CBS <= 64, "a lane per (face, column)");
            if (l < 4 * CBS) {                                         // mass.llt().solve(trace) (:615): one lane per (face, column)
                const int f = l / CBS, c = l % CBS;                    // (the four faces one after the other on cbs lanes: a chain of
                if (a.fs_cnt[cc * 4 + f] != 0) {                       //  4 x 12 square roots and divisions)
                    double Lf[FBS][FBS], ri[FBS], x[FBS];
#pragma unroll
                    for (int j = 0; j < FBS; ++j) {                    // tiny Cholesky, redundantly per lane; 1 / L_jj kept
                        double d = S[oMF + f * FBS * FBS + j + j * FBS];
#pragma unroll
                        for (int k = 0; k < j; ++k) d -= Lf[j][k] * Lf[j][k];
                        ri[j] = fast_rsqrt<2>(d);
                        Lf[j][j] = d * ri[j];
#pragma unroll
                        for (int i = j + 1; i < FBS; ++i) {
                            double s_ = S[oMF + f * FBS * FBS + i + j * FBS];
#pragma unroll
                            for (int k = 0; k < j; ++k) s_ -= Lf[i][k] * Lf[j][k];
                            Lf[i][j] = s_ * ri[j];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < FBS; ++i) {
                        double s_ = S[oTR + (f * FBS + i) + c * NF];
#pragma unroll
                        for (int k = 0; k < i; ++k) s_ -= Lf[i][k] * x[k];
                        x[i] = s_ * ri[i];
                    }
#pragma unroll
                    for (int i = FBS - 1; i >= 0; --i) {
                        double s_ = x[i];
#pragma unroll
                        for (int k = i + 1; k < FBS; ++k) s_ -= Lf[k][i] * x[k];
                        x[i] = s_ * ri[i];
                    }
#pragma unroll
                    for (int i = 0; i < FBS; ++i) S[oPT + (f * FBS + i) + c * NF] = x[i];
                }
            }
            wave_sync();
            // data += oper_F^T mass_F oper_F / hT, oper_F = [ M^-1 trace | -I_F ]  (:599,:615-617); outputs.
            // Two tables first, on the dead face-point table: OPF[(f,k)][i] = oper_F (zero rows for a face without points) and
            // WF[(f,k)][j] = (mass_F oper_F)[k][j]; then every entry is the 4 FBS-term sum  OPF[:, i] . WF[:, j]  over compile-time
            // rows, its reads in flight together -- the same products in the same order as the nested loops over faces and modes
            // they replace (which skipped the zero factors one data-dependent branch at a time: 28 of the kernel's 114 us).
            double *OPF = S + oTPHI, *WF = S + oTPHI + NF * MS;
            static_assert(2 * NF * MS <= CH * ROWW, "the two tables fit the point table");
            for (int e = l; e < NF * MS; e += 64) {
                const int fk = e % NF, i = e / NF, f = fk / FBS, k = fk % FBS;
                const bool on = a.fs_cnt[cc * 4 + f] != 0;
                OPF[fk + i * NF] = !on ? 0.0 : (i < CBS ? S[oPT + fk + i * NF] : (i == CBS + fk ? -1.0 : 0.0));
            }
            wave_sync();
            for (int e = l; e < NF * MS; e += 64) {
                const int fk = e % NF, jc = e / NF, f = fk / FBS, k = fk % FBS;
                double mo = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < FBS; ++k2) mo += S[oMF + f * FBS * FBS + k + k2 * FBS] * OPF[(f * FBS + k2) + jc * NF];
                WF[fk + jc * NF] = mo;
            }
            wave_sync();
            const size_t off = (size_t)cc * (MS * MS);
            // (the stabilization is symmetric like data: the entries i <= jc, two running sums each, both copies stored)
#pragma unroll 1
            for (int e = l; e < MS * (MS + 1) / 2; e += 64) {
                int jc = (int)((__fsqrt_rn((float)(8 * e + 1)) - 1.0f) * 0.5f);
                jc += (jc + 1) * (jc + 2) / 2 <= e ? 1 : 0;
                jc -= jc * (jc + 1) / 2 > e ? 1 : 0;
                const int i = e - jc * (jc + 1) / 2;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int fk = 0; fk < NF; fk += 2) {
                    s0 += OPF[fk + i * NF] * WF[fk + jc * NF];
                    s1 += OPF[fk + 1 + i * NF] * WF[fk + 1 + jc * NF];
                }
                const double st = (s0 + s1) * (1.0 / hT), dt = S[oDATA + i + jc * MS];
                const int e1 = i + jc * MS, e2 = jc + i * MS;
                if (a.lc != nullptr) { a.lc[off + e1] = dt + st; a.lc[off + e2] = dt + st; }
                if (a.data != nullptr) { a.data[off + e1] = dt; a.data[off + e2] = dt; }
                if (a.stab != nullptr) { a.stab[off + e1] = st; a.stab[off + e2] = st; }
            }
            if (a.info != nullptr && l == 0) a.info[cc] = bad;
        }

        PA_CUT_TICK(8);
        // ---- H: right-hand side (cuthho_square.cpp:630-657).  The source / boundary functions are
        // evaluated once per quadrature point (one lane each, chunks of 64), then lane i sums its mode.
        if (a.rhs != nullptr) {
            double s = rhs_acc;                                         // volume part, accumulated in stage A
            const uint32_t r0 = a.ir_off[cc], r1 = a.ir_off[cc + 1];      // integrate_interface(msh, cl, degree, where)  (:647)
            for (uint32_t base = r0; base < r1; base += CH) {
                const uint32_t q = base + l;
                // The point's lane stages the POWERS bx^e, by^e (the products ipow forms, in its order) next to the weighted boundary
                // value and the normal; the mode's lane then reads the four powers its monomial and its gradient need.  (With
                // phi_m / grad_m called per point and mode -- exponent search and power loops inside a serial loop over the points --
                // this stage was 45 of the kernel's 114 us.)
                constexpr int HROW = 2 * (RD + 1) + 3;
                static_assert(HROW <= ROWW, "stage H rows fit the point table of stage A");
                if (q < r1) {
                    const bool pre = DD && PA_CUT_PREFETCH && base == r0;   // (loaded under stage E)
                    const double x = pre ? h_x : a.ir_xyw[3 * q], y = pre ? h_y : a.ir_xyw[3 * q + 1];
                    const double wq = pre ? h_w : a.ir_xyw[3 * q + 2];
                    double nx, ny;
                    a.ls.normal(x, y, nx, ny);
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    double vx = 1.0, vy = 1.0;
#pragma unroll
                    for (int e = 0; e <= RD; ++e) {
                        S[oTPHI + HROW * l + e] = vx;
                        S[oTPHI + HROW * l + (RD + 1) + e] = vy;
                        vx *= bx; vy *= by;
                    }
                    S[oTPHI + HROW * l + 2 * (RD + 1)] = wq * (a.bcs_fn == FN_SAMPLED ? a.bcs_vals[q] : builtin_fn(a.bcs_fn, x, y));
                    S[oTPHI + HROW * l + 2 * (RD + 1) + 1] = nx;
                    S[oTPHI + HROW * l + 2 * (RD + 1) + 2] = ny;
                }
                wave_sync();
                const int nq = (int)((r1 - base) < (uint32_t)CH ? (r1 - base) : (uint32_t)CH);
                if (l < CBS) {
                    int hp, hr;
                    mono_exps(l, hp, hr);
                    const int hp1 = hp > 0 ? hp - 1 : 0, hr1 = hr > 0 ? hr - 1 : 0;
                    const double cpx = hp == 0 ? 0.0 : hp * ih, cpy = hr == 0 ? 0.0 : hr * ih;      // (p ih, r ih: the leading factors of grad_m)
#pragma unroll 4
                    for (int t = 0; t < nq; ++t) {
                        const double *row = S + oTPHI + HROW * t;
                        const double xp = row[hp], yr = row[(RD + 1) + hr], xp1 = row[hp1], yr1 = row[(RD + 1) + hr1];
                        const double gx = hp == 0 ? 0.0 : cpx * xp1 * yr, gy = hr == 0 ? 0.0 : cpy * xp * yr1;
                        s += row[2 * (RD + 1)] * ((xp * yr) * eta_h - (gx * row[2 * (RD + 1) + 1] + gy * row[2 * (RD + 1) + 2]));
                    }
                }
                wave_sync();
            }
            if (l < CBS) a.rhs[(size_t)cc * CBS + l] = s;
        }
        wave_sync();
        PA_CUT_TICK(9);
    }
}

// fictitious-domain merge: the cut cells' operators replace the rows of the cell-major arrays and
// the right-hand side of cells outside `where` is zeroed (cuthho_square.cpp:628-629, 659-664).
// Two launches: one block per CUT cell (the list of cut cells, ~0.5 % of the mesh) copies its matrix and right-hand side;
// one thread per right-hand-side entry of the whole mesh zeroes those of the cells outside the domain (coalesced stores).
// (One block per cell of the WHOLE mesh, most of them with nothing or 80 bytes to write, took 51 us of config 3's 620.)
__global__ __launch_bounds__(64) void cut_merge_cells_kernel(uint32_t ncut, const uint32_t *cut_cells, int msize2, int cbs,
                                                            const double *cut_lc, const double *cut_rhs, double *lc, double *rhs)
{
    const size_t cc = blockIdx.x;
    if (cc >= ncut) return;
    const size_t c = cut_cells[cc];
    if (lc != nullptr && cut_lc != nullptr) {
        // (msize^2 is even for the even msize of the cut configurations; the odd tail entry by itself)
        const double *src = cut_lc + cc * (size_t)msize2;
        double *dst = lc + c * (size_t)msize2;
        for (int e = threadIdx.x; e < msize2; e += 64) dst[e] = src[e];
    }
    if (rhs != nullptr && cut_rhs != nullptr)
        for (int e = threadIdx.x; e < cbs; e += 64) rhs[c * (size_t)cbs + e] = cut_rhs[cc * (size_t)cbs + e];
}

// condensed mode: the cut cells' packed records [upper triangle of S | g] into the cell-major record array
__global__ __launch_bounds__(64) void cut_merge_condensed_kernel(uint32_t ncut, const uint32_t *cut_cells, int ntri, int nf, const double *cut_Sp,
                                                                const double *cut_g, double *cond)
{
    const size_t cc = blockIdx.x;
    if (cc >= ncut) return;
    double *dst = cond + (size_t)cut_cells[cc] * (size_t)(ntri + nf);
    for (int e = threadIdx.x; e < ntri; e += 64) dst[e] = cut_Sp[cc * (size_t)ntri + e];
    for (int e = threadIdx.x; e < nf; e += 64) dst[ntri + e] = cut_g[cc * (size_t)nf + e];
}

__global__ __launch_bounds__(256) void cut_zero_rhs_kernel(size_t total, uint32_t cbs, const int8_t *cell_loc, int where, double *rhs)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int loc = cell_loc[t / cbs];
    if (loc != LOC_CUT && loc != where) rhs[t] = 0.0;
}

}  // namespace pa
