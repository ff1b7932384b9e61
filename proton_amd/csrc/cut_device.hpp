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
    // DD tables on the point table [oTPHI, oTPHI + CH ROWW): stage B, chunks of CHB points: phi (CHB RBS pairs), then dn; stage C:
    // phi and w dn of the NFPT face points, then their face-basis values
    constexpr int CHB = imin(64, (CH * ROWW) / (6 * RBS));     // three tables of (hi, lo) pairs per point: 21 points at k = 2
    constexpr int oBPH = oTPHI, oBDN = oTPHI + 2 * CHB * RBS, oBG = oBDN + 2 * CHB * RBS;
    constexpr int oCPH = oTPHI, oCDN = oTPHI + 2 * NFPT * RBS, oCFB = oCDN + 2 * NFPT * RBS;
    static_assert(!DD || (6 * CHB * RBS <= CH * ROWW && oCFB + 2 * NFPT * FBS <= oTPHI + CH * ROWW && CHB <= CH && CHB >= 1), "the double-double tables fit the point table");
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
        if constexpr (DD) {
        // =========== stages A-E in double-double (see the head of the file) ===========
        static_assert(RBS * (RBS + 1) / 2 <= 64 && MS <= 64 && NFPT <= 64 && CHB <= 64, "one lane per pair / column / point");
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
            for (uint32_t q = c0 + l; q < c1; q += 64) {
                const double x = a.cell_xyw[3 * q], y = a.cell_xyw[3 * q + 1], w = a.cell_xyw[3 * q + 2];
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
#pragma unroll
            for (int m = 0; m < NMOM; ++m) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const dd o = dd{__shfl_xor(macc[m].hi, off), __shfl_xor(macc[m].lo, off)};
                    macc[m] = dd_add_fast(macc[m], o);
                }
                if (l == m) dd_store(S + oMOM + 2 * m, macc[m]);
            }
            if (a.rhs != nullptr) {
#pragma unroll
                for (int m = 0; m < CBS; ++m) {
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) racc[m] += __shfl_xor(racc[m], off);
                    if (l == m) rhs_acc = racc[m];
                }
            }
        }
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
        // the scaled monomials and their gradients at a point, double-double from the double coordinates
        auto basis_dd = [&](double bx, double by, dd (&phi)[RBS], dd (&gx)[RBS], dd (&gy)[RBS]) {
            dd pbx[RD + 1], pby[RD + 1];
            pbx[0] = dd_from(1.0); pby[0] = dd_from(1.0);
#pragma unroll
            for (int e = 1; e <= RD; ++e) { pbx[e] = dd_mul_d(pbx[e - 1], bx); pby[e] = dd_mul_d(pby[e - 1], by); }
            int m = 0;
#pragma unroll
            for (int kk = 0; kk <= RD; ++kk)
#pragma unroll
                for (int ii = 0; ii <= kk; ++ii, ++m) {
                    const int p_ = kk - ii, r_ = ii;                                    // (px, py) = (k - i, i)
                    phi[m] = dd_mul(pbx[p_], pby[r_]);
                    gx[m] = p_ == 0 ? dd_from(0.0) : dd_mul(dd_mul(pbx[p_ > 0 ? p_ - 1 : 0], pby[r_]), two_prod((double)p_, ih));
                    gy[m] = r_ == 0 ? dd_from(0.0) : dd_mul(dd_mul(pbx[p_], pby[r_ > 0 ? r_ - 1 : 0]), two_prod((double)r_, ih));
                }
        };
        PA_CUT_TICK(2);
        // ---- B: Nitsche terms on the interface (cuthho_square.cpp:347-360): chunks of CHB points staged by their lanes, one lane per
        // unordered pair (i, j) of the symmetric matrix accumulates  w (eta/h_T phi_i phi_j - phi_i dn_j - dn_i phi_j)
        {
            const uint32_t i0 = a.il_off[cc], i1 = a.il_off[cc + 1];
            int pi_ = 0, pj_ = 0;                                                       // this lane's pair, i <= j
            {
                int jj = 0;
                while ((jj + 1) * (jj + 2) / 2 <= l && jj + 1 < RBS) ++jj;
                pj_ = jj; pi_ = l - jj * (jj + 1) / 2;
            }
            const bool has_pair = l < RBS * (RBS + 1) / 2;
            dd nacc = dd_from(0.0);
            for (uint32_t base = i0; base < i1; base += CHB) {
                const uint32_t q = base + l;
                if (l < CHB && q < i1) {
                    const double x = a.il_xyw[3 * q], y = a.il_xyw[3 * q + 1];
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    double nx, ny;
                    a.ls.normal(x, y, nx, ny);                    // not flipped for the positive side (:352-355)
                    dd phi[RBS], gx[RBS], gy[RBS];
                    basis_dd(bx, by, phi, gx, gy);
                    const double w = a.il_xyw[3 * q + 2];
                    // staged per point: b_m = w dn_m and g_m = w (eta/h_T phi_m - dn_m): the pair's term is  phi_i g_j - b_i phi_j
#pragma unroll
                    for (int m = 0; m < RBS; ++m) {
                        const dd bm = dd_mul_d(dd_add(dd_mul_d(gx[m], nx), dd_mul_d(gy[m], ny)), w);
                        dd_store(S + oBPH + 2 * (l * RBS + m), phi[m]);
                        dd_store(S + oBDN + 2 * (l * RBS + m), bm);
                        dd_store(S + oBG + 2 * (l * RBS + m), dd_sub(dd_mul_d(dd_mul_d(phi[m], eta_h), w), bm));
                    }
                }
                wave_sync();
                const int nq = (int)((i1 - base) < (uint32_t)CHB ? (i1 - base) : (uint32_t)CHB);
                if (has_pair) {
                    for (int t = 0; t < nq; ++t) {
                        const dd fi = dd_load(S + oBPH + 2 * (t * RBS + pi_)), fj = dd_load(S + oBPH + 2 * (t * RBS + pj_));
                        const dd bi = dd_load(S + oBDN + 2 * (t * RBS + pi_)), gj = dd_load(S + oBG + 2 * (t * RBS + pj_));
                        nacc = dd_add_fast(nacc, dd_sub_fast(dd_mul(fi, gj), dd_mul(bi, fj)));
                    }
                }
                wave_sync();
            }
            if (has_pair) {
                const dd v = dd_add(dd_load(S + oST + 2 * (pi_ + pj_ * LD)), nacc);
                dd_store(S + oST + 2 * (pi_ + pj_ * LD), v);
                dd_store(S + oST + 2 * (pj_ + pi_ * LD), v);
            }
            wave_sync();
        }
        PA_CUT_TICK(3);
        // ---- C: gr_rhs (cuthho_square.cpp:362-383); face points of the `where` part
        {
            if (l < NFPT) {
                const int f = l / FACE_SLOTS, qq = l % FACE_SLOTS;
                const bool ok = qq < a.fl_cnt[cc * 4 + f];
                const double *src = a.fl_xyw + (((size_t)cc * 4 + f) * FACE_SLOTS + qq) * 3;
                const double x = ok ? src[0] : barx, y = ok ? src[1] : bary, w = ok ? src[2] : 0.0;
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
                dd phi[RBS], gx[RBS], gy[RBS];
                basis_dd(bx, by, phi, gx, gy);
#pragma unroll
                for (int m = 0; m < RBS; ++m) {
                    dd_store(S + oCPH + 2 * (l * RBS + m), phi[m]);
                    dd_store(S + oCDN + 2 * (l * RBS + m), dd_mul_d(dd_add(dd_mul_d(gx[m], nx), dd_mul_d(gy[m], ny)), w));
                }
                dd pe = dd_from(1.0);
#pragma unroll
                for (int k = 0; k < FBS; ++k) { dd_store(S + oCFB + 2 * (l * FBS + k), pe); pe = dd_mul_d(pe, ep); }
            }
            wave_sync();
            for (int e = l; e < RBS * MS; e += 64) {
                const int i = e % RBS, j = e / RBS;
                dd sacc;
                if (j < CBS) {
                    sacc = dd_load(S + oST + 2 * (i + j * LD));
                    for (int p_ = 0; p_ < NFPT; ++p_)
                        sacc = dd_sub_fast(sacc, dd_mul(dd_load(S + oCDN + 2 * (p_ * RBS + i)), dd_load(S + oCPH + 2 * (p_ * RBS + j))));
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
        // ---- D: llt(gr_lhs) in place on the stiffness image (cuthho_square.cpp:385), lane i = row i; 1 / L_jj kept in the
        // (dead) moment table
        {
            double *RS = S + oMOM;
#pragma unroll 1
            for (int j = 0; j < RBS; ++j) {
                dd sj = dd_from(0.0);
                if (l >= j && l < RBS) {
                    sj = dd_load(S + oST + 2 * (l + j * LD));
                    for (int k = 0; k < j; ++k)
                        sj = dd_sub(sj, dd_mul(dd_load(S + oST + 2 * (l + k * LD)), dd_load(S + oST + 2 * (j + k * LD))));
                    if (l == j) dd_store(S + oST + 2 * (j + j * LD), sj);                 // the pivot, unscaled, for the other rows
                }
                wave_sync();
                const dd piv = dd_load(S + oST + 2 * (j + j * LD));
                if (!(piv.hi > 0.0) && !bad) bad = j + 1;
                const dd rs = dd_rsqrt(piv);
                wave_sync();
                if (l >= j && l < RBS) dd_store(S + oST + 2 * (l + j * LD), dd_mul(sj, rs));   // L_ij = s_i / sqrt(d); L_jj = sqrt(d)
                if (l == 0) dd_store(RS + 2 * j, rs);
                wave_sync();
            }
            PA_CUT_TICK(5);
            // oper = L^-T L^-1 gr_rhs, column c = lane
            if (l < MS) {
                dd xv[RBS];
#pragma unroll
                for (int k = 0; k < RBS; ++k) xv[k] = dd_load(S + oGR + 2 * (k + l * RBS));
#pragma unroll
                for (int i = 0; i < RBS; ++i) {
                    dd sv = xv[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) sv = dd_sub_fast(sv, dd_mul(dd_load(S + oST + 2 * (i + k * LD)), xv[k]));
                    xv[i] = dd_mul(sv, dd_load(RS + 2 * i));
                }
#pragma unroll
                for (int i = RBS - 1; i >= 0; --i) {
                    dd sv = xv[i];
#pragma unroll
                    for (int k = i + 1; k < RBS; ++k) sv = dd_sub_fast(sv, dd_mul(dd_load(S + oST + 2 * (k + i * LD)), xv[k]));
                    xv[i] = dd_mul(sv, dd_load(RS + 2 * i));
                }
#pragma unroll
                for (int k = 0; k < RBS; ++k) dd_store(S + oOP + 2 * (k + l * RBS), xv[k]);
            }
            wave_sync();
            if (a.oper != nullptr)
                for (int e = l; e < RBS * MS; e += 64) a.oper[(size_t)cc * (RBS * MS) + e] = dd_round(dd_load(S + oOP + 2 * e));
        }
        PA_CUT_TICK(6);
        // ---- E: data = gr_rhs^T oper (cuthho_square.cpp:386), rounded once, kept in the LDS image
#pragma unroll 1
        for (int e = l; e < MS * MS; e += 64) {
            const int i = e % MS, j = e / MS;
            dd sacc = dd_from(0.0);
#pragma unroll
            for (int k = 0; k < RBS; ++k) sacc = dd_add_fast(sacc, dd_mul(dd_load(S + oGR + 2 * (k + i * RBS)), dd_load(S + oOP + 2 * (k + j * RBS))));
            S[oDATA + e] = dd_round(sacc);
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
        PA_CUT_TICK(7);
        // ---- F: cut stabilization (cuthho_square.cpp:566-621): faces without points are skipped
        {
            if (l < NFPT) {
                const int f = l / FACE_SLOTS, qq = l % FACE_SLOTS;
                const bool ok = qq < a.fs_cnt[cc * 4 + f];
                const double *src = a.fs_xyw + (((size_t)cc * 4 + f) * FACE_SLOTS + qq) * 3;
                const double x = ok ? src[0] : barx, y = ok ? src[1] : bary, w = ok ? src[2] : 0.0;
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
            if (l < CBS) {                                             // mass.llt().solve(trace), column l (:615)
#pragma unroll 1
                for (int f = 0; f < 4; ++f) {
                    if (a.fs_cnt[cc * 4 + f] == 0) continue;
                    double Lf[FBS][FBS], x[FBS];
                    for (int j = 0; j < FBS; ++j) {                    // tiny Cholesky, redundantly per lane
                        double d = S[oMF + f * FBS * FBS + j + j * FBS];
                        for (int k = 0; k < j; ++k) d -= Lf[j][k] * Lf[j][k];
                        Lf[j][j] = sqrt(d);
                        for (int i = j + 1; i < FBS; ++i) {
                            double s = S[oMF + f * FBS * FBS + i + j * FBS];
                            for (int k = 0; k < j; ++k) s -= Lf[i][k] * Lf[j][k];
                            Lf[i][j] = s / Lf[j][j];
                        }
                    }
                    for (int i = 0; i < FBS; ++i) {
                        double s = S[oTR + (f * FBS + i) + l * NF];
                        for (int k = 0; k < i; ++k) s -= Lf[i][k] * x[k];
                        x[i] = s / Lf[i][i];
                    }
                    for (int i = FBS - 1; i >= 0; --i) {
                        double s = x[i];
                        for (int k = i + 1; k < FBS; ++k) s -= Lf[k][i] * x[k];
                        x[i] = s / Lf[i][i];
                    }
                    for (int i = 0; i < FBS; ++i) S[oPT + (f * FBS + i) + l * NF] = x[i];
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
#pragma unroll 1
            for (int e = l; e < MS * MS; e += 64) {
                const int i = e % MS, jc = e / MS;
                double s = 0.0;
#pragma unroll
                for (int fk = 0; fk < NF; ++fk) s += OPF[fk + i * NF] * WF[fk + jc * NF];
                const double st = s * (1.0 / hT), dt = S[oDATA + e];
                if (a.lc != nullptr) a.lc[off + e] = dt + st;
                if (a.data != nullptr) a.data[off + e] = dt;
                if (a.stab != nullptr) a.stab[off + e] = st;
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
                    const double x = a.ir_xyw[3 * q], y = a.ir_xyw[3 * q + 1];
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
                    S[oTPHI + HROW * l + 2 * (RD + 1)] = a.ir_xyw[3 * q + 2] * (a.bcs_fn == FN_SAMPLED ? a.bcs_vals[q] : builtin_fn(a.bcs_fn, x, y));
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
