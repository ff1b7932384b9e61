// cut_interface_device.hpp -- the two-sided interface problem of `cuthho_square -i`:
//   make_hho_laplacian_interface                        apps/cuthho/cuthho_square.cpp:390-502
//   make_rhs(msh, cl, degree, where, f)                 src/methods/cuthho_bits/cuthho_utils.hpp:65-84
//   the stabilization blocks of run_cuthho_interface    apps/cuthho/cuthho_square.cpp:1694-1705
//   interface_assembler::assemble / assemble_cut        apps/cuthho/cuthho_square.cpp:1203-1354
// One wavefront per cut cell.  Unknown order of a cut cell: [cell-, cell+, faces-, faces+].
//
// gr_lhs (2 rbs x 2 rbs) is symmetric positive SEMI-definite: its kernel is "the same constant on
// both sides" and every column of gr_rhs is orthogonal to it.  The reference hands it to Eigen's
// pivoted LDLT (:498); here the constant of the negative side is pinned to zero and the remaining
// (2 rbs - 1) system, which is positive definite, is factored by Cholesky.  `data = gr_rhs^T oper`
// (:499) does not depend on the kernel component; `oper` is returned with a zero first row.
#pragma once

#include "cut_device.hpp"

namespace pa {

struct CutInterfaceArgs {
    const double *points;
    const uint32_t *ptids;
    const uint32_t *cut_cells;
    uint32_t ncut;
    const uint32_t *cell_off[2];         // cut-cell quadrature of the negative / positive side
    const double *cell_xyw[2];
    const uint32_t *il_off;              // interface points, integrate_interface(.., IN_NEGATIVE_SIDE)  (:437)
    const double *il_xyw;
    const double *fl_xyw[2];             // face points of each side at degree 2*recdeg
    const int32_t *fl_cnt[2];
    LevelSet ls;
    int rhs_fn;
    double kappa[2], eta;                // params<T>, cuthho_square.cpp:293-299
    double *oper, *data, *rhs;           // ncut x (2rbs x 2msize), ncut x (2msize)^2, ncut x 2cbs
    int32_t *info;
};

// DD (the default): the two-sided reconstruction system in double-double, like the one-sided one (cut_device.hpp, section 3.2 of
// DESIGN.md): its pinned (2 rbs - 1) system is badly conditioned on EVERY cut cell (1-norm condition numbers: median 1e7 at k = 2).
template <int FD, bool DD = true>
__global__ __launch_bounds__(64, 2) void cut_interface_kernel(CutInterfaceArgs a)
{
    constexpr int RD = FD + 1, RBS = P2(RD), CBS = RBS, FBS = FD + 1, NF = 4 * FBS, MS = CBS + NF;
    constexpr int N2 = 2 * RBS, M2 = 2 * MS, NR = N2 - 1, LD2 = (N2 + 1) & ~1, LDR = (NR + 1) & ~1;
    constexpr int NMOM = P2(2 * RD), NFPT = 4 * FACE_SLOTS, CH = 64, NPW = 2 * RD + 1;
    constexpr int PW = 2 * NPW + 1, ROWW = imax(2 * RBS, PW);
    constexpr int W = DD ? 2 : 1;                               // DD: (hi, lo) pairs; the pinned system is factored in place in ST
    constexpr int oMOM = 0, oST = (oMOM + W * NMOM + 1) & ~1, oLL = oST + W * LD2 * N2, oGR = DD ? oLL : oLL + LDR * NR + (LDR * NR & 1);
    constexpr int oOP = oGR + W * N2 * M2, oTPHI = oOP + W * N2 * M2, oTDN = oTPHI + CH * RBS, oTW = oTPHI + CH * ROWW;
    constexpr int oFB = oTW + CH, oEND = oFB + NFPT * FBS;
    // DD tables on the point table: stage B, chunks of CHB points: phi, then w dn; afterwards the sums A (rbs^2 pairs) and C; stage C:
    // phi and w dn of the NFPT face points, then their face-basis values
    constexpr int CHB = imin(32, (CH * ROWW) / (4 * RBS));      // (<= 32: a pair per point in the CH doubles of the weight table)
    constexpr int oBPH = oTPHI, oBDN = oTPHI + 2 * CHB * RBS, oBA = oTPHI, oBC = oTPHI + 2 * RBS * RBS;
    constexpr int oCPH = oTPHI, oCDN = oTPHI + 2 * NFPT * RBS, oCFB = oCDN + 2 * NFPT * RBS;
    static_assert(!DD || (4 * CHB * RBS <= CH * ROWW && 4 * RBS * RBS <= CH * ROWW && oCFB + 2 * NFPT * FBS <= oTPHI + CH * ROWW && CHB >= 1),
                  "the double-double tables fit the point table");
    static_assert(M2 <= 64 && NMOM <= 64 && NR <= 64, "one lane per column / moment / row");
    __shared__ __attribute__((aligned(16))) double S[oEND];
    const int l = threadIdx.x;

    for (uint32_t cc = blockIdx.x; cc < a.ncut; cc += gridDim.x) {
        const uint32_t cell = a.cut_cells[cc];
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
        double hT = 0.0;                                                   // measure(msh, cl): the WHOLE cell (:434)
#pragma unroll
        for (int i = 1; i < 3; ++i)
            hT += fabs((px[i] - px[0]) * (py[i + 1] - py[0]) - (py[i] - py[0]) * (px[i + 1] - px[0])) * 0.5;
        const double eta_h = a.eta / hT;

        auto phi_m = [&](double bx, double by, int m) {
            int p, r; mono_exps(m, p, r);
            return ipow(bx, p) * ipow(by, r);
        };
        auto grad_m = [&](double bx, double by, int m, double &gx, double &gy) {
            int p, r; mono_exps(m, p, r);
            gx = p == 0 ? 0.0 : p * ih * ipow(bx, p - 1) * ipow(by, r);
            gy = r == 0 ? 0.0 : r * ih * ipow(bx, p) * ipow(by, r - 1);
        };

        int bad = 0;
        if constexpr (DD) {
        for (int e = l; e < 2 * LD2 * N2; e += 64) S[oST + e] = 0.0;
        wave_sync();
        const dd ih2 = two_prod(ih, ih);
        // ---- A: per side, moments over the side's cut quadrature (one lane per POINT, double-double accumulators, butterfly over the
        // lanes) -> kappa_s * stiffness block (:419-432); the side's volume right-hand side rides along in double (cuthho_utils.hpp:75-81)
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            const uint32_t c0 = a.cell_off[side][cc], c1 = a.cell_off[side][cc + 1];
            const double *xyw = a.cell_xyw[side];
            {
                dd macc[NMOM];
                double racc[CBS];
#pragma unroll
                for (int m = 0; m < NMOM; ++m) macc[m] = dd_from(0.0);
#pragma unroll
                for (int m = 0; m < CBS; ++m) racc[m] = 0.0;
                for (uint32_t q = c0 + l; q < c1; q += 64) {
                    const double x = xyw[3 * q], y = xyw[3 * q + 1], w = xyw[3 * q + 2];
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    dd pbx[NPW], pby[NPW];                  // w bx^e, by^e
                    pbx[0] = dd_from(w); pby[0] = dd_from(1.0);
#pragma unroll
                    for (int e = 1; e < NPW; ++e) { pbx[e] = dd_mul_d(pbx[e - 1], bx); pby[e] = dd_mul_d(pby[e - 1], by); }
#pragma unroll
                    for (int k = 0; k < NPW; ++k)
#pragma unroll
                        for (int r = 0; r <= k; ++r) macc[k * (k + 1) / 2 + r] = dd_add_fast(macc[k * (k + 1) / 2 + r], dd_mul(pbx[k - r], pby[r]));
                    if (a.rhs != nullptr) {
                        const double fv = builtin_fn(a.rhs_fn, x, y);
#pragma unroll
                        for (int k = 0; k <= RD; ++k)
#pragma unroll
                            for (int r = 0; r <= k; ++r) racc[k * (k + 1) / 2 + r] += fv * (pbx[k - r].hi * pby[r].hi);
                    }
                }
                // (the transposed butterfly of dd_arith.hpp: 29 exchanges for the 28 moments instead of 168)
                {
                    dd tot; bool ok = true;
                    const int m = lanes_transpose_reduce<NMOM, 32>(macc, l, dd_from(0.0), [](dd u, dd v_) { return dd_add_fast(u, v_); },
                                                                   [](dd u, int off) { return dd_shfl_xor(u, off); }, tot, ok);
                    if (ok) dd_store(S + oMOM + 2 * m, tot);
                }
                if (a.rhs != nullptr) {
                    double tot; bool ok = true;
                    const int m = lanes_transpose_reduce<CBS, 32>(racc, l, 0.0, [](double u, double v_) { return u + v_; },
                                                                  [](double u, int off) { return __shfl_xor(u, off); }, tot, ok);
                    if (ok) a.rhs[(size_t)cc * (2 * CBS) + side * CBS + m] = tot;      // :1710-1711 (the lanes that end with one entry hold the same sum)
                }
            }
            wave_sync();
            const dd ks = dd_mul_d(ih2, a.kappa[side]);
            for (int e = l; e < RBS * RBS; e += 64) {
                int ai, bi, aj, bj;
                mono_exps(e % RBS, ai, bi);
                mono_exps(e / RBS, aj, bj);
                dd v = dd_from(0.0);
                if (ai * aj) v = dd_add(v, dd_mul_d(dd_load(S + oMOM + 2 * mono_index(ai + aj - 2, bi + bj)), (double)(ai * aj)));
                if (bi * bj) v = dd_add(v, dd_mul_d(dd_load(S + oMOM + 2 * mono_index(ai + aj, bi + bj - 2)), (double)(bi * bj)));
                dd_store(S + oST + 2 * ((side * RBS + e % RBS) + (side * RBS + e / RBS) * LD2), dd_mul(v, ks));
            }
            wave_sync();
        }
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
                    const int p_ = kk - ii, r_ = ii;
                    phi[m] = dd_mul(pbx[p_], pby[r_]);
                    gx[m] = p_ == 0 ? dd_from(0.0) : dd_mul(dd_mul(pbx[p_ > 0 ? p_ - 1 : 0], pby[r_]), two_prod((double)p_, ih));
                    gy[m] = r_ == 0 ? dd_from(0.0) : dd_mul(dd_mul(pbx[p_], pby[r_ > 0 ? r_ - 1 : 0]), two_prod((double)r_, ih));
                }
        };
        // ---- B: interface terms (:437-459): A_ij = sum k1 w phi_i (dphi_j.n), C_ij = sum k1 w eta/hT phi_i phi_j;
        //   (-,-) += C - A - A^T ;  (+,-) += A - C ;  (-,+) += A^T - C ;  (+,+) += C
        // Through the moments of the interface measures w, w n_x, w n_y as in cut_device.hpp (stage B there): with P = p_i + p_j,
        // R = r_i + r_j:  C_ij = k1 eta/hT M(P, R),  A_ij = k1 ih (p_j Mx(P - 1, R) + r_j My(P, R - 1)).
        {
            const uint32_t i0 = a.il_off[cc], i1 = a.il_off[cc + 1];
            constexpr int DM = 2 * RD, NM1 = P2(DM - 1);
            constexpr int oIM = oTPHI, oIX = oIM + 2 * NMOM, oIY = oIX + 2 * NM1;
            static_assert(2 * (NMOM + 2 * NM1) <= CH * ROWW, "the interface moments fit the point table");
            const bool in0 = i0 + l < i1;
            const double x0 = in0 ? a.il_xyw[3 * (i0 + l)] : barx, y0 = in0 ? a.il_xyw[3 * (i0 + l) + 1] : bary;
            const double w0 = in0 ? a.il_xyw[3 * (i0 + l) + 2] : 0.0;
            double nx0, ny0;
            a.ls.normal(x0, y0, nx0, ny0);
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
                    if (first) {
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
                constexpr int KS = DM >= 4 ? DM - 2 : 0;
                iface_moments(I0(), I0(), std::integral_constant<int, KS>(), S + oIM);
                if constexpr (KS < DM) iface_moments(I0(), std::integral_constant<int, KS + 1>(), std::integral_constant<int, DM>(), S + oIM);
                iface_moments(std::integral_constant<int, 1>(), I0(), std::integral_constant<int, DM - 1>(), S + oIX);
                iface_moments(std::integral_constant<int, 2>(), I0(), std::integral_constant<int, DM - 1>(), S + oIY);
            }
            wave_sync();
            const dd k1e = two_prod(a.kappa[0], eta_h), k1i = two_prod(a.kappa[0], ih);
            dd vA[(RBS * RBS + 63) / 64], vAt[(RBS * RBS + 63) / 64], vC[(RBS * RBS + 63) / 64];
#pragma unroll
            for (int u = 0; u < (RBS * RBS + 63) / 64; ++u) {
                const int e = l + 64 * u;
                vA[u] = vAt[u] = vC[u] = dd_from(0.0);
                if (e < RBS * RBS) {
                    const int i = e % RBS, j = e / RBS;
                    int p1, r1, p2, r2;
                    mono_exps(i, p1, r1);
                    mono_exps(j, p2, r2);
                    const int P = p1 + p2, R = r1 + r2;
                    const dd mx = P > 0 ? dd_load(S + oIX + 2 * mono_index(P > 0 ? P - 1 : 0, R)) : dd_from(0.0);
                    const dd my = R > 0 ? dd_load(S + oIY + 2 * mono_index(P, R > 0 ? R - 1 : 0)) : dd_from(0.0);
                    vC[u] = dd_mul(dd_load(S + oIM + 2 * mono_index(P, R)), k1e);
                    vA[u] = dd_mul(dd_add(dd_mul_d(mx, (double)p2), dd_mul_d(my, (double)r2)), k1i);
                    vAt[u] = dd_mul(dd_add(dd_mul_d(mx, (double)p1), dd_mul_d(my, (double)r1)), k1i);
                }
            }
            wave_sync();                                   // (the moment tables are dead: the point table is free again)
#pragma unroll
            for (int u = 0; u < (RBS * RBS + 63) / 64; ++u) {
                const int e = l + 64 * u;
                if (e < RBS * RBS) {
                    const int i = e % RBS, j = e / RBS;
                    const dd A_ = vA[u], At = vAt[u], C_ = vC[u];
                    double *p00 = S + oST + 2 * (i + j * LD2), *p10 = S + oST + 2 * ((RBS + i) + j * LD2);
                    double *p01 = S + oST + 2 * (i + (RBS + j) * LD2), *p11 = S + oST + 2 * ((RBS + i) + (RBS + j) * LD2);
                    dd_store(p00, dd_add(dd_load(p00), dd_sub(dd_sub(C_, A_), At)));
                    dd_store(p10, dd_add(dd_load(p10), dd_sub(A_, C_)));
                    dd_store(p01, dd_add(dd_load(p01), dd_sub(At, C_)));
                    dd_store(p11, dd_add(dd_load(p11), C_));
                }
            }
            wave_sync();
        }
        // ---- C: gr_rhs (:461-495).  Columns: [cell- | cell+ | faces- | faces+]
        for (int e = l; e < N2 * M2; e += 64) {
            const int i = e % N2, j = e / N2;
            const dd v = j < CBS ? dd_load(S + oST + 2 * (i + j * LD2)) : (j < 2 * CBS ? dd_load(S + oST + 2 * (i + (RBS + j - CBS) * LD2)) : dd_from(0.0));
            dd_store(S + oGR + 2 * e, v);
        }
        wave_sync();
#pragma unroll 1
        for (int side = 0; side < 2; ++side) {
            if (l < NFPT) {
                const int f = l / FACE_SLOTS, qq = l % FACE_SLOTS;
                const bool ok = qq < a.fl_cnt[side][cc * 4 + f];
                const double *src = a.fl_xyw[side] + (((size_t)cc * 4 + f) * FACE_SLOTS + qq) * 3;
                const double x = ok ? src[0] : barx, y = ok ? src[1] : bary;
                const dd kw = ok ? two_prod(a.kappa[side], src[2]) : dd_from(0.0);
                const int f1 = (f + 1) & 3;
                const double ex = px[f1] - px[f], ey = py[f1] - py[f];
                const double len = sqrt(ex * ex + ey * ey);
                const double nx = ey / len, ny = -ex / len;                      // outward normal of the CELL on both sides (:464)
                const bool flip = ids[f] > ids[f1];
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
                    dd_store(S + oCDN + 2 * (l * RBS + m), dd_mul(dd_add(dd_mul_d(gx[m], nx), dd_mul_d(gy[m], ny)), kw));
                }
                dd pe = dd_from(1.0);
#pragma unroll
                for (int k = 0; k < FBS; ++k) { dd_store(S + oCFB + 2 * (l * FBS + k), pe); pe = dd_mul_d(pe, ep); }
            }
            wave_sync();
            for (int e = l; e < RBS * MS; e += 64) {
                const int i = e % RBS, j = e / RBS;
                if (j < CBS) {                                              // :479, :491
                    double *dst = S + oGR + 2 * ((side * RBS + i) + (side * CBS + j) * N2);
                    dd sacc = dd_load(dst);
                    for (int p_ = 0; p_ < NFPT; ++p_) sacc = dd_sub_fast(sacc, dd_mul(dd_load(S + oCDN + 2 * (p_ * RBS + i)), dd_load(S + oCPH + 2 * (p_ * RBS + j))));
                    dd_store(dst, sacc);
                } else {                                                    // :480-481, :492-493
                    const int f = (j - CBS) / FBS, k = (j - CBS) % FBS;
                    dd sacc = dd_from(0.0);
                    for (int qq = 0; qq < FACE_SLOTS; ++qq)
                        sacc = dd_add_fast(sacc, dd_mul(dd_load(S + oCDN + 2 * ((f * FACE_SLOTS + qq) * RBS + i)), dd_load(S + oCFB + 2 * ((f * FACE_SLOTS + qq) * FBS + k))));
                    dd_store(S + oGR + 2 * ((side * RBS + i) + (2 * CBS + side * NF + f * FBS + k) * N2), sacc);
                }
            }
            wave_sync();
        }
        // ---- D: the first unknown pinned (see the header comment): Cholesky of ST[1:, 1:] and both substitutions in REGISTERS, one lane
        // per column of [ST[1:, 1:] | gr_rhs[1:, :]] (NR + M2 lanes: 63 of the 64 at k = 2), right-looking as in cut_device.hpp: at
        // pivot j every lane forms y = v_j / sqrt(d), then v_i -= L_ij y for the rows below with L_ij read from lane j (v_readlane);
        // the backward substitution column-oriented the same way.  (The left-looking factorization on the LDS image and the
        // substitutions of NR (NR - 1) / 2 = 171 dependent steps each way on M2 lanes were most of this kernel's time.)
        {
            static_assert(NR + M2 <= 64, "a lane per column of the pinned system and of gr_rhs");
            const bool isA = l < NR, isG = l >= NR && l < NR + M2;
            const int gc = isG ? l - NR : 0;
            dd v[NR], rsv[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i)
                v[i] = isA ? dd_load(S + oST + 2 * ((i + 1) + (l + 1) * LD2)) : isG ? dd_load(S + oGR + 2 * ((i + 1) + gc * N2)) : dd_from(1.0);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const dd piv = dd_readlane(v[j], j);
                if (!(piv.hi > 0.0) && !bad) bad = j + 1;
                const dd rs = dd_rsqrt_1(piv);
                rsv[j] = rs;
                const dd y = dd_mul(v[j], rs);
#pragma unroll
                for (int i = j + 1; i < NR; ++i) {
                    const dd scaled = dd_mul(v[i], rs);                  // (lane j: L_ij)
                    const dd lij = dd_readlane(scaled, j);
                    const dd upd = dd_sub_fast(v[i], dd_mul(lij, y));
                    v[i].hi = l == j ? scaled.hi : l > j ? upd.hi : v[i].hi;
                    v[i].lo = l == j ? scaled.lo : l > j ? upd.lo : v[i].lo;
                }
                v[j].hi = l >= j ? y.hi : v[j].hi;
                v[j].lo = l >= j ? y.lo : v[j].lo;
            }
#pragma unroll
            for (int i = NR - 1; i >= 0; --i) {
                const dd x = dd_mul(v[i], rsv[i]);
                if (isG) v[i] = x;
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    const dd lik = dd_readlane(v[i], k);                 // entry i of column k of L (lanes < NR are not touched here)
                    const dd upd = dd_sub_fast(v[k], dd_mul(lik, x));
                    if (isG) v[k] = upd;
                }
            }
            if (isG) {
                dd_store(S + oOP + 2 * (gc * N2), dd_from(0.0));          // the pinned unknown
#pragma unroll
                for (int k = 0; k < NR; ++k) dd_store(S + oOP + 2 * ((k + 1) + gc * N2), v[k]);
            }
            wave_sync();
            if (a.oper != nullptr)
                for (int e = l; e < N2 * M2; e += 64) a.oper[(size_t)cc * (N2 * M2) + e] = dd_round(dd_load(S + oOP + 2 * e));
        }
        // ---- E: data = gr_rhs^T oper (:499), symmetric: the entries i <= j, two running sums each, rounded once, both copies stored
        if (a.data != nullptr) {
            const size_t off = (size_t)cc * (M2 * M2);
#pragma unroll 1
            for (int e = l; e < M2 * (M2 + 1) / 2; e += 64) {
                int j = (int)((__fsqrt_rn((float)(8 * e + 1)) - 1.0f) * 0.5f);
                j += (j + 1) * (j + 2) / 2 <= e ? 1 : 0;
                j -= j * (j + 1) / 2 > e ? 1 : 0;
                const int i = e - j * (j + 1) / 2;
                dd s0 = dd_from(0.0), s1 = dd_from(0.0);
#pragma unroll 4
                for (int k = 0; k < N2; k += 2) {
                    s0 = dd_add_fast(s0, dd_mul(dd_load(S + oGR + 2 * (k + i * N2)), dd_load(S + oOP + 2 * (k + j * N2))));
                    s1 = dd_add_fast(s1, dd_mul(dd_load(S + oGR + 2 * (k + 1 + i * N2)), dd_load(S + oOP + 2 * (k + 1 + j * N2))));
                }
                const double r = dd_round(dd_add_fast(s0, s1));
                a.data[off + i + (size_t)j * M2] = r;
                a.data[off + j + (size_t)i * M2] = r;
            }
        }
        } else {
            for (int e = l; e < LD2 * N2; e += 64) S[oST + e] = 0.0;
            wave_sync();

            // ---- A: per side, cell moments -> kappa_s * stiffness block (:419-432) and the volume
            // right-hand side of that side (cuthho_utils.hpp:75-81; degree == recdeg: the same points)
            int mp = 0, mr = 0, rp = 0, rr = 0;
            if (l < NMOM) mono_exps(l, mp, mr);
            if (l < CBS) mono_exps(l, rp, rr);
    #pragma unroll 1
            for (int side = 0; side < 2; ++side) {
                const uint32_t c0 = a.cell_off[side][cc], c1 = a.cell_off[side][cc + 1];
                const double *xyw = a.cell_xyw[side];
                double mom_acc = 0.0, rhs_acc = 0.0;
                for (uint32_t base = c0; base < c1; base += CH) {
                    const uint32_t q = base + l;
                    if (q < c1) {
                        const double x = xyw[3 * q], y = xyw[3 * q + 1], w = xyw[3 * q + 2];
                        const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                        double vx = w, vy = 1.0;
                        for (int e = 0; e < NPW; ++e) {
                            S[oTPHI + l * ROWW + e] = vx;
                            S[oTPHI + l * ROWW + NPW + e] = vy;
                            vx *= bx; vy *= by;
                        }
                        S[oTPHI + l * ROWW + 2 * NPW] = a.rhs != nullptr ? builtin_fn(a.rhs_fn, x, y) : 0.0;
                    }
                    wave_sync();
                    const int nq = (int)((c1 - base) < (uint32_t)CH ? (c1 - base) : (uint32_t)CH);
                    if (l < NMOM)
                        for (int t = 0; t < nq; ++t) mom_acc += S[oTPHI + t * ROWW + mp] * S[oTPHI + t * ROWW + NPW + mr];
                    if (l < CBS)
                        for (int t = 0; t < nq; ++t)
                            rhs_acc += (S[oTPHI + t * ROWW + rp] * S[oTPHI + t * ROWW + NPW + rr]) * S[oTPHI + t * ROWW + 2 * NPW];
                    wave_sync();
                }
                if (l < NMOM) S[oMOM + l] = mom_acc;
                if (a.rhs != nullptr && l < CBS) a.rhs[(size_t)cc * (2 * CBS) + side * CBS + l] = rhs_acc;      // :1710-1711
                wave_sync();
                const double ks = a.kappa[side] * ih * ih;
                for (int e = l; e < RBS * RBS; e += 64) {
                    int ai, bi, aj, bj;
                    mono_exps(e % RBS, ai, bi);
                    mono_exps(e / RBS, aj, bj);
                    double v = 0.0;
                    if (ai * aj) v += (double)(ai * aj) * S[oMOM + mono_index(ai + aj - 2, bi + bj)];
                    if (bi * bj) v += (double)(bi * bj) * S[oMOM + mono_index(ai + aj, bi + bj - 2)];
                    S[oST + (side * RBS + e % RBS) + (side * RBS + e / RBS) * LD2] = ks * v;
                }
                wave_sync();
            }

            // ---- B: interface terms (:437-459) with a = k1 w phi (dphi.n)^T, b = a^T, c = k1 w eta/hT phi phi^T:
            //   (-,-) -= a + b - c ;  (+,-) += a - c ;  (-,+) += b - c ;  (+,+) += c
            {
                const uint32_t i0 = a.il_off[cc], i1 = a.il_off[cc + 1];
                double acc_a[(RBS * RBS + 63) / 64], acc_at[(RBS * RBS + 63) / 64], acc_c[(RBS * RBS + 63) / 64];
    #pragma unroll
                for (int u = 0; u < (RBS * RBS + 63) / 64; ++u) acc_a[u] = acc_at[u] = acc_c[u] = 0.0;
                for (uint32_t base = i0; base < i1; base += CH) {
                    const uint32_t q = base + l;
                    if (q < i1) {
                        const double x = a.il_xyw[3 * q], y = a.il_xyw[3 * q + 1];
                        const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                        double nx, ny;
                        a.ls.normal(x, y, nx, ny);
                        for (int m = 0; m < RBS; ++m) {
                            double gx, gy;
                            grad_m(bx, by, m, gx, gy);
                            S[oTPHI + l * RBS + m] = phi_m(bx, by, m);
                            S[oTDN + l * RBS + m] = gx * nx + gy * ny;
                        }
                        S[oTW + l] = a.kappa[0] * a.il_xyw[3 * q + 2];
                    } else S[oTW + l] = 0.0;
                    wave_sync();
                    const int nq = (int)((i1 - base) < (uint32_t)CH ? (i1 - base) : (uint32_t)CH);
    #pragma unroll
                    for (int u = 0; u < (RBS * RBS + 63) / 64; ++u) {
                        const int e = l + 64 * u;
                        if (e < RBS * RBS) {
                            const int i = e % RBS, j = e / RBS;
                            for (int t = 0; t < nq; ++t) {
                                const double w = S[oTW + t], pi_ = S[oTPHI + t * RBS + i], pj = S[oTPHI + t * RBS + j];
                                acc_a[u] += w * pi_ * S[oTDN + t * RBS + j];
                                acc_at[u] += w * S[oTDN + t * RBS + i] * pj;
                                acc_c[u] += w * eta_h * pi_ * pj;
                            }
                        }
                    }
                    wave_sync();
                }
    #pragma unroll
                for (int u = 0; u < (RBS * RBS + 63) / 64; ++u) {
                    const int e = l + 64 * u;
                    if (e < RBS * RBS) {
                        const int i = e % RBS, j = e / RBS;
                        S[oST + i + j * LD2] += -acc_a[u] - acc_at[u] + acc_c[u];
                        S[oST + (RBS + i) + j * LD2] += acc_a[u] - acc_c[u];
                        S[oST + i + (RBS + j) * LD2] += acc_at[u] - acc_c[u];
                        S[oST + (RBS + i) + (RBS + j) * LD2] += acc_c[u];
                    }
                }
                wave_sync();
            }

            // ---- C: gr_rhs (:461-495).  Columns: [cell- | cell+ | faces- | faces+]
            for (int e = l; e < N2 * M2; e += 64) {
                const int i = e % N2, j = e / N2;
                S[oGR + e] = j < CBS ? S[oST + i + j * LD2] : (j < 2 * CBS ? S[oST + i + (RBS + j - CBS) * LD2] : 0.0);
            }
            wave_sync();
    #pragma unroll 1
            for (int side = 0; side < 2; ++side) {
                if (l < NFPT) {
                    const int f = l / FACE_SLOTS, qq = l % FACE_SLOTS;
                    const bool ok = qq < a.fl_cnt[side][cc * 4 + f];
                    const double *src = a.fl_xyw[side] + (((size_t)cc * 4 + f) * FACE_SLOTS + qq) * 3;
                    const double x = ok ? src[0] : barx, y = ok ? src[1] : bary, w = ok ? a.kappa[side] * src[2] : 0.0;
                    const int f1 = (f + 1) & 3;
                    const double ex = px[f1] - px[f], ey = py[f1] - py[f];
                    const double len = sqrt(ex * ex + ey * ey);
                    const double nx = ey / len, ny = -ex / len;                      // outward normal of the CELL on both sides (:464)
                    const bool flip = ids[f] > ids[f1];
                    const double ax = flip ? px[f1] : px[f], ay = flip ? py[f1] : py[f];
                    const double bxx = flip ? px[f] : px[f1], byy = flip ? py[f] : py[f1];
                    const double fbx = 0.5 * (ax + bxx), fby = 0.5 * (ay + byy);
                    const double ep = 4.0 * ((fbx - ax) * (x - fbx) + (fby - ay) * (y - fby)) / (len * len);
                    const double bx = (x - barx) * ihalf, by = (y - bary) * ihalf;
                    for (int m = 0; m < RBS; ++m) {
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
                    if (j < CBS) {                                              // :479, :491
                        double s = 0.0;
                        for (int p = 0; p < NFPT; ++p) s += S[oTDN + p * RBS + i] * S[oTPHI + p * RBS + j];
                        S[oGR + (side * RBS + i) + (side * CBS + j) * N2] -= s;
                    } else {                                                    // :480-481, :492-493
                        const int f = (j - CBS) / FBS, k = (j - CBS) % FBS;
                        double s = 0.0;
                        for (int qq = 0; qq < FACE_SLOTS; ++qq) s += S[oTDN + (f * FACE_SLOTS + qq) * RBS + i] * S[oFB + (f * FACE_SLOTS + qq) * FBS + k];
                        S[oGR + (side * RBS + i) + (2 * CBS + side * NF + f * FBS + k) * N2] = s;
                    }
                }
                wave_sync();
            }

            // ---- D: oper = gr_lhs^+ gr_rhs with the first unknown pinned (see the header comment)
            for (int e = l; e < NR * LDR; e += 64) {
                const int i = e / LDR, k = e % LDR;
                S[oLL + e] = k < NR ? S[oST + (i + 1) + (k + 1) * LD2] : 0.0;
            }
            wave_sync();
            bad = lds_cholesky<NR, LDR, 64, 2>(S + oLL, l);
            {
                double x[NR];
                const int c = l < M2 ? l : 0;
    #pragma unroll
                for (int k = 0; k < NR; ++k) x[k] = S[oGR + (k + 1) + c * N2];
                lds_forward<NR, LDR>(S + oLL, x);
                lds_backward<NR, LDR>(S + oLL, x);
                if (l < M2) {
                    S[oOP + c * N2] = 0.0;
    #pragma unroll
                    for (int k = 0; k < NR; ++k) S[oOP + (k + 1) + c * N2] = x[k];
                }
            }
            wave_sync();
            if (a.oper != nullptr)
                for (int e = l; e < N2 * M2; e += 64) a.oper[(size_t)cc * (N2 * M2) + e] = S[oOP + e];

            // ---- E: data = gr_rhs^T oper (:499)
            if (a.data != nullptr) {
                const size_t off = (size_t)cc * (M2 * M2);
    #pragma unroll 1
                for (int e = l; e < M2 * M2; e += 64) {
                    const int i = e % M2, j = e / M2;
                    double s = 0.0;
    #pragma unroll
                    for (int k = 0; k < N2; ++k) s += S[oGR + k + i * N2] * S[oOP + k + j * N2];
                    a.data[off + e] = s;
                }
            }
        }
        if (a.info != nullptr && l == 0) a.info[cc] = bad;
        wave_sync();
    }
}

// lc of a cut cell (cuthho_square.cpp:1692-1705): data of make_hho_laplacian_interface plus
// kappa_1 * stab_n scattered over the (cell-, faces-) unknowns and kappa_2 * stab_p over (cell+, faces+)
__global__ __launch_bounds__(256) void cut_interface_lc_kernel(uint32_t ncut, int cbs, int nfd, double k1, double k2,
                                                               const double *data, const double *stab_n, const double *stab_p,
                                                               double *lc)
{
    const int ms = cbs + nfd, m2 = 2 * ms;
    for (size_t cc = blockIdx.x; cc < ncut; cc += gridDim.x)
        for (int e = threadIdx.x; e < m2 * m2; e += blockDim.x) {
            const int i = e % m2, j = e / m2;
            // side and one-sided index of a two-sided unknown
            const int si = i < 2 * cbs ? i / cbs : (i - 2 * cbs) / nfd, li = i < 2 * cbs ? i % cbs : cbs + (i - 2 * cbs) % nfd;
            const int sj = j < 2 * cbs ? j / cbs : (j - 2 * cbs) / nfd, lj = j < 2 * cbs ? j % cbs : cbs + (j - 2 * cbs) % nfd;
            double v = data[cc * (size_t)(m2 * m2) + e];
            if (si == sj) v += (si == 0 ? k1 * stab_n[cc * (size_t)(ms * ms) + li + lj * ms] : k2 * stab_p[cc * (size_t)(ms * ms) + li + lj * ms]);
            lc[cc * (size_t)(m2 * m2) + e] = v;
        }
}

// lc = kappa(side of the cell) * data + stab for the uncut cells (cuthho_square.cpp:1670-1679)
__global__ __launch_bounds__(256) void cut_interface_uncut_lc_kernel(size_t ncells, int mm, const int8_t *cell_loc, double k1, double k2,
                                                                     const double *data, const double *stab, double *lc)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncells * mm) return;
    const int8_t loc = cell_loc[t / mm];
    lc[t] = (loc == LOC_NEG ? k1 : k2) * data[t] + stab[t];
}

// interface_assembler (cuthho_square.cpp:1091-1443): tables built on the host by pa_cut_preprocess
struct InterfaceTripletArgs {
    const uint32_t *cell_faces;          // ncells x 4 (global face ids: the cut mesh is never partitioned)
    const int8_t *cell_loc, *face_loc;
    const int32_t *cut_index;
    const int32_t *cell_table, *face_table;   // face_table -1 for Dirichlet faces
    const double *g;                     // nfaces x fbs Dirichlet data or null
    const double *lc, *rhs;              // uncut: ncells x msize^2 / ncells x cbs (rows of cut cells unused)
    const double *lc_cut, *rhs_cut;      // ncut x (2 msize)^2 / ncut x 2cbs
    uint64_t ncells, num_all_cells;
    int cbs, fbs;
    int32_t *rows, *cols; double *vals;              // uncut slots: ncells x msize^2 (cut cells: all -1)
    int32_t *rows_cut, *cols_cut; double *vals_cut;  // ncut x (2 msize)^2
    int32_t *rhs_rows; double *rhs_vals;             // ncells x msize
    int32_t *rhs_rows_cut; double *rhs_vals_cut;     // ncut x 2 msize
};

__global__ __launch_bounds__(256) void interface_triplets_kernel(InterfaceTripletArgs a)
{
    extern __shared__ double sh[];       // dirichlet data (2 msize), then int32 idx (2 msize)
    const int msize = a.cbs + 4 * a.fbs, m2 = 2 * msize;
    double *dd = sh;
    int32_t *idx = reinterpret_cast<int32_t *>(sh + m2);
    for (size_t c = blockIdx.x; c < a.ncells; c += gridDim.x) {
        const bool cut = a.cell_loc[c] == LOC_CUT;
        const int n = cut ? m2 : msize, ncd = cut ? 2 * a.cbs : a.cbs;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int32_t gi; double d = 0.0;
            if (i < ncd) {
                gi = (int32_t)((uint64_t)a.cell_table[c] * a.cbs + i);                               // :1223, :1291
            } else {
                const int u = i - ncd, pass = u / (4 * a.fbs), f = (u % (4 * a.fbs)) / a.fbs, k = u % a.fbs;
                const uint32_t fid = a.cell_faces[4 * c + f];
                const int32_t ft = a.face_table[fid];
                const int dup = (pass == 1 && a.face_loc[fid] == LOC_CUT) ? a.fbs : 0;                // :1319
                gi = ft < 0 ? -1 : (int32_t)(a.num_all_cells * a.cbs + (uint64_t)ft * a.fbs + dup + k);   // :1234, :1321
                if (ft < 0 && a.g != nullptr) d = a.g[(size_t)fid * a.fbs + k];
            }
            idx[i] = gi; dd[i] = d;
        }
        __syncthreads();
        if (!cut) {
            const double *A = a.lc + c * (size_t)(msize * msize);
            for (int e = threadIdx.x; e < msize * msize; e += blockDim.x) {
                const int i = e / msize, j = e % msize;
                const bool keep = idx[i] >= 0 && idx[j] >= 0;
                const size_t o = c * (size_t)(msize * msize) + e;
                a.rows[o] = keep ? idx[i] : -1;
                a.cols[o] = keep ? idx[j] : -1;
                a.vals[o] = A[i + j * msize];
            }
            for (int i = threadIdx.x; i < msize; i += blockDim.x) {
                double s = (i < a.cbs && a.rhs != nullptr) ? a.rhs[c * a.cbs + i] : 0.0;              // :1265
                if (idx[i] >= 0)
                    for (int j = a.cbs; j < msize; ++j)
                        if (idx[j] < 0) s -= A[i + j * msize] * dd[j];                                // :1261
                a.rhs_rows[c * msize + i] = idx[i];
                a.rhs_vals[c * msize + i] = idx[i] >= 0 ? s : 0.0;
            }
        } else {
            const size_t cc = (size_t)a.cut_index[c];
            const double *A = a.lc_cut + cc * (size_t)(m2 * m2);
            for (int e = threadIdx.x; e < msize * msize; e += blockDim.x) {                           // nothing pushed in the uncut slots
                const size_t o = c * (size_t)(msize * msize) + e;
                a.rows[o] = -1; a.cols[o] = -1; a.vals[o] = 0.0;
            }
            for (int i = threadIdx.x; i < msize; i += blockDim.x) { a.rhs_rows[c * msize + i] = -1; a.rhs_vals[c * msize + i] = 0.0; }
            for (int e = threadIdx.x; e < m2 * m2; e += blockDim.x) {                                 // :1337-1347
                const int i = e / m2, j = e % m2;
                const size_t o = cc * (size_t)(m2 * m2) + e;
                const bool keep = idx[i] >= 0 && idx[j] >= 0;      // always true: the reference rejects Dirichlet faces on cut cells (:1304-1305)
                a.rows_cut[o] = keep ? idx[i] : -1; a.cols_cut[o] = keep ? idx[j] : -1;
                a.vals_cut[o] = A[i + j * m2];
            }
            for (int i = threadIdx.x; i < m2; i += blockDim.x) {                                      // :1349
                a.rhs_rows_cut[cc * m2 + i] = idx[i];
                a.rhs_vals_cut[cc * m2 + i] = (i < 2 * a.cbs && a.rhs_cut != nullptr) ? a.rhs_cut[cc * 2 * a.cbs + i] : 0.0;
            }
        }
        __syncthreads();
    }
}

}  // namespace pa
