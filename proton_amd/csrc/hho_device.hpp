// hho_device.hpp -- CDNA4 (gfx950) device code for the per-cell HHO local operators.
//
// WHAT is computed is defined by the reference (paths relative to the reference root):
//   make_hho_laplacian            src/methods/hho_bits/hho.hpp:32-96
//   make_hho_naive_stabilization  hho.hpp:99-148      (h = cell AREA,     hho.hpp:119)
//   make_hho_fancy_stabilization  hho.hpp:155-237     (h = cell DIAMETER, hho.hpp:201)
//   cell_basis / face_basis       src/core/core_bits/bases.hpp:70-195, 241-291
//   integrate (tensor / fan / face), gauss_legendre, triangle_quadrature
//                                 src/core/core_bits/quadratures.hpp:78-158, 238-432
//   barycenter/diameter/measure/normals  src/core/core_bits/basic_geom.hpp:247-372
//
// HOW is not the reference's.  The path is two kernels: the per-cell head -- geometry, cell quadrature, moments,
// stiffness, Cholesky of gr_lhs (and the mass rows / factor of the dense fancy form) -- runs one THREAD per cell
// (hho_pre.hpp) and hands a small record per cell to the kernel of this file, in which G lanes of a 64-wide
// wavefront cooperate on one cell (a -DPA_USE_PRE=0 build keeps the head in this kernel: stages S0-S4 below):
//   * the cell basis is a set of scaled monomials, so every cell integral of phi_i phi_j or
//     grad phi_i . grad phi_j is a *moment*  sum_q w_q bx_q^p by_q^r  of the same quadrature
//     rule: P2(2 recdeg) moments are accumulated and the stiffness / mass
//     matrices are gathered from them -- term by term the same sums the reference forms;
//   * the face basis evaluated at the face Gauss points is t_q^k exactly, so every face mass
//     matrix is (|F|/2) M^ with one constant factorization shared by all faces of all cells;
//   * lane c of a cell owns COLUMN c of every msize-wide matrix (gr_rhs, Y, oper, T, U): uniform
//     loops, broadcast LDS reads, no per-entry index arithmetic;
//   * data = gr_rhs^T (L L^T)^-1 gr_rhs = Y^T Y with Y = L^-1 gr_rhs (forward substitution only);
//   * stab = (1/h) sum_F B_F^T M_F B_F = U^T U with U_F = sqrt(|F|/2h) L^^T B_F, where
//     B_F = M_F^-1 T_F - E_F is the reference's proj2 + proj3 (hho.hpp:222-231); T_F = [trace_F | 0]
//     for the naive stabilization and for the fancy one when celdeg == recdeg (then
//     pi_T^k p_T^k v == p_T^k v and hho.hpp:184-190 cancels);
//   * lc = Z^T Z with Z = [Y; U]: one symmetric rank update -- on v_mfma_f64_16x16x4_f64 when only lc is
//     asked for (one LDS read per lane feeds 16 FMAs), accumulators to HBM directly or through an LDS image;
//     each lane forming the entries (c, c+d mod msize), d = 0..msize/2, with vector FMAs when data and stab
//     are wanted apart;
//   * the block's one wavefront synchronises on its LDS data with wave_sync() (no wait for outstanding
//     stores), and the next cell's record is in flight while the current cell is processed;
//   * reciprocals and square roots are v_rcp/v_rsq seeds refined by Newton steps (<= 1 ulp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__
// argument); 0 = per configuration (Cfg::WAVES)
#ifndef PA_WAVES_PER_EU
#define PA_WAVES_PER_EU 0
#endif
// 1: the per-cell head (geometry, cell quadrature, moments, stiffness, Cholesky of gr_lhs) comes from the
// one-thread-per-cell pre-pass of hho_pre.hpp; 0: the cooperative kernel computes it itself (A/B builds)
// the same for the condensed-mode instance (0 = as the local-operator kernel)
#ifndef PA_COND_WAVES_PER_EU
#define PA_COND_WAVES_PER_EU 0
#endif
#ifndef PA_USE_PRE
#define PA_USE_PRE 1
#endif

// A/B switches of measured design choices; the defaults are what ships:
// msize from which the lc-only kernel stores its accumulators straight to HBM instead of through an LDS image
#ifndef PA_DIRECT_MIN
#define PA_DIRECT_MIN 14
#endif
// 1: lc = Z^T Z with vector FMAs also when only lc is wanted (the matrix pipe is 15 ... 50 % faster)
#ifndef PA_LC_VALU
#define PA_LC_VALU 0
#endif
// bottom-right tile of Z^T Z (msize 17..24) on the vector pipe
#ifndef PA_CORNER_VALU
#define PA_CORNER_VALU 1
#endif
// cell part of U formed by (face, column) units instead of by the column owners
#ifndef PA_UNIT_U
#define PA_UNIT_U 1
#endif
// dense fancy instances (celdeg != recdeg) take mass rows + chol(M1) from the pre-pass record
#ifndef PA_PRE_DENSE
#define PA_PRE_DENSE 1
#endif
// k = 3 class: cell columns of gr_rhs on the matrix pipe
#ifndef PA_GRC_MFMA
#define PA_GRC_MFMA 1
#endif
// ... from this many rows of gr_lhs on (14 at k = 3; the 9 of k = 2 measured slower: see DESIGN.md section 6)
#ifndef PA_GRC_MFMA_MIN_NR
#define PA_GRC_MFMA_MIN_NR 12
#endif
// cell columns of gr_rhs from the cell moments (divergence theorem) instead of the face-point tables
#ifndef PA_LAPG
#define PA_LAPG 1
#endif
// face-basis constants of a lane selected per pass from uniform tables instead of held in registers (fbs <= 3)
#ifndef PA_FACE_SEL
#define PA_FACE_SEL 0
#endif
// corner tile of Z^T Z: U term in closed form, only the Y rows read
#ifndef PA_CORNER_SHORT
#define PA_CORNER_SHORT 1
#endif
// face columns of gr_rhs: the whole table of the face read before the products (k <= 2)
#ifndef PA_S3B_BATCH
#define PA_S3B_BATCH 1
#endif
// forward substitution Y = L^-1 gr_rhs: doubles of L read per LDS round trip (0: row by row)
#ifndef PA_FWD_CAP
#define PA_FWD_CAP 24
#endif
// rows of U ordered so that the second column tile of Z^T Z needs fewer k-steps (see Cfg::UPERM)
#ifndef PA_UPERM
#define PA_UPERM 1
#endif
// forward / backward substitution with L in registers, broadcast by DPP (see Cfg::DPPFWD)
#ifndef PA_DPPFWD
#define PA_DPPFWD 1
#endif
// lc through the LDS image: each cell's image stored as soon as it is complete, all 64 lanes on it (measured: +-1 %, off)
#ifndef PA_EARLY_OUT
#define PA_EARLY_OUT 0
#endif
// condensed mode: the elimination chain in registers with DPP broadcasts of the pivot row (see hho_local_ops_kernel, S9)
#ifndef PA_COND_DPP
#define PA_COND_DPP 1
#endif
// condensed mode: region P outside the image of S9 (more LDS, the record deposit where the other modes have it)
#ifndef PA_COND_OWN_P
#define PA_COND_OWN_P 1
#endif
// blocks of one XCD (blockIdx mod 8) take consecutive cells
#ifndef PA_XCD_MAP
#define PA_XCD_MAP 1
#endif
// condensed mode: pivots per LDS round trip of the partial factorization (1 = the unblocked chain)
#ifndef PA_COND_NB
#define PA_COND_NB 1
#endif
// condensed mode: face count from which the Schur complement A_FF - W^T W is formed on the matrix pipe
#ifndef PA_COND_SCHUR_MFMA_MIN
#define PA_COND_SCHUR_MFMA_MIN 12
#endif
// the per-cell opaque lane index is masked to its range, so that index arithmetic uses 24-bit multiplies
#ifndef PA_LANE_RANGE
#define PA_LANE_RANGE 1
#endif
// matrix-pipe product: no select on the operand of lanes beyond the last column of Z (they feed entries that are never stored)
#ifndef PA_ZCOL_NOSEL
#define PA_ZCOL_NOSEL 1
#endif
// lc through the LDS image: columns of Z permuted in the tiles so that a lane's accumulators are four consecutive rows (16-byte image writes)
#ifndef PA_ACC_PERM
#define PA_ACC_PERM 1
#endif
// two column tiles without the vector-pipe corner (msize 25..32): tile (1,1) of Z^T Z holds face columns only -- its U term in closed form,
// its matrix instructions stop after the rows of Y
#ifndef PA_T11_SHORT
#define PA_T11_SHORT 1
#endif
// condensed mode: number of pivots from which the elimination chain runs in its right-looking form (see cond_dpp_chain_rl;
// measured: -2.5 % at the 15 pivots of k = 3, +-0.5 % at the 10 of k = 2 and the 6 of k = 1)
#ifndef PA_COND_RL_MIN
#define PA_COND_RL_MIN 12
#endif
// 8-byte LDS reads of the product's operands and of the unit form of U kept single (volatile): no ds_read2_b64
#ifndef PA_LDS_NOPAIR
#define PA_LDS_NOPAIR 0
#endif
// S1 with the pre-pass: TWO lanes per face point -- one forms its row of scaled monomials, the other its row of weighted normal
// derivatives -- and ONE common run of LDS stores (a store instruction costs its 6 ... 13 cycles of the store path whatever the
// number of active lanes: 5 instead of 10 of them per pass at k = 2)
#ifndef PA_S1_SPLIT
#define PA_S1_SPLIT 1
#endif
// 1: no separate pre-pass kernel and no record array of the whole piece: every 64 / CPW passes a wavefront of the cooperative
// kernel forms the heads of the 64 cells it visits next itself -- one cell per lane, the thread-per-cell code of hho_pre.hpp -- into a
// ring of 64 records of its own in global memory (45 KB at k = 2: written and read back by the same compute unit within ~0.2 ms), and
// reads them from there as before (per instance: proton_amd/_build.py)
// Wavefront priority by stage (s_setprio 0..3): a wavefront raises its issue priority as it moves through a pass -- head 0, cell
// columns of gr_rhs (S3b) 1, substitutions and the face part (S4..S6) 2, product + output (S7/S8, and the elimination of the
// condensed mode) 3 -- so that of the three wavefronts of a SIMD the one closest to its stores wins the issue slot: the stores of
// a pass go out earlier and the wavefronts of a SIMD drift apart in phase instead of queueing for the same pipe in the same stage.
// Measured (A/B of tagged builds, 5 rounds each in one call, kernels of 1024 x 1024 cells): k = 2 L 1.204 -> 1.143 ms, k = 1 L
// 0.438 -> 0.424, k = 3 L 2.470 -> 2.380, k = 2 C (rhs + operators + fill) 2.21 -> 2.05; out 3 alone 1.175 / 0.429 / 2.399; out 3
// + substitutions 1: 1.160 / 0.424 / 2.395; the reverse (head 3, out 0) 1.202 / 0.433 / 2.483 against 1.212 / 0.440 / 2.513.
// PA_PRIO_OUT = -1: no s_setprio at all (the kernels of rounds 1-2).
#ifndef PA_PRIO_OUT
#define PA_PRIO_OUT 3
#endif
#ifndef PA_PRIO_HEAD
#define PA_PRIO_HEAD 0
#endif
#ifndef PA_PRIO_S3B      /* from S3b (cell columns of gr_rhs) on; -1: as the head */
#define PA_PRIO_S3B 1
#endif
#ifndef PA_PRIO_MID      /* from S4 (substitutions) on; -1: as the stage before */
#define PA_PRIO_MID 2
#endif
#ifndef PA_PRIO_S6       /* from S6 (face part of U, Z complete) on; -1: as the stage before */
#define PA_PRIO_S6 -1
#endif
#ifndef PA_SELF_PRE
#define PA_SELF_PRE 0
#endif
#ifdef PA_MARKERS
#define PA_MARK(x) asm volatile("; PAMARK " x)
#else
#define PA_MARK(x)
#endif
// diagnostic build (-DPA_STAGE_CLOCK): shader-clock stamps at the stage boundaries, summed per block over its cells
// and written to LocalOpsArgs::dbg[block][stage]; PA_TICK(i) closes stage i
#ifdef PA_STAGE_CLOCK
#define PA_NSTAGE 16
#define PA_TICK(i) do { const long long t_ = clock64(); tk_sum[i] += t_ - tk_last; tk_last = t_; } while (0)
#else
#define PA_TICK(i)
#endif
namespace pa {

enum { QUAD_TENSOR = 0, QUAD_FAN = 1 };
enum { STAB_NONE = 0, STAB_NAIVE = 1, STAB_FANCY = 2 };

// Quadrature tables, filled by the host (quad_tables.hpp) and passed by device pointer:
//   gauss_*[n][i]: the n-node rule of gauss_legendre() in the reference's emission order;
//   dun[r][row] = (l0, l1, l2, w) of dunavant rules[r] (0-based, rules[r] == rule_{r+1}).
// Face tables for face degree fd (n = fd + 1 Gauss points, fbs = fd + 1 modes): the face basis at
// the Gauss points is t_q^k, so M_F = (|F|/2) M^ with M^ = sum_q w_q t_q^(i+j) = L^ L^^T.
struct FaceTables {
    double fb[4][4];      // [q][k] t_q^k
    double cw[4][4];      // [q][k] w_q t_q^k
    double lf[4][4];      // [i][k] L^ (lower), diagonal holds 1 / L^[i][i]
    double lft[4][4];     // [j][k] (L^^T)[j][k] = L^[k][j], j <= k
};
struct QuadTables {
    double gauss_x[9][8];      // up to 8 nodes: closed forms to 5 (quadratures.hpp:96-150), golub_welsch's rules beyond (:32-75)
    double gauss_w[9][8];
    double dun[9][16][4];
    int dun_n[9];
    FaceTables face[4];
};

__host__ __device__ constexpr int P2(int d) { return (d + 2) * (d + 1) / 2; }
// points of triangle_quadrature(deg) including the rules[deg] off-by-one (quadratures.hpp:257-268)
__host__ __device__ constexpr int dunavant_points(int deg)
{
    const int d = deg == 0 ? 1 : deg;
    return d == 1 ? 3 : d == 2 ? 4 : d == 3 ? 6 : d == 4 ? 7 : d == 5 ? 12 : d == 6 ? 13 : d == 7 ? 16 : 0;
}
__host__ __device__ constexpr int gauss_nodes(int deg) { return ((deg | 1) + 1) / 2; }
__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ constexpr int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }

// COND_ = 1: the instance of the condensed mode (static condensation fused behind the product, no lc in HBM): its
// region Q also holds the (MS + 1)-row image of [lc f_T; f_T^T 0] the partial factorization works on.
template <int CD_, int FD_, int QUAD_, int STAB_, int G_, int COND_ = 0>
struct Cfg {
    static constexpr int CD = CD_, FD = FD_, RD = FD_ + 1, QUAD = QUAD_, STAB = STAB_, G = G_;
    static constexpr bool COND = COND_ != 0;
    static constexpr int RBS = P2(RD), CBS = P2(CD), FBS = FD + 1;
    static constexpr int NF = 4 * FBS, MS = CBS + NF, NR = RBS - 1;
    static constexpr int QDEG = 2 * RD;                       // hho.hpp:55,174
    static constexpr int NG = gauss_nodes(QDEG);
    static constexpr int NT = dunavant_points(QDEG);          // per fan triangle
    static constexpr int NQ = QUAD == QUAD_TENSOR ? NG * NG : 4 * NT;
    static constexpr int NFQ = gauss_nodes(2 * FD);           // hho.hpp:74,132,208
    static constexpr int NFP = 4 * NFQ;
    static constexpr bool FANCY = STAB == STAB_FANCY, NAIVE = STAB == STAB_NAIVE;
    static constexpr bool GENERAL_FANCY = FANCY && CD != RD;  // T_F is dense; otherwise T_F = [trace_F | 0]
    // The per-cell head is taken from the pre-pass (hho_pre.hpp); for the dense fancy form its record also carries the
    // rows of the cell mass matrix and the Cholesky factor of its leading block (hho.hpp:173-190).
    static constexpr bool USE_PRE = PA_USE_PRE && (PA_PRE_DENSE || !GENERAL_FANCY);
    // Cell columns of gr_rhs (hho.hpp:64-85): stiff[i, c] - sum_F int_F (grad phi_i . n) phi_c = -int_T phi_c lap(phi_i)
    // (divergence theorem; both quadratures of the reference are exact for these integrands), i.e. 0, 1 or 2 cell
    // moments of degree <= recdeg - 2 + celdeg with integer coefficients.  The moments travel with the record.
    static constexpr bool LAPG = USE_PRE && PA_LAPG;
    static constexpr int DG = RD - 2 + CD;                    // highest moment degree that is needed
    static constexpr int NMG = (LAPG && RD >= 2) ? P2(DG) : 0, NMGP = (NMG + 1) & ~1;
    // Every lane of a cell reads the SAME element of L in every step of the substitutions Y = L^-1 gr_rhs and L^-T Y: as LDS
    // reads that is a 16-byte broadcast per two FMAs, a quarter of the kernel's LDS cycles at k = 3 -- and the LDS pipeline is
    // its busiest unit.  Instead the packed factor (and the reciprocal diagonal) sits in registers, element e in lane e mod 16 of
    // each row of 16 lanes (ceil((NL + NR) / 16) doubles per lane, one 8-byte LDS read each), and the FMAs take it through the
    // DPP operand  v_fmac_f64_dpp ... row_newbcast:n  (gfx90a+: lane n of the row of 16, for 64-bit operands).  No image of L.
    static constexpr bool DPPFWD = LAPG && !GENERAL_FANCY && PA_DPPFWD;
    static constexpr int NQB = USE_PRE ? 0 : NQ;              // cell points evaluated by THIS kernel
    static constexpr int NP = NQB + NFP;
    static constexpr int NPW = 2 * RD + 1;                    // powers 0..2 recdeg
    static constexpr int NMOM = P2(2 * RD);
    static constexpr int CPW = 64 / G;                        // cells per wavefront
    static constexpr int PPL = cdiv(NP, G);                   // evaluation points per lane
    static constexpr int MPL = cdiv(NMOM, G);                 // moments per lane
    static constexpr int NSYM = RBS * (RBS + 1) / 2;          // stiffness is symmetric: packed lower triangle
    static constexpr int SPL = cdiv(NSYM, G);                 // stiffness entries per lane
    static constexpr int ND = MS / 2 + 1;                     // rotations of the symmetric update
    // register budget: the k = 3 kernels need > 168 VGPRs to run without spills (measured: 2 waves/SIMD
    // without spills beat 3 with), the others fit 3 waves/SIMD
    // (the dense fancy form carries the mass factor and the dense T_F as well: one step lower, or it spills by the hundred)
    static constexpr bool DENSE_FANCY = STAB_ == STAB_FANCY && CD_ != FD_ + 1;
#ifdef PA_WAVES_OVERRIDE      /* A/B builds: beats the per-instance setting of _build.py */
    static constexpr int WAVES_LC = PA_WAVES_OVERRIDE;
#else
    static constexpr int WAVES_LC = PA_WAVES_PER_EU ? PA_WAVES_PER_EU
                                 : DENSE_FANCY ? ((MS > 24 || RBS > 10) ? 1 : (MS > 16 || RBS > 6) ? 2 : (CBS <= 1 ? 4 : 3))      /* (cell degree 0 -- the obstacle pair (0,1): 120 registers --: 4 against 3 waves: -5 % lc, -17 % condensed; (1,1) spills at 4) */
                                               : ((MS > 24 || RBS > 10) ? 2 : 3);
#endif
    static constexpr int WAVES = (COND_ != 0 && PA_COND_WAVES_PER_EU) ? PA_COND_WAVES_PER_EU : WAVES_LC;
    static constexpr bool HAS_STAB = STAB != STAB_NONE;
    // lc-only path: accumulators -> HBM directly (no LDS image) where the matrix is big enough for the
    // 128-byte runs to pay (measured: -5 % at msize 14, -2 % at 22, -1 % at 31, but +19 % at msize 9)
    static constexpr bool DIRECT_STORE = CBS + 4 * FBS >= PA_DIRECT_MIN;
    // msize 17..24: the bottom-right tile of Z^T Z holds at most 8 x 8 useful entries and would cost the matrix pipe as
    // much as a full one -> vector FMAs, one lane per unordered pair of its columns
    static constexpr int NCORNER = CBS + 4 * FBS - 16;
    static constexpr bool CORNER_VALU = PA_CORNER_VALU && NCORNER > 0 && NCORNER <= 8 && NCORNER * (NCORNER + 1) / 2 <= G;

    // lc with T_F = [trace_F | 0] and 17..32 columns: the second column tile holds face columns only, of the faces UF1..3,
    // and the U rows of the other faces are zero in it.  With the rows of those faces FIRST behind Y (a permutation of the
    // rows of Z does not change Z^T Z) the products of that tile stop after the Y rows and theirs: KS1 k-steps instead of KS.
    static constexpr bool UPERM = PA_UPERM && HAS_STAB && !GENERAL_FANCY && PA_UNIT_U && CBS <= 16 && CBS + 4 * FBS > 16 && CBS + 4 * FBS <= 32;
    static constexpr int UF1 = UPERM ? (16 - CBS) / FBS : 0;
    static constexpr int KS1 = cdiv(((RBS - 1 + 1) & ~1) + (4 - UF1) * FBS, 4);
    // ---- record of a cell written by the pre-pass (doubles): packed lower triangle of L = chol(gr_lhs), row-major,
    // TRUE diagonal | 1/diagonal | pad | sqrt(|F|/2h) x 4, barycenter, 2/h_T, pivot status, the 4 vertices, face
    // orientation bits (bit f: the first vertex of local face f has the HIGHER point id), pad
    struct Pre {
        static constexpr int NL = (RBS - 1) * RBS / 2;
        static constexpr int oSCAL = (NL + (RBS - 1) + 1) & ~1;
        static constexpr int NSCAL = 18 + NMGP;               // ([17]: pivot status of the mass factor; [18..]: moments, LAPG)
        // dense fancy form only: rows i < CBS of the cell mass matrix, [j][i]; packed chol(M1), true diagonal; 1/diagonal
        static constexpr int oMR = oSCAL + NSCAL, NMR = GENERAL_FANCY ? CBS * RBS : 0;
        static constexpr int oMC = oMR + NMR, NMC = GENERAL_FANCY ? CBS * (CBS + 1) / 2 : 0;
        static constexpr int oMCR = oMC + NMC;
        static constexpr int NPRE = (oMCR + (GENERAL_FANCY ? CBS : 0) + 1) & ~1;
        static constexpr int NP2 = NPRE / 2;                  // 16-byte pairs
    };
    static constexpr int PLC = cdiv(Pre::NP2, G);             // pairs of the record per lane
    // the wavefront produces the records it consumes (see PA_SELF_PRE): SELF_IT passes per batch of 64 cells
    static constexpr bool SELF_PRE = USE_PRE && PA_SELF_PRE;
    static constexpr int SELF_IT = 64 / (64 / G);

    // ---- LDS map (doubles, per cell).  Every vector that is read as a contiguous run starts at
    // an even offset and has an even stride, so the reads are 16-byte ds_read_b128.
    static constexpr int NRP = (NR + 1) & ~1;                 // padded row count of gr_lhs / Y
    // stride of the (symmetric) stiffness matrix / of the image of L: even (16-byte rows); for the 14 rows of k = 3 also
    // LD / 2 odd, so that lanes reading DIFFERENT rows with 16-byte loads (row c-1 of L added to cell column c) do not
    // share banks (LD = 16 put the 14 rows on two bank sets: 8-way conflicts, a quarter of the kernel's conflict cycles)
    static constexpr int LD = (((RBS + 1) & ~1) % 4 == 0 && RBS > 10) ? ((RBS + 1) & ~1) + 2 : ((RBS + 1) & ~1);
    static constexpr int ZR = NRP + (HAS_STAB ? NF : 0);      // rows of Z = [Y; pad; U]
    static constexpr int ZS = ((ZR + 1) & ~1) % 4 == 2 ? ((ZR + 1) & ~1) : ((ZR + 1) & ~1) + 2;   // even, ZS/2 odd
    // region Q: quadrature-point tables and moments -- dead once the gr_rhs columns are in registers
    // (after S3b); Z = [Y; U] and later the output image reuse it.
    static constexpr int oWPX = 0;                            // NQ x NPW   w * bx^e
    static constexpr int oPY = oWPX + NQB * NPW;              // NQ x NPW   by^e
    static constexpr int oPHF = oPY + NQB * NPW;              // NFP x RBS  phi at face points
    static constexpr int oDN = (oPHF + NFP * RBS + 1) & ~1;   // NFP x NRP  (w_q/2) (grad phi . edge normal)
    static constexpr int oMOM = oDN + NFP * NRP;              // NMOM moments
    static constexpr int endQ0 = oMOM + (USE_PRE ? 0 : NMOM);
    // gr_rhs cell columns are assembled by SPC lanes each (RPP rows per lane) and meet in GRC, which
    // lies on the w*bx^e / by^e tables (dead after the moments)
    // (measured: -1.7 % at k = 2, nothing at k = 3, +1 % with 16 lanes per cell -> only for G = 32)
    static constexpr int SPC = LAPG ? 1 : G >= 32 ? imin(G / CBS, NR) : 1, RPP = cdiv(NR, SPC < 1 ? 1 : SPC);
    // (with the pre-pass there are no such tables: GRC follows the face tables)
    static constexpr int oGRC = USE_PRE ? endQ0 : 0;          // CBS x NRP
    static constexpr int endQ = endQ0 + (USE_PRE && SPC > 1 ? CBS * NRP : 0);
    static_assert(USE_PRE || SPC <= 1 || CBS * NRP <= 2 * NQ * NPW, "GRC must fit the dead quadrature tables");
    static constexpr int oZ = 0;                              // ZS x MS    (written from S5 on)
    static constexpr int oOUT = 0;                            // MS x MS    (written in S8, after the last read of Z)
    // condensed mode: image of the symmetric matrix [lc f_T; f_T^T 0], MS + 1 rows, row-major == column-major; stride
    // LDI even (16-byte rows) with LDI / 2 odd (the row-per-lane reads touch every bank once).  It lies over regions Q
    // AND P: nothing of P is needed after S6, and the condensed mode deposits the next cell's record only after S9.
    // The packed result (upper triangle of S, then g) is staged on the rows of the cell block once they are dead, or
    // behind the image where that block is too small (cbs <= 3).
    static constexpr int LDI = ((MS + 1) & ~1) % 4 == 2 ? ((MS + 1) & ~1) : ((MS + 1) & ~1) + 2;
    static constexpr int NSP = NF * (NF + 1) / 2, NCOND = NSP + NF;
    static constexpr int oSTG = CBS * LDI >= NCOND ? 0 : (MS + 1) * LDI;
    static constexpr int condNeed = (MS + 1) * LDI + (CBS * LDI >= NCOND ? 0 : NCOND);
    // (condensed mode with COND_OWN_P: the image of S9 lies over region Q only, region P keeps its place behind it and
    // takes the next cell's record as soon as S6 is done, as in the other modes)
    static constexpr bool COND_OWN_P = COND && PA_COND_OWN_P && USE_PRE;
    static constexpr int sizeQ = imax(imax(imax(endQ, ZS * MS), MS * MS), COND_OWN_P ? ((condNeed + 1) & ~1) : 0);
    // region P: lives until the forward substitutions are done.  Stiffness, stride LD; its [1:,1:]
    // block becomes chol(gr_lhs) row by row: oST is odd so that the block (and every row of it)
    // starts on a 16-byte boundary
    static constexpr int oST = sizeQ | 1;
    static constexpr int oMA = (oST + LD * RBS + 1) & ~1;     // RBS x RBS  mass (general fancy); chol(M1) in place
    static constexpr int oFTo = oMA + (GENERAL_FANCY ? LD * RBS : 0);    // NF x RBS trace / (|F|/2) (general fancy)
    static constexpr int oSUo = (oFTo + (GENERAL_FANCY ? NF * RBS : 0) + 1) & ~1;   // 4: sqrt(|F| / 2h)
    // region P with the pre-pass: the image of L (NR x LD, row-major, true diagonal; the upper triangle and one
    // more row are zeroed once per kernel and never written again) and the tail of the record as it comes
    // (reciprocals, scalars), shifted so that the scalars start on a 16-byte boundary
    // ... with DPPFWD: the record as it comes (packed L, reciprocals, scalars, moments), 16-byte pairs in place
    static constexpr int oREC = (sizeQ + 1) & ~1;
    static constexpr int NLR = cdiv(Pre::NL + (RBS - 1), 16);   // registers of the packed factor + reciprocals
    static constexpr int oLGp = (sizeQ + 1) & ~1;
    // (the zero row NR behind the image is what the columns without a stiffness part add; not needed with LAPG)
    static constexpr int LGR = LAPG ? NR : RBS;               // rows of the image of L
    static constexpr int oLIN = ((oLGp + LGR * LD + 1) & ~1) + ((Pre::oSCAL - Pre::NL) & 1);
    static constexpr int oRCP = DPPFWD ? oREC + Pre::NL : oLIN;      // NR: 1 / L[i][i]
    static constexpr int oSU = DPPFWD ? oREC + Pre::oSCAL : USE_PRE ? oLIN + (Pre::oSCAL - Pre::NL) : oSUo;
    static constexpr int oLG = USE_PRE ? oLGp : oST + 1 + LD;  // chol(gr_lhs): stiff[1:,1:] in place without the pre-pass
    static constexpr int oMG = oSU + 18;                      // NMG moments (LAPG)
    static constexpr int oDUMMY = oSU + (USE_PRE ? Pre::NSCAL : 4);    // 4: sink of masked-out stores
    // dense fancy form on the pre-pass: mass rows, image of chol(M1) (row-major, stride LDM), its reciprocals, trace table
    static constexpr int LDM = (CBS + 1) & ~1;
    static constexpr int oMRl = (oDUMMY + 4 + 1) & ~1;
    static constexpr int oMCl = (oMRl + Pre::NMR + 1) & ~1;
    static constexpr int oMCRl = oMCl + CBS * LDM;
    static constexpr int oFTp = (oMCRl + CBS + 1) & ~1;
    static constexpr int oFT = USE_PRE ? oFTp : oFTo;
    static constexpr int LDS_BASE = (USE_PRE && GENERAL_FANCY) ? ((oFTp + NF * RBS + 1) & ~1) : ((oDUMMY + 4 + 1) & ~1);
    static constexpr int LDS_PER_CELL = COND ? imax(LDS_BASE, (condNeed + 1) & ~1) : LDS_BASE;
    static constexpr int LDS_DOUBLES = CPW * LDS_PER_CELL;
};

struct LocalOpsArgs {
    const QuadTables *tab;     // device copy of the quadrature tables
    const double *points;      // np x 2
    const uint32_t *ptids;     // nc x 4
    size_t first, n;
    const double *pre;         // records of Cfg::Pre::NPRE doubles (hho_pre.hpp) in tiles of 8 cells; Cfg::USE_PRE only
    double *pre_ring;          // Cfg::SELF_PRE: gridDim.x rings of 64 records (8 tiles), one per wavefront, written by the kernel itself
    double *oper, *data, *stab, *lc;
    int32_t *info;
    // condensed mode (MODE_COND): f_T per cell (n x cbs, may be null = 0) in; per cell the packed upper triangle of the
    // Schur complement S = A_FF - A_FT A_TT^-1 A_TF (column-packed: S(i,j), i <= j, at j(j+1)/2 + i) followed by
    // g = -A_FT A_TT^-1 f_T, NCOND = nf(nf+1)/2 + nf doubles, out.  With uF (n x nf) the same pass recovers the cell
    // unknowns instead: uT = A_TT^-1 (f_T - A_TF uF) (n x cbs).
    const double *rhs;
    double *cond;
    const double *uF;
    double *uT;
    // Stage mask, 0 in production.  Profiling (PA_ABLATE): bit i skips stage i and the results are garbage.
    // It is ALSO load-bearing: the stages sit in branches on this runtime value, which the compiler cannot
    // prove taken, so it does not hoist their per-lane, cell-invariant subexpressions out of the cell loop.
    // With the branches compiled out the k = 2 kernel spills 26 more VGPRs and runs 7 % slower (measured).
    uint32_t ablate;
    long long *dbg;            // -DPA_STAGE_CLOCK builds only
};

// index of the monomial bx^p by^r in the graded ordering (total degree, then r)  bases.hpp:114-128
__device__ __forceinline__ int mono_index(int p, int r) { const int k = p + r; return k * (k + 1) / 2 + r; }
// exponents (p, r) of monomial m
__device__ __forceinline__ void mono_exps(int m, int &p, int &r)
{
    int k = 0;
    while ((k + 1) * (k + 2) / 2 <= m) ++k;
    r = m - k * (k + 1) / 2;
    p = k - r;
}

// 1/sqrt(x), 1/x, sqrt(x) for x > 0: hardware seed (~2^-26) + two Newton steps, <= 1 ulp
template <int ITERS = 2>
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const double t = x * y;
        const double e = __builtin_fma(-t, y, 1.0);
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-x, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}
__device__ __forceinline__ double fast_sqrt(double x)
{
    const double y = fast_rsqrt<2>(x);
    const double s = x * y;
    const double r = __builtin_fma(-s, s, x);
    return __builtin_fma(0.5 * y, r, s);
}

// Synchronisation of the ONE wavefront of a block on its LDS data.  LDS instructions of a wavefront execute in
// order, so a later read by any lane sees an earlier write by any other: what is needed is that the compiler
// keeps the order, not s_barrier -- and not what __syncthreads() adds to it, a wait for every outstanding global
// store (vmcnt(0)): with it each cell ended with the wavefront idle until HBM had acknowledged its local matrix.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 8-byte LDS read that the compiler may not pair with a neighbour into ds_read2_b64 (volatile, in the LDS address space: a volatile
// access through a generic pointer would become a flat load)
__device__ __forceinline__ double lds_single(const double *p)
{
    typedef const volatile __attribute__((address_space(3))) double *lds_cvp;
    return *(lds_cvp)p;
}

// 16-byte LDS read of two consecutive doubles (p is 16-byte aligned by construction of the LDS map)
__device__ __forceinline__ double2 lds_pair(const double *p) { return *reinterpret_cast<const double2 *>(p); }

// s - sum_{k<N} p[k] x[k] with p a contiguous, 16-byte aligned LDS vector
template <int N>
__device__ __forceinline__ double lds_dotsub(double s, const double *p, const double *x)
{
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        const double2 v = lds_pair(p + k);
        s = __builtin_fma(-v.x, x[k], s);
        s = __builtin_fma(-v.y, x[k + 1], s);
    }
    if (N & 1) s = __builtin_fma(-p[N - 1], x[N - 1], s);
    return s;
}
// same with a count that is a constant after unrolling
__device__ __forceinline__ double lds_dotsub_n(double s, const double *p, const double *x, int n)
{
#pragma unroll
    for (int k = 0; k + 1 < n; k += 2) {
        const double2 v = lds_pair(p + k);
        s = __builtin_fma(-v.x, x[k], s);
        s = __builtin_fma(-v.y, x[k + 1], s);
    }
    if (n & 1) s = __builtin_fma(-p[n - 1], x[n - 1], s);
    return s;
}
template <int N>
__device__ __forceinline__ double lds_dotadd(double s, const double *p, const double *x)
{
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        const double2 v = lds_pair(p + k);
        s = __builtin_fma(v.x, x[k], s);
        s = __builtin_fma(v.y, x[k + 1], s);
    }
    if (N & 1) s = __builtin_fma(p[N - 1], x[N - 1], s);
    return s;
}

// value of lane j of every G-lane group, broadcast to the lanes of that group.  j is a constant after
// unrolling: v_readlane_b32 (scalar result, no LDS round trip like ds_bpermute) + a select per group.
template <int G>
__device__ __forceinline__ double group_broadcast(double v, int j)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int rlo = __builtin_amdgcn_readlane(lo, j), rhi = __builtin_amdgcn_readlane(hi, j);
    if (G < 64) {
        const int grp = (int)(threadIdx.x & 63) / G;
#pragma unroll
        for (int g = 1; g < 64 / G; ++g) {
            const int slo = __builtin_amdgcn_readlane(lo, g * G + j), shi = __builtin_amdgcn_readlane(hi, g * G + j);
            rlo = grp == g ? slo : rlo;
            rhi = grp == g ? shi : rhi;
        }
    }
    return __hiloint2double(rhi, rlo);
}

// -------------------------------------------------------------------------------------
// Cholesky of an N x N SPD matrix in LDS, one lane per row.  The matrix is symmetric and is
// overwritten ROW-major: L[i][k] lands at A[i * LD + k] (rows contiguous and 16-byte aligned,
// so every later use of a row is a run of ds_read_b128); the diagonal receives 1/L[j][j].
// Returns 0 or 1+index of the first non-positive pivot (uniform over the G-lane group).
// -------------------------------------------------------------------------------------
template <int N, int LD, int G, int RSQ_ITERS = 1>
__device__ __forceinline__ int lds_cholesky(double *A, int l)
{
    double row[N];
    const bool act = l < N;
    const int i = act ? l : 0;
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        const double2 v = lds_pair(A + i * LD + k);
        row[k] = v.x; row[k + 1] = v.y;
    }
    if (N & 1) row[N - 1] = A[i * LD + N - 1];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // A[i][j] - sum_{k<j} L[i][k] L[j][k]; row j's prefix is already in LDS
        const double s = j == 0 ? row[0] : lds_dotsub_n(row[j], A + j * LD, row, j);
        const double d = group_broadcast<G>(s, j);
        if (!(d > 0.0) && !bad) bad = j + 1;
        const double r = fast_rsqrt<RSQ_ITERS>(d);   // seed 2^-26 -> ~3e-16 after one Newton step
        row[j] = s * r;
        if (act && l >= j) A[i * LD + j] = (l == j) ? r : row[j];
        wave_sync();
    }
    return bad;
}

// -------------------------------------------------------------------------------------
// Blocked variant (block size 3), same input / output convention as lds_cholesky.  The unblocked
// form is a chain of N steps, each with three dependent LDS round trips (row prefix, broadcast of
// the pivot, write of the new column) that nothing overlaps; here every lane factors the 3 x 3
// diagonal block redundantly in registers, so there is one round trip per block:
//   D = A[J,J] - L[J,:j0] L[J,:j0]^T  (all lanes, from the LDS rows of J)      D = Ld Ld^T
//   lane i: t = A[i,J] - L[i,:j0] L[J,:j0]^T (own row in registers),  L[i,J] = t Ld^-T
// -------------------------------------------------------------------------------------
template <int N, int LD, int G, int RSQ_ITERS = 1>
__device__ __forceinline__ int lds_cholesky_blocked(double *A, int l)
{
    constexpr int NB = 3;
    double row[N];
    const bool act = l < N;
    const int i = act ? l : 0;
#pragma unroll
    for (int k = 0; k + 1 < N; k += 2) {
        const double2 v = lds_pair(A + i * LD + k);
        row[k] = v.x; row[k + 1] = v.y;
    }
    if (N & 1) row[N - 1] = A[i * LD + N - 1];
    int bad = 0;
#pragma unroll
    for (int j0 = 0; j0 < N; j0 += NB) {
        const int nb = (N - j0 < NB) ? (N - j0) : NB;       // constant after unrolling
        // Schur complement of the diagonal block (lower triangle) and of the own row, sharing the reads of rows J
        double d[NB][NB], t[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            if (r < nb) {
                t[r] = row[j0 + r];
#pragma unroll
                for (int c = 0; c <= r; ++c) d[r][c] = A[(j0 + r) * LD + j0 + c];
            }
        }
#pragma unroll
        for (int k = 0; k < j0; k += 2) {                    // rows of J, two columns per 16-byte read
            const bool two = k + 1 < j0;
            double lj[NB], lj1[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                if (r < nb) {
                    if (two) { const double2 v = lds_pair(A + (j0 + r) * LD + k); lj[r] = v.x; lj1[r] = v.y; }
                    else { lj[r] = A[(j0 + r) * LD + k]; lj1[r] = 0.0; }
                }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                if (r < nb) {
                    t[r] = __builtin_fma(-row[k], lj[r], t[r]);
                    if (two) t[r] = __builtin_fma(-row[k + 1], lj1[r], t[r]);
#pragma unroll
                    for (int c = 0; c <= r; ++c) {
                        d[r][c] = __builtin_fma(-lj[r], lj[c], d[r][c]);
                        if (two) d[r][c] = __builtin_fma(-lj1[r], lj1[c], d[r][c]);
                    }
                }
            }
        }
        // factor the block in registers (reciprocal diagonal), solve the own row against it
        double rd[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {
                double p = d[c][c];
#pragma unroll
                for (int k = 0; k < c; ++k) p = __builtin_fma(-d[c][k], d[c][k], p);
                if (!(p > 0.0) && !bad) bad = j0 + c + 1;
                rd[c] = fast_rsqrt<RSQ_ITERS>(p);
#pragma unroll
                for (int r = c + 1; r < NB; ++r) {
                    if (r < nb) {
                        double q = d[r][c];
#pragma unroll
                        for (int k = 0; k < c; ++k) q = __builtin_fma(-d[r][k], d[c][k], q);
                        d[r][c] = q * rd[c];                    // Ld[r][c]
                    }
                }
                double x = t[c];
#pragma unroll
                for (int k = 0; k < c; ++k) x = __builtin_fma(-row[j0 + k], d[c][k], x);
                row[j0 + c] = x * rd[c];                        // L[i][j0 + c] (meaningful for i > j0 + c; == Ld for i in J)
            }
        }
        // column block to LDS: L below the diagonal, 1/L[j][j] on it
#pragma unroll
        for (int c = 0; c < NB; ++c)
            if (c < nb && act && l >= j0 + c) A[i * LD + j0 + c] = (l == j0 + c) ? rd[c] : row[j0 + c];
        wave_sync();
    }
    return bad;
}

// -------------------------------------------------------------------------------------
// The first NPV steps of the Cholesky factorization of a symmetric NROWS x NROWS matrix (NROWS >= NPV), one lane per
// row, NB pivots per LDS round trip -- the condensed mode's elimination of the cell unknowns.  Lane i holds the first
// NPV entries of row i in `row` and leaves L[i][0..NPV) there; the rows in LDS (row-major, stride LD) receive the same,
// with 1 / L[j][j] on the diagonal.  Per block J = j0 .. j0+nb-1 every lane forms, from the LDS rows of J,
//   D = M[J,J] - L[J,:j0] L[J,:j0]^T   (redundantly: it is the pivot block every lane needs)  and
//   t = M[i,J] - L[i,:j0] L[J,:j0]^T   (its own row),
// factors D in registers and solves its row against it: one round trip per block instead of one per pivot.
// Returns 0 or 1 + index of the first non-positive pivot.
// -------------------------------------------------------------------------------------
template <int NPV, int NROWS, int NB, int LD>
__device__ __forceinline__ int lds_partial_cholesky(double *A, int l, double (&row)[NPV])
{
    const bool act = l < NROWS;
    const int i = act ? l : 0;
    int bad = 0;
#pragma unroll
    for (int j0 = 0; j0 < NPV; j0 += NB) {
        const int nb = (NPV - j0 < NB) ? (NPV - j0) : NB;       // constant after unrolling
        double d[NB][NB], t[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            if (r < nb) {
                t[r] = row[j0 + r];
#pragma unroll
                for (int c = 0; c <= r; ++c) d[r][c] = A[(j0 + r) * LD + j0 + c];
            }
        }
#pragma unroll
        for (int k = 0; k < j0; k += 2) {                    // rows of J, two columns per 16-byte read
            const bool two = k + 1 < j0;
            double lj[NB], lj1[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                if (r < nb) {
                    if (two) { const double2 v = lds_pair(A + (j0 + r) * LD + k); lj[r] = v.x; lj1[r] = v.y; }
                    else { lj[r] = A[(j0 + r) * LD + k]; lj1[r] = 0.0; }
                }
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                if (r < nb) {
                    t[r] = __builtin_fma(-row[k], lj[r], t[r]);
                    if (two) t[r] = __builtin_fma(-row[k + 1], lj1[r], t[r]);
#pragma unroll
                    for (int c = 0; c <= r; ++c) {
                        d[r][c] = __builtin_fma(-lj[r], lj[c], d[r][c]);
                        if (two) d[r][c] = __builtin_fma(-lj1[r], lj1[c], d[r][c]);
                    }
                }
            }
        }
        double rd[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {
                double p = d[c][c];
#pragma unroll
                for (int k = 0; k < c; ++k) p = __builtin_fma(-d[c][k], d[c][k], p);
                if (!(p > 0.0) && !bad) bad = j0 + c + 1;
                rd[c] = fast_rsqrt<1>(p);
#pragma unroll
                for (int r = c + 1; r < NB; ++r) {
                    if (r < nb) {
                        double q = d[r][c];
#pragma unroll
                        for (int k = 0; k < c; ++k) q = __builtin_fma(-d[r][k], d[c][k], q);
                        d[r][c] = q * rd[c];                    // Ld[r][c]
                    }
                }
                double x = t[c];
#pragma unroll
                for (int k = 0; k < c; ++k) x = __builtin_fma(-row[j0 + k], d[c][k], x);
                row[j0 + c] = x * rd[c];                        // L[i][j0 + c] (for i in J, i > j0 + c: == Ld; i == j0 + c: L[j][j])
            }
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
            if (c < nb && act && l >= j0 + c) A[i * LD + j0 + c] = (l == j0 + c) ? rd[c] : row[j0 + c];
        wave_sync();
    }
    return bad;
}

// x <- L^-1 x (forward) and x <- L^-T x (backward); L as left by lds_cholesky (row-major).
template <int N, int LD>
__device__ __forceinline__ void lds_forward(const double *L, double (&x)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = i == 0 ? x[0] : lds_dotsub_n(x[i], L + i * LD, x, i);
        x[i] = s * L[i * LD + i];
    }
}
template <int N, int LD>
__device__ __forceinline__ void lds_backward(const double *L, double (&x)[N])
{
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[k * LD + i] * x[k];
        x[i] = s * L[i * LD + i];
    }
}

// the same with the TRUE diagonal in L and the reciprocals in rd (the image the pre-pass delivers)
template <int N, int LD>
__device__ __forceinline__ void lds_forward_rd(const double *L, const double *rd, double (&x)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s = i == 0 ? x[0] : lds_dotsub_n(x[i], L + i * LD, x, i);
        x[i] = s * rd[i];
    }
}
// The same in batches of rows: the prefixes (and reciprocals) of a batch's rows are read together, one LDS round trip per
// batch instead of one per row -- with a read next to each use the compiler recycles one register quad and every row waits
// for its own reads.  CAP = doubles of L per batch (registers: 2 CAP).
__host__ __device__ constexpr int fwd_batch_end(int b0, int n, int cap)
{
    int used = 0, r = b0;
    while (r < n && (r == b0 || used + 2 * ((r + 1) / 2) <= cap)) { used += 2 * ((r + 1) / 2); ++r; }
    return r;
}
template <int N, int LD, int CAP, int B0 = 0>
__device__ __forceinline__ void lds_forward_rd_batched(const double *L, const double *rd, double (&x)[N])
{
    typedef double v2d_ __attribute__((ext_vector_type(2)));
    constexpr int B1 = fwd_batch_end(B0, N, CAP), NB = B1 - B0, MP = (B1 + 1) / 2;
    v2d_ lr[NB][MP > 0 ? MP : 1];
    double rv[NB];
#pragma unroll
    for (int r = B0; r < B1; ++r) {
#pragma unroll
        for (int k = 0; k < r; k += 2) lr[r - B0][k / 2] = *reinterpret_cast<const v2d_ *>(L + r * LD + k);   // (k = r - 1: the pair ends on the diagonal)
        rv[r - B0] = rd[r];
    }
    // column by column: the updates of a column are independent of each other (written row by row the FMAs of a row are one
    // dependent chain after the other); every row still sums in ascending k, so the results are those of the row form
#pragma unroll
    for (int k = 0; k < B1; ++k) {
        if (k >= B0) x[k] *= rv[k - B0];
#pragma unroll
        for (int r = (k + 1 > B0 ? k + 1 : B0); r < B1; ++r)
            x[r] = __builtin_fma(-((k & 1) ? lr[r - B0][k / 2].y : lr[r - B0][k / 2].x), x[k], x[r]);
    }
    if constexpr (B1 < N) lds_forward_rd_batched<N, LD, CAP, B1>(L, rd, x);
}
// ---- the same substitutions with the packed factor in registers (Cfg::DPPFWD): element e of [packed L | 1 / diagonal] is
// held by lane e mod 16 of each row of 16 lanes in lr[e / 16]; an FMA takes it through its DPP operand (row_newbcast).
// The DPP source registers are written by LDS loads only: the 2 wait states a VALU write would need before a DPP read of
// the same register never arise (tools/spills.py checks the ISA for it).
template <int E>
__device__ __forceinline__ void dpp_fnma(double &acc, double lv, double x)     // acc -= bcast(lv, E) * x
{
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(lv), "v"(x), "n"(E));
}
template <int E>
__device__ __forceinline__ double dpp_bcast(double lv)                        // bcast(lv, E)
{
    double r;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(lv), "n"(E));
    return r;
}
template <int NLR, int E>
__device__ __forceinline__ void dpp_fnma_e(double &acc, const double (&lr)[NLR], double x) { dpp_fnma<E % 16>(acc, lr[E / 16], x); }
template <int NLR, int E>
__device__ __forceinline__ double dpp_bcast_e(const double (&lr)[NLR]) { return dpp_bcast<E % 16>(lr[E / 16]); }

// x = L^-1 x, column by column (the updates of a column are independent FMAs; every row sums in ascending k)
template <int N, int NLR, int K, int R>
__device__ __forceinline__ void dpp_forward_col(const double (&lr)[NLR], double (&x)[N])
{
    if constexpr (R < N) {
        dpp_fnma_e<NLR, R * (R + 1) / 2 + K>(x[R], lr, x[K]);
        dpp_forward_col<N, NLR, K, R + 1>(lr, x);
    }
}
template <int N, int NLR, int K = 0>
__device__ __forceinline__ void dpp_forward(const double (&lr)[NLR], double (&x)[N])
{
    constexpr int NL = N * (N + 1) / 2;
    x[K] *= dpp_bcast_e<NLR, NL + K>(lr);
    dpp_forward_col<N, NLR, K, K + 1>(lr, x);
    if constexpr (K + 1 < N) dpp_forward<N, NLR, K + 1>(lr, x);
}
// x = L^-T x: x[K] = (x[K] - sum_{k > K} L[k][K] x[k]) / L[K][K], K descending; column K of L^T is row K of L
template <int N, int NLR, int K, int I>
__device__ __forceinline__ void dpp_backward_row(const double (&lr)[NLR], double (&x)[N])
{
    if constexpr (I < K) {
        dpp_fnma_e<NLR, K * (K + 1) / 2 + I>(x[I], lr, x[K]);
        dpp_backward_row<N, NLR, K, I + 1>(lr, x);
    }
}
template <int N, int NLR, int K = N - 1>
__device__ __forceinline__ void dpp_backward(const double (&lr)[NLR], double (&x)[N])
{
    constexpr int NL = N * (N + 1) / 2;
    x[K] *= dpp_bcast_e<NLR, NL + K>(lr);
    dpp_backward_row<N, NLR, K, 0>(lr, x);
    if constexpr (K > 0) dpp_backward<N, NLR, K - 1>(lr, x);
}

// ---- condensed mode: the first NPV steps of the row-by-row Cholesky factorization of the image with the rows in registers
// (hho_local_ops_kernel, S9).  Step J: s = M[i][J] - sum_{k<J} L[i][k] L[J][k] with L[J][k] = register rA[k] of lane J of the
// row of 16 lanes (DPP), d = s of lane J, L[i][J] = s / sqrt(d).  rA[J - 1] and s were just written by the vector ALU: the
// s_nop ties give their DPP reads the two wait states they need.
template <int NPV, bool TWO, int J, int K>
__device__ __forceinline__ void cond_dpp_dot(double (&sA)[2], double (&sB)[2], const double (&rA)[NPV], const double (&rB)[TWO ? NPV : 1])
{
    if constexpr (K < J) {                                 // (two partial sums per row: the FMAs of a sum are a dependent chain)
        dpp_fnma<J>(sA[K & 1], rA[K], rA[K]);
        if (TWO) dpp_fnma<J>(sB[K & 1], rA[K], rB[K]);
        cond_dpp_dot<NPV, TWO, J, K + 1>(sA, sB, rA, rB);
    }
}
template <int NPV, bool TWO, int J = 0>
__device__ __forceinline__ void cond_dpp_chain(double (&rA)[NPV], double (&rB)[TWO ? NPV : 1], double &rdiag, int &bad, int q)
{
    double pA[2] = {rA[J], 0.0}, pB[2] = {TWO ? rB[J] : 0.0, 0.0};
    if (J > 0) asm volatile("s_nop 1" : "+v"(rA[J > 0 ? J - 1 : 0]));
    cond_dpp_dot<NPV, TWO, J, 0>(pA, pB, rA, rB);
    double sA = J > 1 ? pA[0] + pA[1] : pA[0];
    const double sB = J > 1 ? pB[0] + pB[1] : pB[0];
    asm volatile("s_nop 1" : "+v"(sA));
    const double d = dpp_bcast<J>(sA);
    if (!(d > 0.0) && !bad) bad = J + 1;
    const double r = fast_rsqrt<1>(d);
    rA[J] = sA * r;
    if (TWO) rB[J] = sB * r;
    if (q == J) rdiag = r;
    if constexpr (J + 1 < NPV) cond_dpp_chain<NPV, TWO, J + 1>(rA, rB, rdiag, bad, q);
}

// The same chain RIGHT-looking: once column J is final (s / sqrt(d)), every later column takes its term at once,
//   M[i][K] -= L[i][J] L[K][J],  K = J+1 .. NPV-1   (L[K][J] = register rA[J] of lane K of the row of 16: DPP),
// independent FMAs, so that a step's dependent path is pivot broadcast -> rsqrt -> scale -> ONE FMA into the next pivot
// column, where the left-looking form has the J-term dot product in front of the rsqrt.  Each entry still receives its
// terms in ascending J (the order of the row-by-row LLT).
template <int NPV, bool TWO, int J, int K>
__device__ __forceinline__ void cond_dpp_update(double (&rA)[NPV], double (&rB)[TWO ? NPV : 1])
{
    if constexpr (K < NPV) {
        dpp_fnma<K>(rA[K], rA[J], rA[J]);
        if (TWO) dpp_fnma<K>(rB[K], rA[J], rB[J]);
        cond_dpp_update<NPV, TWO, J, K + 1>(rA, rB);
    }
}
template <int NPV, bool TWO, int J = 0>
__device__ __forceinline__ void cond_dpp_chain_rl(double (&rA)[NPV], double (&rB)[TWO ? NPV : 1], double &rdiag, int &bad, int q)
{
    double sA = rA[J];
    asm volatile("s_nop 1" : "+v"(sA));                    // (written by the previous step's FMA: wait states of its DPP read)
    const double d = dpp_bcast<J>(sA);
    if (!(d > 0.0) && !bad) bad = J + 1;
    const double r = fast_rsqrt<1>(d);
    rA[J] = sA * r;
    if (TWO) rB[J] = rB[J] * r;
    if (q == J) rdiag = r;
    if constexpr (J + 1 < NPV) {
        asm volatile("s_nop 1" : "+v"(rA[J]));
        cond_dpp_update<NPV, TWO, J, J + 1>(rA, rB);
        cond_dpp_chain_rl<NPV, TWO, J + 1>(rA, rB, rdiag, bad, q);
    }
}

template <int N, int LD>
__device__ __forceinline__ void lds_backward_rd(const double *L, const double *rd, double (&x)[N])
{
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[k * LD + i] * x[k];
        x[i] = s * rd[i];
    }
}

// forward / backward substitution with the constant face factor L^ (uniform table reads)
template <int FBS>
__device__ __forceinline__ void face_forward(const FaceTables &ft, double (&x)[FBS])
{
#pragma unroll
    for (int i = 0; i < FBS; ++i) {
        double s = x[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= ft.lf[i][k] * x[k];
        x[i] = s * ft.lf[i][i];
    }
}

// force a wave-uniform double into scalar registers
__device__ __forceinline__ double to_sgpr(double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// lane-dependent choice among four register values (kept as scalars: an array indexed this way
// is demoted to scratch memory)
__device__ __forceinline__ double sel4(double v0, double v1, double v2, double v3, int i)
{
    double r = v0;
    r = (i == 1) ? v1 : r;
    r = (i == 2) ? v2 : r;
    r = (i == 3) ? v3 : r;
    return r;
}
__device__ __forceinline__ uint32_t sel4u(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, int i)
{
    uint32_t r = v0;
    r = (i == 1) ? v1 : r;
    r = (i == 2) ? v2 : r;
    r = (i == 3) ? v3 : r;
    return r;
}

// -------------------------------------------------------------------------------------
// The kernel.  One wavefront per block; G lanes per cell; persistent over cells.
// -------------------------------------------------------------------------------------
// MODE_LC:    only lc = data + stab is produced (one accumulator per entry);
// MODE_SPLIT: data and stab are kept apart so that any of lc / data / stab can be written
//             (two accumulator sets and the columns of Y and U in registers: at most 3 waves/SIMD);
// MODE_COND:  lc never leaves the chip: the product lands in an LDS image and the cell unknowns are eliminated there
//             (partial Cholesky of [lc f_T; f_T^T 0] over the cbs cell pivots); the packed Schur complement and the
//             condensed right-hand side are the only output -- or, given the face unknowns, the recovered cell unknowns.
enum { MODE_LC = 0, MODE_SPLIT = 1, MODE_COND = 2 };
// hho_pre.hpp: the head of one cell in the registers of the calling lane, its record to `out` (pairs 16 doubles apart)
struct PreArgs {
    const QuadTables *tab;
    const double *points;
    const uint32_t *ptids;
    size_t first, n;           // cells first .. first+n-1; record of cell first+i: tile i / 8, slot i % 8 (see the kernel)
    double *pre;
};
template <class C>
__device__ __forceinline__ void cell_pre_record(const PreArgs &a, size_t cell, double *out);
template <class C, int MODE>
__global__ __launch_bounds__(64, (MODE == MODE_SPLIT && C::WAVES > 3) ? 3 : C::WAVES) void hho_local_ops_kernel(LocalOpsArgs a)
{
    constexpr bool SPLIT = MODE == MODE_SPLIT, COND = MODE == MODE_COND;
    static_assert(COND == C::COND, "the condensed mode runs on the Cfg instance that reserves its LDS image");
    static_assert(!COND || (C::CBS + C::NF + 1 <= C::G && C::HAS_STAB), "condensed mode: one lane per row of [lc f_T; f_T^T 0]");
    constexpr int G = C::G, RBS = C::RBS, CBS = C::CBS, FBS = C::FBS, MS = C::MS, NR = C::NR, NF = C::NF;
    constexpr int NQ = C::NQ, NFQ = C::NFQ, NFP = C::NFP, NP = C::NP, RD = C::RD, NPW = C::NPW;
    constexpr int ZS = C::ZS, ND = C::ND, NRP = C::NRP, LD = C::LD;
    static_assert(MS <= G, "one lane per local-matrix column");
    static_assert(RBS <= G && NF <= G, "one lane per row in the factorizations");
    static_assert(C::NQ > 0, "empty quadrature rule (the rules[8] hole)");

#if PA_SELF_PRE
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const QuadTables *__restrict__ tab = a.tab;
    typedef typename C::Pre PRE;
    // Blocks b and b + 8 share an XCD (round-robin dispatch; a speed assumption only): give the blocks of an XCD
    // consecutive cells, so that the 8 cells of a record tile, and the lines of lc they share, meet in one L2.
    const size_t lblock = (PA_XCD_MAP && gridDim.x % 8 == 0) ? (size_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (size_t)blockIdx.x;
    const size_t stride = (size_t)gridDim.x * C::CPW;
    // Cfg::SELF_PRE: the wavefront's own ring of 64 records; pass ib of a batch reads the slots ib CPW + g
    constexpr bool SELF = C::SELF_PRE;
    constexpr int SELF_IT = C::SELF_IT;
    double *ring = SELF ? a.pre_ring + (size_t)blockIdx.x * (size_t)(64 * PRE::NPRE) : nullptr;
#ifdef PA_STAGE_CLOCK
    long long tk_sum[PA_NSTAGE] = {0}, tk_last = clock64();
#endif
    // Cfg::SELF_PRE: an outer loop over batches of SELF_IT passes -- the heads of the batch's 64 cells first (the thread-per-cell code
    // of hho_pre.hpp, which needs every register the kernel has), then the per-lane bookkeeping of the cooperative passes, re-derived
    // per batch from an opaque lane index so that nothing of it is live across the production, then the passes.  Otherwise one batch.
    size_t base = lblock * C::CPW;
    // the FIRST batch of a wavefront is shorter by a pseudo-random number of passes: the wavefronts of a compute unit then stop for their
    // production at different times, next to the others' passes (all at once they are the pre-pass again: the whole chip writing records)
    int nb = SELF ? SELF_IT - (int)(((uint32_t)blockIdx.x * 2654435761u) >> 8) % SELF_IT : 1;
    while (base < a.n) {
    int lane = threadIdx.x;
    if (SELF) {
        // lane j's cell is the one group j % CPW of pass j / CPW of the batch will work on.  Every record of the previous batch has
        // been consumed (the last one was deposited in region P a pass ago), so the ring is free.  Own stores, then own loads:
        // s_waitcnt vmcnt(0) orders them (the vector L1 is write-through and the lines belong to this compute unit alone).
        PA_MARK("SELFPRE");
        const size_t idx = base + (size_t)(lane / C::CPW) * stride + (size_t)(lane % C::CPW);
        if (idx < a.n && lane / C::CPW < nb)
            cell_pre_record<C>(PreArgs{tab, a.points, a.ptids, 0, 0, nullptr}, a.first + idx,
                               ring + ((uint32_t)lane >> 3) * (uint32_t)(PRE::NP2 * 16) + ((uint32_t)lane & 7u) * 2u);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(lane) : : "memory");
    }
    const int g = lane / G, l0 = lane % G;
    const int l = l0;
    double *S = smem + g * C::LDS_PER_CELL;
#else
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int g = lane / G, l0 = lane % G;
    const int l = l0;
    double *S = smem + g * C::LDS_PER_CELL;
    const QuadTables *__restrict__ tab = a.tab;
    typedef typename C::Pre PRE;
    // Blocks b and b + 8 share an XCD (round-robin dispatch; a speed assumption only): give the blocks of an XCD
    // consecutive cells, so that the 8 cells of a record tile, and the lines of lc they share, meet in one L2.
    const size_t lblock = (PA_XCD_MAP && gridDim.x % 8 == 0) ? (size_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (size_t)blockIdx.x;
    const size_t stride = (size_t)gridDim.x * C::CPW;
    constexpr bool SELF = false;
#ifdef PA_STAGE_CLOCK
    long long tk_sum[PA_NSTAGE] = {0}, tk_last = clock64();
#endif
#endif
    // constant face tables, kept in scalar registers
    FaceTables ft;
#pragma unroll
    for (int q = 0; q < NFQ; ++q)
#pragma unroll
        for (int k = 0; k < FBS; ++k) ft.cw[q][k] = to_sgpr(tab->face[C::FD].cw[q][k]);
#pragma unroll
    for (int i = 0; i < FBS; ++i)
#pragma unroll
        for (int k = 0; k <= i; ++k) ft.lf[i][k] = to_sgpr(tab->face[C::FD].lf[i][k]);

    constexpr bool S1_SPLIT = PA_S1_SPLIT && C::USE_PRE && 2 * NFP <= G && C::PPL == 1;
    // ---- per-lane, cell-invariant bookkeeping -------------------------------------------
    // reference coordinates of the evaluation points this lane owns
    double r0[C::PPL], r1[C::PPL], r2[C::PPL], rw[C::PPL];
#pragma unroll
    for (int r = 0; r < C::PPL; ++r) {
        const int p = l + r * G;
        r0[r] = r1[r] = r2[r] = rw[r] = 0.0;
        if (C::NQB > 0 && p < C::NQB) {
            if (C::QUAD == QUAD_TENSOR) {
                const int i = p % C::NG, j = p / C::NG;            // outer eta, inner xi  quadratures.hpp:355-357
                r0[r] = tab->gauss_x[C::NG][i];
                r1[r] = tab->gauss_x[C::NG][j];
                rw[r] = tab->gauss_w[C::NG][i] * tab->gauss_w[C::NG][j];
            } else {
                const int row = p % C::NT;
                constexpr int R = C::QDEG == 0 ? 1 : C::QDEG;      // rules[deg]  quadratures.hpp:257
                r0[r] = tab->dun[R][row][0];
                r1[r] = tab->dun[R][row][1];
                r2[r] = tab->dun[R][row][2];
                rw[r] = tab->dun[R][row][3];
            }
        } else if (p < NP || (S1_SPLIT && p < 2 * NFP)) {
            const int q = (S1_SPLIT && p >= NFP ? p - NFP : p - C::NQB) % NFQ;
            r0[r] = tab->gauss_x[NFQ][q];
            rw[r] = 0.5 * tab->gauss_w[NFQ][q];
        }
    }
    // moments owned by this lane: LDS offsets of w bx^p and by^r
    int mom_ox[C::MPL], mom_oy[C::MPL];
#pragma unroll
    for (int t = 0; t < C::MPL; ++t) {
        const int mu = l + t * G;
        int p = 0, r = 0;
        if (mu < C::NMOM) mono_exps(mu, p, r);
        mom_ox[t] = C::oWPX + p;
        mom_oy[t] = C::oPY + r;
    }
    // stiffness entries owned by this lane: stiff(i,j) = ih^2 (a a' MOM(a+a'-2, b+b') + b b' MOM(a+a', b+b'-2))
    // packed as idx1 | idx2<<8 | c1<<16 | c2<<24   (bases.hpp:170-176 with hho.hpp:57-61)
    // (i, j <= i of the packed lower triangle; the mirror entry is written with it) in bits 0-7 / 8-15 of st_ij
    uint32_t st_code[C::SPL], st_ij[C::SPL];
#pragma unroll
    for (int t = 0; t < C::SPL; ++t) {
        const int e = l + t * G;
        uint32_t code = 0, ij = 0;
        if (e < C::NSYM) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= e) ++i;
            const int j = e - i * (i + 1) / 2;
            int ai, bi, aj, bj;
            mono_exps(i, ai, bi);
            mono_exps(j, aj, bj);
            const int c1 = ai * aj, c2 = bi * bj;
            const int i1 = c1 ? mono_index(ai + aj - 2, bi + bj) : 0;
            const int i2 = c2 ? mono_index(ai + aj, bi + bj - 2) : 0;
            code = (uint32_t)i1 | ((uint32_t)i2 << 8) | ((uint32_t)c1 << 16) | ((uint32_t)c2 << 24);
            ij = (uint32_t)i | ((uint32_t)j << 8);
        }
        st_code[t] = code; st_ij[t] = ij;
    }
    // column role of this lane: cell column (c < CBS) or column kf of face fc
    const bool is_col = l < MS, is_cellcol = l < CBS;
    const int c0 = is_col ? l : 0;
    const int fc = is_cellcol ? 0 : (c0 - CBS) / FBS, kf = is_cellcol ? 0 : (c0 - CBS) % FBS;
    // t_q^kf ; column kf of L^^T.  Per lane and cell-invariant: 2 (NFQ + FBS) registers for the whole kernel -- or, where
    // the tables are small (FACE_SEL), the uniform tables in scalar registers and a select per use from the lane index
    // of the pass.
    // (measured: -4 % for the k = 2 tensor lc kernel while it ran four waves and spilled six registers at its 128; +4 % in
    // its condensed form and at k = 1, where nothing spills -- and +3 % at the three waves it runs now: off)
    constexpr bool FACE_SEL = PA_FACE_SEL && FBS == 3 && C::CD == 3 && C::QUAD == QUAD_TENSOR && G == 32 && MODE == MODE_LC;
    double fbq0[NFQ], ufc0[FBS];
    double fbt[FACE_SEL ? NFQ : 1][FACE_SEL ? FBS : 1], lftt[FACE_SEL ? FBS : 1][FACE_SEL ? FBS : 1];
#pragma unroll
    for (int q = 0; q < NFQ; ++q) fbq0[q] = FACE_SEL ? 0.0 : tab->face[C::FD].fb[q][kf];
#pragma unroll
    for (int j = 0; j < FBS; ++j) ufc0[j] = FACE_SEL ? 0.0 : tab->face[C::FD].lft[j][kf];
    if (FACE_SEL) {
#pragma unroll
        for (int q = 0; q < NFQ; ++q)
#pragma unroll
            for (int k = 0; k < FBS; ++k) fbt[q][k] = tab->face[C::FD].fb[q][k];
#pragma unroll
        for (int j = 0; j < FBS; ++j)
#pragma unroll
            for (int k = 0; k < FBS; ++k) lftt[j][k] = tab->face[C::FD].lft[j][k];
    }

    // Corner tile of Z^T Z on the vector pipe (msize 17..24): with T_F = [trace_F | 0] its columns are face columns, whose
    // U parts are -sqrt(|F|/2h) L^^T e_k on the rows of their own face: the U term of an entry is su_F^2 (L^^ L^^T)[ki][kj]
    // for two columns of one face and nothing otherwise -- a constant per lane; only the Y rows are read from LDS.
    constexpr bool CORNER_SHORT = C::CORNER_VALU && PA_CORNER_SHORT && C::HAS_STAB && !C::GENERAL_FANCY && PA_UNIT_U && MODE != MODE_SPLIT &&
                                  CBS <= 16 && !PA_LC_VALU;
    double cornerT = 0.0;
    if (CORNER_SHORT && l < C::NCORNER * (C::NCORNER + 1) / 2) {
        int cj_ = 0;
        while ((cj_ + 1) * (cj_ + 2) / 2 <= l) ++cj_;
        const int ci_ = l - cj_ * (cj_ + 1) / 2;
        const int fi_ = (16 + ci_ - CBS) / FBS, ki_ = (16 + ci_ - CBS) % FBS, fj_ = (16 + cj_ - CBS) / FBS, kj_ = (16 + cj_ - CBS) % FBS;
        if (fi_ == fj_) {
#pragma unroll
            for (int j = 0; j < FBS; ++j) cornerT += tab->face[C::FD].lft[j][ki_] * tab->face[C::FD].lft[j][kj_];
        }
    }

    // Per-lane index constants of the unit form of U (S4b / S6) and of the corner tile, PACKED one word each: kept across
    // the cell loop in one register and unpacked per pass (a field extract each) -- recomputed from the opaque lane index
    // they cost ten to seventeen integer instructions per use and pass, held unpacked a register each for the whole kernel.
    //   unit word:   bits 0-11 offset of the unit's first phi value in the face-point table (doubles from oPHF),
    //                bits 12-23 offset of its first U entry in Z (doubles from S; the sink for lanes without a unit), bits 24-25 face
    //   corner word: bits 0-7 ci, 8-15 cj (ci <= cj: the pair of corner columns of this lane), 16-17 face of column ci
    constexpr bool UNIT_U = C::HAS_STAB && !C::GENERAL_FANCY && !SPLIT && PA_UNIT_U;
    constexpr bool UPERM_ON = C::UPERM && UNIT_U && !PA_LC_VALU;
    constexpr int NU = 4 * CBS, UR = UNIT_U ? cdiv(NU, G) : 1;
    uint32_t unit_pk[UR];
#pragma unroll
    for (int r = 0; r < UR; ++r) {
        const int u = l0 + r * G;
        const bool on = u < NU;
        const int uu = on ? u : 0;
        const int f = uu / CBS, cc = uu - f * CBS;
        const int ur = (UPERM_ON ? ((f + 4 - C::UF1) & 3) : f) * FBS;           // row block of face f (see Cfg::UPERM)
        const int zoff = on ? C::oZ + NRP + ur + cc * ZS : C::oDUMMY;            // (the sink holds 4 doubles)
        static_assert(!UNIT_U || (NFP * RBS < 4096 && C::oDUMMY < 4096 && ZS * MS < 4096), "packed unit offsets: 12 bits each");
        unit_pk[r] = (uint32_t)(f * NFQ * RBS + cc) | ((uint32_t)zoff << 12) | ((uint32_t)f << 24);
    }
    uint32_t corner_pk = 0;
    if (C::CORNER_VALU) {
        int cj_ = 0;
#pragma unroll
        for (int j = 1; j < C::NCORNER; ++j) cj_ += l0 >= j * (j + 1) / 2 ? 1 : 0;
        const int ci_ = l0 - cj_ * (cj_ + 1) / 2;                                // ci_ <= cj_
        const int cf_ = (l0 < C::NCORNER * (C::NCORNER + 1) / 2) ? (16 + ci_ - CBS) / FBS : 0;
        corner_pk = (uint32_t)(ci_ & 0xff) | ((uint32_t)cj_ << 8) | ((uint32_t)(cf_ & 3) << 16);
    }

    // Tile (1,1) of Z^T Z where it runs on the matrix pipe (msize 25..32, no vector-pipe corner) with T_F = [trace_F | 0]: its columns
    // are face columns, whose U parts are -sqrt(|F|/2h) L^^T e_k on the rows of their own face, so the U term of an entry is
    // su_F^2 (L^^ L^^T)[ki][kj] for two columns of one face and nothing otherwise: a constant per accumulator times the face's
    // scale, added after the products -- which then stop after the rows of Y (k = 3: 4 of 8 k-steps, 20 instead of 24 matrix
    // instructions per cell).
    constexpr int NTL_ = (MS + 15) / 16;
    constexpr bool T11_SHORT = PA_T11_SHORT && UNIT_U && NTL_ == 2 && !C::CORNER_VALU && CBS <= 16 && !PA_LC_VALU && !SPLIT;
    constexpr bool ACC_PERM_ = PA_ACC_PERM && !(C::DIRECT_STORE && !COND) && !COND && MS % 2 == 0;
    double t11c[4] = {0.0, 0.0, 0.0, 0.0};
    int t11f = 0;
    if (T11_SHORT) {
        const int kk_ = lane >> 4, jj_ = lane & 15;
        const int col = 16 + (ACC_PERM_ ? 4 * (jj_ & 3) + (jj_ >> 2) : jj_);
        if (col < MS) {
            t11f = (col - CBS) / FBS;
            const int kc = (col - CBS) % FBS;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 + (ACC_PERM_ ? 4 * kk_ + r : kk_ + 4 * r);
                if (row < MS && (row - CBS) / FBS == t11f) {
                    const int kr = (row - CBS) % FBS;
#pragma unroll
                    for (int j = 0; j < FBS; ++j) t11c[r] += tab->face[C::FD].lft[j][kr] * tab->face[C::FD].lft[j][kc];
                }
            }
        }
    }

    // pairs of the cell's pre-pass record this lane moves to LDS, and where their two doubles go
    int pre_dst[C::PLC][2];
#pragma unroll
    for (int t = 0; t < C::PLC; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 2 * (l + t * G) + h;
            int dst = C::oDUMMY + h;
            if (C::DPPFWD) {
                if (e < PRE::NPRE) dst = C::oREC + e;            // the record as it comes
            } else if (e < PRE::NL) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= e) ++i;
                dst = C::oLG + i * LD + (e - i * (i + 1) / 2);
            } else if (e < PRE::oMR) {
                dst = C::oLIN + (e - PRE::NL);
            } else if (e < PRE::oMC) {
                dst = C::oMRl + (e - PRE::oMR);
            } else if (e < PRE::oMCR) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= e - PRE::oMC) ++i;
                dst = C::oMCl + i * C::LDM + (e - PRE::oMC - i * (i + 1) / 2);
            } else if (e < PRE::oMCR + (C::GENERAL_FANCY ? CBS : 0)) {
                dst = C::oMCRl + (e - PRE::oMCR);
            }
            pre_dst[t][h] = dst;
        }
    }

    // The record of a cell is fetched one cell ahead: the loads are issued at the top of a cell's pass, land in
    // registers while S1-S6 run, and go to region P when S6 is done with it -- before the stores of S8 are
    // issued, so that no load ever queues behind them (vector memory operations complete in order).
    double2 rec[C::PLC];
#if PA_SELF_PRE
    auto rec_issue_slot = [&](int ib) {
        const uint32_t s = (uint32_t)(ib * C::CPW + g);
        const double *pc = ring + (s >> 3) * (uint32_t)(PRE::NP2 * 16) + (s & 7u) * 2u;
#pragma unroll
        for (int t = 0; t < C::PLC; ++t) {
            const int e2 = l0 + t * G;
            rec[t] = *reinterpret_cast<const double2 *>(pc + 16 * (e2 < PRE::NP2 ? e2 : 0));
        }
    };
#else
    auto rec_issue_slot = [&](int) {};
#endif
    auto rec_issue = [&](size_t b) {
        // records lie in tiles of 8 cells, [tile][pair][cell % 8] (hho_pre.hpp): the pairs of one record are 128 bytes apart
        // (b is a multiple of the CPW cells of a wavefront, CPW divides 8: the wavefront's cells share a tile, whose address is
        // wave-uniform -- scalar arithmetic; a lane adds its slot.  Past the end the lanes read records of the first tile, or
        // the unwritten slots of the last one -- the buffer holds whole tiles --, and nothing they compute is stored.)
        static_assert(8 % C::CPW == 0, "the cells of a wavefront share a record tile");
        const size_t bu = b < a.n ? b : 0;
        const double *pc = a.pre + (bu >> 3) * (size_t)(PRE::NP2 * 16) + (((uint32_t)bu & 7u) + (uint32_t)g) * 2u;
#pragma unroll
        for (int t = 0; t < C::PLC; ++t) {
            const int e2 = l0 + t * G;
            rec[t] = *reinterpret_cast<const double2 *>(pc + 16 * (e2 < PRE::NP2 ? e2 : 0));
        }
    };
    auto rec_deposit = [&]() {
#pragma unroll
        for (int t = 0; t < C::PLC; ++t) {
            if (C::DPPFWD) {
                typedef double v2d_ __attribute__((ext_vector_type(2)));
                *reinterpret_cast<v2d_ *>(S + pre_dst[t][0]) = v2d_{rec[t].x, rec[t].y};      // (16-byte aligned: oREC, oDUMMY and 2 e are even)
            } else {
                S[pre_dst[t][0]] = rec[t].x;
                S[pre_dst[t][1]] = rec[t].y;
            }
        }
    };
    if (C::USE_PRE) {
        if (!C::DPPFWD) {
            for (int e = l; e < C::LGR * LD; e += G) S[C::oLG + e] = 0.0;
            wave_sync();
        }
        if (!SELF) rec_issue(lblock * C::CPW);
        else rec_issue_slot(0);
        rec_deposit();
        wave_sync();
    }

#if PA_SELF_PRE
    const int nbc = nb;
    nb = SELF_IT;
    for (int ib = 0; (!SELF || ib < nbc) && base < a.n; base += stride, ++ib) {
#else
    constexpr int ib = 0, nbc = 1;
    for (size_t base = lblock * C::CPW; base < a.n; base += stride) {
#endif
        // Re-derive the lane index opaquely per cell: otherwise LICM hoists the index computations
        // of every stage out of the cell loop and the kernel spills.
        int l = l0;
        asm volatile("" : "+v"(l));
        if (PA_PRIO_OUT >= 0) __builtin_amdgcn_s_setprio(PA_PRIO_HEAD);
#if PA_LANE_RANGE
        l &= G - 1;      // tell the compiler the range again: its index products fit 24 bits (v_mul_u32_u24 / v_mad_u32_u24 are full rate)
#endif
        const bool valid = base + g < a.n;
        const size_t cell = a.first + (valid ? base + g : a.n - 1);
        // offset of the lane's cell in an output / input array of X doubles per cell: the block's part is wave-uniform (scalar
        // arithmetic), the lane adds its group's (lanes past the end: the first cell of the pass, which exists)
        const uint32_t gl = valid ? (uint32_t)g : 0u;
        auto rel = [&](uint32_t X) -> size_t { return base * (size_t)X + (size_t)(gl * X); };
        double fbq[NFQ], ufc[FBS];
        if (FACE_SEL) {
            const int kfl = (l >= CBS && l < MS) ? (l - CBS) % FBS : 0;
#pragma unroll
            for (int q = 0; q < NFQ; ++q) fbq[q] = kfl == 0 ? fbt[q][0] : (kfl == 1 || FBS < 3) ? fbt[q][FBS > 1 ? 1 : 0] : fbt[q][FBS > 2 ? 2 : 0];
#pragma unroll
            for (int j = 0; j < FBS; ++j) ufc[j] = kfl == 0 ? lftt[j][0] : (kfl == 1 || FBS < 3) ? lftt[j][FBS > 1 ? 1 : 0] : lftt[j][FBS > 2 ? 2 : 0];
        } else {
#pragma unroll
            for (int q = 0; q < NFQ; ++q) fbq[q] = fbq0[q];
#pragma unroll
            for (int j = 0; j < FBS; ++j) ufc[j] = ufc0[j];
        }

        // ================= S0: geometry (every lane of the group, registers) ==========
        PA_MARK("S0");
        uint4 idv = {0u, 0u, 0u, 0u};
        double px0 = 0.0, py0 = 0.0, px1 = 0.0, py1 = 0.0, px2 = 0.0, py2 = 0.0, px3 = 0.0, py3 = 0.0;
        double barx, bary, ih;
        double e0x = 0.0, e0y = 0.0, e1x = 0.0, e1y = 0.0, e2x = 0.0, e2y = 0.0, e3x = 0.0, e3y = 0.0;
        int bad_pre = 0;
        if (C::USE_PRE) {
            // the per-cell head comes from the pre-pass; its record is in region P already (prefetched)
            if (!SELF) rec_issue(base + stride);
            else if (ib + 1 < nbc) rec_issue_slot(ib + 1);
            const double2 bb = lds_pair(S + C::oSU + 4), ib = lds_pair(S + C::oSU + 6);
            barx = bb.x; bary = bb.y; ih = ib.x; bad_pre = (int)ib.y;
        } else {
        idv = *reinterpret_cast<const uint4 *>(a.ptids + 4 * cell);
        {
            const double2 q0 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.x);
            const double2 q1 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.y);
            const double2 q2 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.z);
            const double2 q3 = *reinterpret_cast<const double2 *>(a.points + 2 * (size_t)idv.w);
            px0 = q0.x; py0 = q0.y; px1 = q1.x; py1 = q1.y; px2 = q2.x; py2 = q2.y; px3 = q3.x; py3 = q3.y;
        }
        {                                           // barycenter  basic_geom.hpp:247-270
            const double ax = px1 - px0, ay = py1 - py0, bx = px2 - px0, by = py2 - py0;
            const double cx = px3 - px0, cy = py3 - py0;
            const double d1 = (ax * by - ay * bx) * 0.5, d2 = (bx * cy - by * cx) * 0.5;
            const double rx = (ax + bx) * d1 + (bx + cx) * d2, ry = (ay + by) * d1 + (by + cy) * d2;
            const double iden = fast_rcp((d1 + d2) * 3);
            barx = px0 + rx * iden; bary = py0 + ry * iden;
        }
        // edge vectors in cell (CCW) order, squared lengths of edges and diagonals
        e0x = px1 - px0; e0y = py1 - py0; e1x = px2 - px1; e1y = py2 - py1;
        e2x = px3 - px2; e2y = py3 - py2; e3x = px0 - px3; e3y = py0 - py3;
        const double s0 = e0x * e0x + e0y * e0y, s1 = e1x * e1x + e1y * e1y;
        const double s2 = e2x * e2x + e2y * e2y, s3 = e3x * e3x + e3y * e3y;
        double h2;                                  // diameter^2  basic_geom.hpp:288-305
        {
            const double d02x = px2 - px0, d02y = py2 - py0, d13x = px3 - px1, d13y = py3 - py1;
            h2 = fmax(fmax(s0, s1), fmax(s2, s3));
            h2 = fmax(h2, fmax(d02x * d02x + d02y * d02y, d13x * d13x + d13y * d13y));
        }
        const double rh = fast_rsqrt(h2);           // 1 / h_T
        ih = 2.0 * rh;                              // bx = (x - bar)/(h/2), ih = 2/h  bases.hpp:98-99,142
        double hinv = rh;                           // fancy: h = cell diameter  hho.hpp:201
        if (C::NAIVE) {                             // naive: h = cell area      hho.hpp:119, basic_geom.hpp:317-334
            const double ux = px1 - px0, uy = py1 - py0, vx = px2 - px0, vy = py2 - py0;
            const double wx = px3 - px0, wy = py3 - py0;
            hinv = fast_rcp(fabs(ux * vy - uy * vx) * 0.5 + fabs(vx * wy - vy * wx) * 0.5);
        }
        // lanes 0..3: sqrt(|F| / (2 h)) of local face l, shared through LDS
        if (C::HAS_STAB && l < 4) {
            const double sq = sel4(s0, s1, s2, s3, l);
            const double len = sq * fast_rsqrt(sq);
            S[C::oSU + l] = fast_sqrt(0.5 * len * hinv);
        }
        }

        PA_TICK(0);
        // ================= S1: evaluation points ======================================
        PA_MARK("S1");
        if (S1_SPLIT) {
            // face points only (the cell points belong to the pre-pass): lane p < NFP forms phi at point p, lane NFP + p the
            // weighted normal derivatives there; both rows leave through the same store instructions
            if (!(a.ablate & 1u) && l < 2 * NFP) {
                const bool dnrow = l >= NFP;
                const int pf = dnrow ? l - NFP : l;
                const int f = pf / NFQ;
                const double *sc = S + C::oSU;
                const double2 pa_ = lds_pair(sc + 8 + 2 * f), pb_ = lds_pair(sc + 8 + 2 * ((f + 1) & 3));
                const bool descending = (((int)sc[16]) >> f) & 1;                  // the face's first vertex has the higher point id
                const double wnx = rw[0] * (pb_.y - pa_.y), wny = -rw[0] * (pb_.x - pa_.x);      // (w_q |F|/2) n: the edge length cancels
                const double t = descending ? -r0[0] : r0[0];                      // bases.hpp:260-261
                const double x = 0.5 * (1 - t) * pa_.x + 0.5 * (1 + t) * pb_.x;    // quadratures.hpp:420-428
                const double y = 0.5 * (1 - t) * pa_.y + 0.5 * (1 + t) * pb_.y;
                const double bx_ = (x - barx) * ih, by_ = (y - bary) * ih;
                double pwx[RD + 1], pwy[RD + 1];
                pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
                for (int e = 1; e <= RD; ++e) { pwx[e] = pwx[e - 1] * bx_; pwy[e] = pwy[e - 1] * by_; }
                double out[RBS];
                // (an empty volatile asm in each arm: a real branch on the execution mask -- if-converted, the arms' results meet in two
                // selects per double)
                if (!dnrow) {
                    asm volatile("");
                    int m = 0;
#pragma unroll
                    for (int kk = 0; kk <= RD; ++kk)
#pragma unroll
                        for (int ii = 0; ii <= kk; ++ii, ++m) out[m] = pwx[kk - ii] * pwy[ii];      // (px,py) = (k-i, i)  bases.hpp:119-120
                } else {
                    asm volatile("");
                    const double gnx = ih * wnx, gny = ih * wny;
                    int m = 0;
#pragma unroll
                    for (int kk = 0; kk <= RD; ++kk)
#pragma unroll
                        for (int ii = 0; ii <= kk; ++ii, ++m) {
                            const int ex_ = kk - ii, ey_ = ii;
                            if (m > 0) {
                                const double gx = ex_ == 0 ? 0.0 : (ex_ * gnx) * pwx[ex_ > 0 ? ex_ - 1 : 0] * pwy[ey_];
                                const double gy = ey_ == 0 ? 0.0 : (ey_ * gny) * pwx[ex_] * pwy[ey_ > 0 ? ey_ - 1 : 0];
                                out[m - 1] = gx + gy;
                            }
                        }
                    out[RBS - 1] = 0.0;
                }
                double *dst = S + (dnrow ? C::oDN + pf * NRP : C::oPHF + pf * RBS);
#pragma unroll
                for (int m = 0; m < RBS; ++m)
                    if (m < NR || !dnrow) dst[m] = out[m];
            }
        } else
        if (!(a.ablate & 1u)) {   // S1
#pragma unroll
        for (int r = 0; r < C::PPL; ++r) {
            const int p = l + r * G;
            if (p < NP) {
                double x, y, w, wnx = 0.0, wny = 0.0;
                const bool is_cell = C::NQB > 0 && p < C::NQB;      // (the opaque lane index has no known sign)
                if (is_cell) {
                    if (C::QUAD == QUAD_TENSOR) {
                        // bilinear map and |det J| (quadratures.hpp:331-352) with the shape functions
                        // 0.25 (1 +- xi)(1 +- eta) factored once
                        double xi = r0[r], eta = r1[r];
                        asm volatile("" : "+v"(xi), "+v"(eta));      // keep the shape functions out of LICM's reach
                        const double am = 0.25 * (1 - eta), ap = 0.25 * (1 + eta);
                        const double bm = 0.25 * (1 - xi), bp = 0.25 * (1 + xi);
                        const double n0 = (1 - xi) * am, n1 = (1 + xi) * am, n2 = (1 + xi) * ap, n3 = (1 - xi) * ap;
                        x = n0 * px0 + n1 * px1 + n2 * px2 + n3 * px3;
                        y = n0 * py0 + n1 * py1 + n2 * py2 + n3 * py3;
                        const double j11 = e0x * am - e2x * ap, j12 = e0y * am - e2y * ap;
                        const double j21 = e1x * bp - e3x * bm, j22 = e1y * bp - e3y * bm;
                        w = rw[r] * fabs(j11 * j22 - j12 * j21);
                    } else {
                        const int t = p / C::NT;                   // fan triangle (p_t, p_{t+1}, bar)  quadratures.hpp:390-396
                        const double ax = sel4(px0, px1, px2, px3, t), ay = sel4(py0, py1, py2, py3, t);
                        const double bx = sel4(px1, px2, px3, px0, t), by = sel4(py1, py2, py3, py0, t);
                        const double v0x = bx - ax, v0y = by - ay, v1x = barx - ax, v1y = bary - ay;
                        const double tarea = fabs((v0x * v1y - v0y * v1x) * 0.5);      // quadratures.hpp:248-251
                        x = ax * r0[r] + bx * r1[r] + barx * r2[r];
                        y = ay * r0[r] + by * r1[r] + bary * r2[r];
                        w = tarea * rw[r];
                    }
                } else {
                    const int f = (p - C::NQB) / NFQ;
                    double ax, ay, bx, by;
                    bool descending;                               // the face's first vertex has the higher point id
                    if (C::USE_PRE) {
                        const double *sc = S + C::oSU;
                        const double2 pa_ = lds_pair(sc + 8 + 2 * f), pb_ = lds_pair(sc + 8 + 2 * ((f + 1) & 3));
                        ax = pa_.x; ay = pa_.y; bx = pb_.x; by = pb_.y;
                        descending = (((int)sc[16]) >> f) & 1;
                    } else {
                        ax = sel4(px0, px1, px2, px3, f); ay = sel4(py0, py1, py2, py3, f);
                        bx = sel4(px1, px2, px3, px0, f); by = sel4(py1, py2, py3, py0, f);
                        const uint32_t ia = sel4u(idv.x, idv.y, idv.z, idv.w, f), ib = sel4u(idv.y, idv.z, idv.w, idv.x, f);
                        descending = ia > ib;
                    }
                    // w n = (w_q |F|/2) (e_y, -e_x)/|F|: the edge length cancels  (basic_geom.hpp:361-369,
                    // quadratures.hpp:426); rw holds w_q / 2
                    wnx = rw[r] * (by - ay); wny = -rw[r] * (bx - ax);
                    // the face runs from its LOWER-id endpoint (basic_geom.hpp:202-203, bases.hpp:260-261):
                    // its q-th point sits at -t_q in cell order when the ids are descending
                    const double t = descending ? -r0[r] : r0[r];
                    x = 0.5 * (1 - t) * ax + 0.5 * (1 + t) * bx;   // quadratures.hpp:420-428
                    y = 0.5 * (1 - t) * ay + 0.5 * (1 + t) * by;
                    w = 0.0;
                }
                const double bx_ = (x - barx) * ih, by_ = (y - bary) * ih;
                if (is_cell) {
                    // w * bx^e and by^e, e = 0..2 recdeg: the factors of every cell moment
                    double vx = w, vy = 1.0;
#pragma unroll
                    for (int e = 0; e < NPW; ++e) {
                        S[C::oWPX + p * NPW + e] = vx;
                        S[C::oPY + p * NPW + e] = vy;
                        vx *= bx_; vy *= by_;
                    }
                } else {
                    // scaled monomials and weighted normal derivatives at a face point  bases.hpp:93-184, hho.hpp:77-83
                    const int pf = p - C::NQB;
                    double pwx[RD + 1], pwy[RD + 1];
                    pwx[0] = 1.0; pwy[0] = 1.0;
#pragma unroll
                    for (int e = 1; e <= RD; ++e) { pwx[e] = pwx[e - 1] * bx_; pwy[e] = pwy[e - 1] * by_; }
                    const double gnx = ih * wnx, gny = ih * wny;
                    int m = 0;
#pragma unroll
                    for (int kk = 0; kk <= RD; ++kk) {
#pragma unroll
                        for (int ii = 0; ii <= kk; ++ii, ++m) {
                            const int ex_ = kk - ii, ey_ = ii;      // (px,py) = (k-i, i)  bases.hpp:119-120
                            S[C::oPHF + pf * RBS + m] = pwx[ex_] * pwy[ey_];
                            if (m > 0) {
                                const double gx = ex_ == 0 ? 0.0 : (ex_ * gnx) * pwx[ex_ > 0 ? ex_ - 1 : 0] * pwy[ey_];
                                const double gy = ey_ == 0 ? 0.0 : (ey_ * gny) * pwx[ex_] * pwy[ey_ > 0 ? ey_ - 1 : 0];
                                S[C::oDN + pf * NRP + (m - 1)] = gx + gy;
                            }
                        }
                    }
                }
            }
        }
        }
        wave_sync();

        // ================= S2: cell moments ===========================================
        PA_MARK("S2");
        if (!C::USE_PRE) {
        if (!(a.ablate & 2u)) {   // S2
#pragma unroll
        for (int t = 0; t < C::MPL; ++t) {
            if (l + t * G < C::NMOM) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) s += S[mom_ox[t] + q * NPW] * S[mom_oy[t] + q * NPW];
                S[C::oMOM + l + t * G] = s;
            }
        }
        }
        wave_sync();
        }

        // ================= S3: stiffness (+mass) from moments ========================
        PA_MARK("S3");
        if (!C::USE_PRE) {
            const double ih2 = ih * ih;
#pragma unroll
            for (int t = 0; t < C::SPL; ++t) {
                const int e = l + t * G;
                if (e < C::NSYM) {
                    // decoded here, per cell: hoisted, the decoded offsets and coefficients would cost
                    // six registers per entry for the whole kernel
                    uint32_t code = st_code[t], ij = st_ij[t];
                    asm volatile("" : "+v"(code), "+v"(ij));
                    const double c1 = (double)((code >> 16) & 0xff), c2 = (double)(code >> 24);
                    const double v = ih2 * (c1 * S[C::oMOM + (code & 0xff)] + c2 * S[C::oMOM + ((code >> 8) & 0xff)]);
                    const int i = (int)(ij & 0xff), j = (int)(ij >> 8);
                    S[C::oST + i + j * LD] = v;
                    S[C::oST + j + i * LD] = v;
                }
            }
            if (C::GENERAL_FANCY) {
                for (int e = l; e < RBS * RBS; e += G) {
                    int ai, bi, aj, bj;
                    mono_exps(e % RBS, ai, bi);
                    mono_exps(e / RBS, aj, bj);
                    S[C::oMA + (e % RBS) + (e / RBS) * LD] = S[C::oMOM + mono_index(ai + aj, bi + bj)];
                }
            }
            wave_sync();
        }

        PA_TICK(1);
        // ================= S3b: column c of gr_rhs  hho.hpp:64-85 =====================
        if (PA_PRIO_OUT >= 0 && PA_PRIO_S3B >= 0) __builtin_amdgcn_s_setprio(PA_PRIO_S3B >= 0 ? PA_PRIO_S3B : 0);
        PA_MARK("S3b");
        const int c = l < MS ? l : 0;
        double col[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) asm volatile("" : "=v"(col[i]));      // (every lane writes every row below; only a profiling build's skipped stage leaves them as they are -- a constant here cost a move per row and pass)
        // Cell columns: stiff[1:, c] minus sum_pf (w dphi.n)[pf][:] phi_c(x_pf).  The NR rows of a column
        // are split over SPC lanes (there are only CBS cell columns for G lanes); the pieces meet in
        // the GRC scratch (on the dead quadrature tables) and the column owner reloads them after
        // the barrier below.
        constexpr int SPC = C::SPC, RPP = C::RPP;
        // With the pre-pass the cell columns start from zero: -F = -DN^T PHF (NR x CBS, inner dimension the NFP face
        // points) is one 16 x 16 tile of the matrix pipe per cell, NFP/4 instructions, each lane reading ONE element of
        // either table per step (the vector form reads 4 per 3 FMAs, and the LDS pipeline is the busiest of the kernel).
        // (measured: -5 % at k = 3; nothing at k = 2, where it costs the 4th wave its last registers)
        constexpr bool GRC_MFMA = C::USE_PRE && SPC > 1 && PA_GRC_MFMA && NR >= PA_GRC_MFMA_MIN_NR && NR <= 16 && CBS <= 16;
        if (a.ablate & 4u) {
        } else if (GRC_MFMA) {
            typedef double v4d_ __attribute__((ext_vector_type(4)));
            const int kk_ = lane >> 4, jj_ = lane & 15;
#pragma unroll
            for (int gi = 0; gi < C::CPW; ++gi) {
                double *Sg = smem + gi * C::LDS_PER_CELL;
                v4d_ f = v4d_{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < NFP / 4; ++ks) {
                    const int pf = 4 * ks + kk_;
                    const double av = Sg[C::oDN + pf * NRP + (jj_ < NR ? jj_ : 0)];
                    const double bv = Sg[C::oPHF + pf * RBS + (jj_ < CBS ? jj_ : 0)];
                    f = __builtin_amdgcn_mfma_f64_16x16x4f64(jj_ < NR ? av : 0.0, jj_ < CBS ? bv : 0.0, f, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = kk_ + 4 * r;
                    if (4 * r < NR && (4 * r + 3 < NR || row < NR) && jj_ < CBS) Sg[C::oGRC + jj_ * NRP + row] = -f[r];
                }
            }
        } else if (SPC > 1) {
            if (l < SPC * CBS) {
                const int part = l / CBS, cc = l % CBS, r0 = part * RPP;
                double acc[RPP];
#pragma unroll
                for (int r = 0; r < RPP; ++r)
                    acc[r] = C::USE_PRE ? 0.0 : S[C::oST + 1 + (r0 + r < NR ? r0 + r : NR - 1) + cc * LD];
#pragma unroll
                for (int pf = 0; pf < NFP; ++pf) {
                    const double ph = S[C::oPHF + pf * RBS + cc];
                    const double *dn = S + C::oDN + pf * NRP + r0;
#pragma unroll
                    for (int r = 0; r < RPP; ++r) acc[r] = __builtin_fma(-dn[r], ph, acc[r]);      // rows >= NR: unused
                }
#pragma unroll
                for (int r = 0; r < RPP; ++r)
                    if (r0 + r < NR) S[C::oGRC + cc * NRP + r0 + r] = acc[r];
            }
        } else if (C::LAPG) {
            if (l < CBS) {
                // -int_T phi_c lap(phi_i) = -(2/h)^2 (a (a-1) MOM(a-2+a', b+b') + b (b-1) MOM(a+a', b-2+b')) for phi_i = bx^a by^b,
                // phi_c = bx^a' by^b'.  In the graded ordering the moment of phi_c times the monomial m of degree d sits at
                // c + d (a' + b') + m: one LDS read per monomial of degree <= recdeg - 2, the rest is in registers.
                constexpr int NDM = RD >= 2 ? P2(RD - 2) : 1;
                double mc[NDM];
                int kcv = 0;                                 // a' + b' (per cell from the opaque lane index: no register held)
#pragma unroll
                for (int k = 1; k <= C::CD; ++k) kcv += l >= k * (k + 1) / 2 ? 1 : 0;
                const double *mg = S + C::oMG + l;
                mc[0] = 0.0;
#pragma unroll
                for (int d = 0; d + 2 <= RD; ++d)
#pragma unroll
                    for (int bm = 0; bm <= d; ++bm) mc[d * (d + 1) / 2 + bm] = mg[d * kcv + d * (d + 1) / 2 + bm];
                const double nih2 = -(ih * ih);
#pragma unroll
                for (int ki = 1; ki <= RD; ++ki)
#pragma unroll
                    for (int ri = 0; ri <= ki; ++ri) {
                        const int ai = ki - ri, bi = ri, row = ki * (ki + 1) / 2 + ri - 1;
                        double v = 0.0;
                        if (ai >= 2) v = (double)(ai * (ai - 1)) * mc[(ki - 2) * (ki - 1) / 2 + bi];
                        if (bi >= 2) v = __builtin_fma((double)(bi * (bi - 1)), mc[(ki - 2) * (ki - 1) / 2 + bi - 2], v);
                        col[row] = nih2 * v;
                    }
            }
        } else if (l < CBS) {
            if (C::USE_PRE) {
#pragma unroll
                for (int i = 0; i < NR; ++i) col[i] = 0.0;
            } else {
                const double *stc = S + C::oST + 1 + c * LD;
#pragma unroll
                for (int i = 0; i + 1 < NR; i += 2) {
                    const double2 v = lds_pair(stc + i);
                    col[i] = v.x; col[i + 1] = v.y;
                }
                if (NR & 1) col[NR - 1] = stc[NR - 1];
            }
#pragma unroll
            for (int pf = 0; pf < NFP; ++pf) {
                const double ph = S[C::oPHF + pf * RBS + c];
                const double *dn = S + C::oDN + pf * NRP;
#pragma unroll
                for (int i = 0; i + 1 < NR; i += 2) {
                    const double2 v = lds_pair(dn + i);
                    col[i] = __builtin_fma(-v.x, ph, col[i]);
                    col[i + 1] = __builtin_fma(-v.y, ph, col[i + 1]);
                }
                if (NR & 1) col[NR - 1] = __builtin_fma(-dn[NR - 1], ph, col[NR - 1]);
            }
        }
        if (a.ablate & 4u) {
        } else if (l < CBS) {
        } else {
            if (PA_S3B_BATCH) {
                // the rows of the face's table in batches of QG points (<= 32 doubles), one LDS round trip each, then the products
                typedef double v2d_ __attribute__((ext_vector_type(2)));
                constexpr int QG = imax(1, imin(NFQ, 32 / NRP));
#pragma unroll
                for (int i = 0; i < NR; ++i) col[i] = 0.0;
#pragma unroll
                for (int q0 = 0; q0 < NFQ; q0 += QG) {
                    v2d_ dnv[QG][NRP / 2];
#pragma unroll
                    for (int q = q0; q < q0 + QG; ++q)
#pragma unroll
                        for (int i = 0; i < NRP; i += 2)
                            if (q < NFQ) dnv[q - q0][i / 2] = *reinterpret_cast<const v2d_ *>(S + C::oDN + (fc * NFQ + q) * NRP + i);
#pragma unroll
                    for (int i = 0; i < NR; ++i)
#pragma unroll
                        for (int q = q0; q < q0 + QG; ++q)
                            if (q < NFQ) col[i] = __builtin_fma((i & 1) ? dnv[q - q0][i / 2].y : dnv[q - q0][i / 2].x, fbq[q], col[i]);
                }
            } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) col[i] = 0.0;
#pragma unroll
            for (int q = 0; q < NFQ; ++q) {
                const double *dn = S + C::oDN + (fc * NFQ + q) * NRP;
#pragma unroll
                for (int i = 0; i + 1 < NR; i += 2) {
                    const double2 v = lds_pair(dn + i);
                    col[i] = __builtin_fma(v.x, fbq[q], col[i]);
                    col[i + 1] = __builtin_fma(v.y, fbq[q], col[i + 1]);
                }
                if (NR & 1) col[NR - 1] = __builtin_fma(dn[NR - 1], fbq[q], col[NR - 1]);
            }
            }
        }
        wave_sync();
        if (SPC > 1 && l < CBS && !(a.ablate & 4u)) {
            const double *gc = S + C::oGRC + c * NRP;
#pragma unroll
            for (int i = 0; i + 1 < NR; i += 2) {
                const double2 v = lds_pair(gc + i);
                col[i] = v.x; col[i + 1] = v.y;
            }
            if (NR & 1) col[NR - 1] = gc[NR - 1];
        }

        // ================= S4/S5: L L^T = gr_lhs (in place in ST[1:,1:]) ; Y = L^-1 gr_rhs  hho.hpp:63,92
        PA_TICK(2);
        if (PA_PRIO_OUT >= 0 && PA_PRIO_MID >= 0) __builtin_amdgcn_s_setprio(PA_PRIO_MID >= 0 ? PA_PRIO_MID : 0);
        PA_MARK("S4");
        double *LG = S + C::oLG;                 // without the pre-pass: stiff[1:,1:], symmetric: row-major == column-major
        double lreg[C::NLR];                     // DPPFWD: [packed L | 1 / diagonal], element e in lane e mod 16 of a row, register e / 16
        int bad = bad_pre;
        // blocked form where its extra registers are free (measured: -3 % at k = 1; +4 % at k = 2, where it
        // pushes the kernel into spills)
        if (!C::USE_PRE && !(a.ablate & 8u)) bad = NR <= 6 ? lds_cholesky_blocked<NR, LD, G>(LG, l) : lds_cholesky<NR, LD, G>(LG, l);
        if (C::USE_PRE) {
            if (!(a.ablate & 16u)) {
                if (C::DPPFWD) {
#pragma unroll
                    for (int t = 0; t < C::NLR; ++t) lreg[t] = S[C::oREC + (lane & 15) + 16 * t];      // (reads past NL + NR stay inside the record)
                    dpp_forward<NR, C::NLR>(lreg, col);
                } else if (PA_FWD_CAP > 0) lds_forward_rd_batched<NR, LD, PA_FWD_CAP>(LG, S + C::oRCP, col);
                else lds_forward_rd<NR, LD>(LG, S + C::oRCP, col);
                if (!C::LAPG) {
                // cell column c >= 1: gr_rhs[:, c] = stiff[1:, c] - F_c and L^-1 stiff[1:, c] = L^T e_(c-1): add row c-1 of L
                // (the image is zero above the diagonal; the other columns add its zero row NR)
                const double *lr = LG + ((l >= 1 && l < CBS) ? l - 1 : NR) * LD;
#pragma unroll
                for (int i = 0; i + 1 < NR; i += 2) {
                    const double2 v = lds_pair(lr + i);
                    col[i] += v.x;
                    col[i + 1] += v.y;
                }
                if (NR & 1) col[NR - 1] += lr[NR - 1];
                }
            }
        } else if (!(a.ablate & 16u)) lds_forward<NR, LD>(LG, col);
        PA_TICK(11);
        // The trace columns are formed only now (not next to the gr_rhs columns in S3b): they stay
        // out of the register budget of the factorization.  They read the face tables of region Q,
        // which Z overwrites: Y goes to LDS after the barrier below.
        // trace columns / (|F|/2):  tr[fk] = sum_q w_q t_q^k phi_c(x_fq)   hho.hpp:209-216 / 133-140.
        // Point q of face f IS the reference's q-th face point (S1 mirrors t for faces whose lower-id
        // endpoint comes second), so its face-basis value is t_q^k for either orientation.
        // lc only, T_F = [trace_F | 0]: the cell part of U is formed by UNITS (face f, cell column cc), one lane each --
        // 4 CBS units of FBS entries (trace, L^^-1, scale) instead of every column lane carrying all NF rows of its
        // column through the same steps with CBS of G lanes useful; the face columns of U are -sqrt(|F|/2h) L^^T E_F,
        // constants times the face's scale.  Nothing of U is held in registers across the stages.
        double uval[UR][FBS];
        double uph[UR][NFQ], usu[UR];
        uint32_t upk[UR];
#pragma unroll
        for (int r = 0; r < UR; ++r) { upk[r] = unit_pk[r]; asm volatile("" : "+v"(upk[r])); }      // (unpacked per pass, not hoisted)
        if (UNIT_U && !(a.ablate & 32u)) {
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                const double *ph = S + C::oPHF + (upk[r] & 0xfffu);
#pragma unroll
                for (int q = 0; q < NFQ; ++q) uph[r][q] = PA_LDS_NOPAIR ? lds_single(ph + q * RBS) : ph[q * RBS];
                usu[r] = S[C::oSU + (upk[r] >> 24)];
            }
            // (every read of the units first: one LDS round trip)
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                double x[FBS];
#pragma unroll
                for (int k = 0; k < FBS; ++k) x[k] = 0.0;
#pragma unroll
                for (int q = 0; q < NFQ; ++q)
#pragma unroll
                    for (int k = 0; k < FBS; ++k) x[k] += ft.cw[q][k] * uph[r][q];
                face_forward<FBS>(ft, x);
#pragma unroll
                for (int k = 0; k < FBS; ++k) uval[r][k] = usu[r] * x[k];
            }
        }
        PA_TICK(12);
        double ucol[C::HAS_STAB ? NF : 1];
#pragma unroll
        for (int r = 0; r < (C::HAS_STAB ? NF : 1); ++r) asm volatile("" : "=v"(ucol[r]));
        if (C::HAS_STAB && !UNIT_U && !(a.ablate & 32u)) {
            constexpr int TCOLS = C::GENERAL_FANCY ? RBS : CBS;
            const int m = l < TCOLS ? l : 0;
#pragma unroll
            for (int r = 0; r < NF; ++r) ucol[r] = 0.0;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int q = 0; q < NFQ; ++q) {
                    const double ph = S[C::oPHF + (f * NFQ + q) * RBS + m];
#pragma unroll
                    for (int k = 0; k < FBS; ++k) ucol[f * FBS + k] += ft.cw[q][k] * ph;
                }
            if (C::GENERAL_FANCY && l < RBS) {
#pragma unroll
                for (int r = 0; r < NF; ++r) S[C::oFT + r + m * NF] = ucol[r];
            }
        }
        wave_sync();
        if (l < MS) {
#pragma unroll
            for (int k = 0; k < NR; ++k) S[C::oZ + k + c * ZS] = col[k];
            if (NRP != NR) S[C::oZ + NR + c * ZS] = 0.0;
        }
        double ycol[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) ycol[k] = col[k];
        if (C::GENERAL_FANCY || a.oper != nullptr) {
            if (C::DPPFWD) dpp_backward<NR, C::NLR>(lreg, col);
            else if (C::USE_PRE) lds_backward_rd<NR, LD>(LG, S + C::oRCP, col);
            else lds_backward<NR, LD>(LG, col);            // col = oper[:, c]
            if (a.oper != nullptr && valid && l < MS) {
                double *dst = a.oper + rel(NR * MS) + (size_t)c * NR;
#pragma unroll
                for (int k = 0; k < NR; ++k) dst[k] = col[k];
            }
        }

        PA_TICK(3);
        // ================= S6: column c of U ==========================================
        if (PA_PRIO_OUT >= 0 && PA_PRIO_S6 >= 0) __builtin_amdgcn_s_setprio(PA_PRIO_S6 >= 0 ? PA_PRIO_S6 : 0);
        PA_MARK("S6");
        if (UNIT_U && !(a.ablate & 32u)) {
#pragma unroll
            for (int r = 0; r < UR; ++r) {
                double *zu = S + ((upk[r] >> 12) & 0xfffu);
#pragma unroll
                for (int k = 0; k < FBS; ++k) zu[k] = uval[r][k];
            }
            if (l >= CBS && l < MS) {
                // the NF rows of a face column: zeros (16-byte stores), then the block of its own face on top of them
                // (LDS writes of a lane land in order)
                double *zc = S + C::oZ + NRP + c * ZS;
                static_assert(NF % 2 == 0 && NRP % 2 == 0 && ZS % 2 == 0, "16-byte aligned U columns");
#pragma unroll
                for (int r = 0; r < NF; r += 2) *reinterpret_cast<double2 *>(zc + r) = double2{0.0, 0.0};
                const double msu = -S[C::oSU + fc];
#pragma unroll
                for (int k = 0; k < FBS; ++k) zc[(UPERM_ON ? ((fc + 4 - C::UF1) & 3) : fc) * FBS + k] = msu * ufc[k];
            }
        }
        if (C::HAS_STAB && !UNIT_U && !(a.ablate & 32u)) {
            if (C::GENERAL_FANCY) {
                // proj1[:, c] = e_c - M1^-1 (M2 R[:, c])   hho.hpp:184-190
                double pr[CBS];
#pragma unroll
                for (int i = 0; i < CBS; ++i) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < NR; ++k)
                        s += (C::USE_PRE ? S[C::oMRl + (1 + k) * CBS + i] : S[C::oMA + i + (1 + k) * LD]) * col[k];
                    pr[i] = s;
                }
                int badm;
                if (C::USE_PRE) {          // chol(M1) comes with the record
                    badm = (int)S[C::oSU + 17];
                    lds_forward_rd<CBS, C::LDM>(S + C::oMCl, S + C::oMCRl, pr);
                    lds_backward_rd<CBS, C::LDM>(S + C::oMCl, S + C::oMCRl, pr);
                } else {
                    wave_sync();
                    badm = lds_cholesky<CBS, LD, G>(S + C::oMA, l);
                    lds_forward<CBS, LD>(S + C::oMA, pr);
                    lds_backward<CBS, LD>(S + C::oMA, pr);
                }
                if (badm && !bad) bad = 100 + badm;
#pragma unroll
                for (int i = 0; i < CBS; ++i) pr[i] = (i == c ? 1.0 : 0.0) - pr[i];
                // T_F[:, c] / (|F|/2) = MR1 R[:, c] + MR2 proj1[:, c]   (hho.hpp:222-230; piKF.solve is linear)
#pragma unroll
                for (int r = 0; r < NF; ++r) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < NR; ++k) s += S[C::oFT + r + (1 + k) * NF] * col[k];
#pragma unroll
                    for (int k = 0; k < CBS; ++k) s += S[C::oFT + r + k * NF] * pr[k];
                    ucol[r] = s;
                }
            } else if (l >= CBS) {
#pragma unroll
                for (int r = 0; r < NF; ++r) ucol[r] = 0.0;     // T_F = [trace_F | 0]
            }
            // U_F = sqrt(|F|/2h) ( L^^-1 T_F/(|F|/2) - L^^T E_F )
            const double4 su4 = *reinterpret_cast<const double4 *>(S + C::oSU);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const double su = f == 0 ? su4.x : f == 1 ? su4.y : f == 2 ? su4.z : su4.w;
                double xf[FBS];
#pragma unroll
                for (int k = 0; k < FBS; ++k) xf[k] = ucol[f * FBS + k];
                face_forward<FBS>(ft, xf);
#pragma unroll
                for (int k = 0; k < FBS; ++k) {
                    const double e = (l >= CBS && fc == f) ? ufc[k] : 0.0;      // (L^^T E_F)[k][c]
                    ucol[f * FBS + k] = su * (xf[k] - e);
                }
            }
            if (l < MS) {
#pragma unroll
                for (int r = 0; r < NF; ++r) S[C::oZ + NRP + r + c * ZS] = ucol[r];
            }
        }
        wave_sync();
        PA_TICK(4);
        // region P is free (L, reciprocals, scalars all consumed): next cell's record -- in the condensed mode only after
        // S9, whose image lies over P as well
        // corner tile on the vector pipe, short form: its U term needs the face scale, which the next record is about to replace
        double cornerU = 0.0;
        int cic = 0, cjc = 0;
        if (C::CORNER_VALU && !SPLIT && !PA_LC_VALU) {
            uint32_t cpk = corner_pk;
            asm volatile("" : "+v"(cpk));
            cic = (int)(cpk & 0xffu); cjc = (int)((cpk >> 8) & 0xffu);       // cic <= cjc
            if (CORNER_SHORT && l < C::NCORNER * (C::NCORNER + 1) / 2) {
                const double su = S[C::oSU + (cpk >> 16)];
                cornerU = cornerT * su * su;
            }
        }
        double t11s[T11_SHORT ? C::CPW : 1];      // squared scale of the lane's tile-(1,1) column face, for each cell of the wavefront
        if (T11_SHORT) {
#pragma unroll
            for (int gi = 0; gi < C::CPW; ++gi) {
                const double su = smem[gi * C::LDS_PER_CELL + C::oSU + t11f];
                t11s[gi] = su * su;
            }
        }
        if (C::USE_PRE && (!COND || C::COND_OWN_P) && (!SELF || ib + 1 < nbc)) rec_deposit();
        // condensed mode: the cell's right-hand side (lanes < CBS) and, for the recovery, its face unknowns (lanes < NF),
        // one value per lane, in flight during the product
        double fT_l = 0.0, uF_l = 0.0;
        if (COND) {
            if (a.rhs != nullptr && l < CBS) fT_l = a.rhs[rel(CBS) + l];
            if (a.uF != nullptr && l < NF) uF_l = a.uF[rel(NF) + l];
        }

        // ================= S7/S8 (lc only): lc = Z^T Z on the matrix pipe ============
        // v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4) B(4x16); lane l supplies A[l&15][l>>4] and
        // B[l>>4][l&15] and receives D[(l>>4) + 4r][l&15], r = 0..3.  For a tile (I, J) of Z^T Z both
        // operands are elements of Z: A[i][k] = Z[k][16I+i], B[k][j] = Z[k][16J+j] -- one LDS read per
        // lane feeds 16 FMAs (the vector form reads one LDS double per FMA).  All 64 lanes work on one
        // cell at a time; the CPW cells of the wavefront are processed in turn.
        PA_TICK(5);
        if (PA_PRIO_OUT >= 0) __builtin_amdgcn_s_setprio(PA_PRIO_OUT >= 0 ? PA_PRIO_OUT : 0);
        if (!SPLIT && !PA_LC_VALU) {
            PA_MARK("S7m");
            constexpr int NTL = (MS + 15) / 16;              // column tiles
            constexpr int KS = (C::ZR + 3) / 4;              // k-steps
            constexpr int NPAIRS = NTL * (NTL + 1) / 2;
            typedef double v4d __attribute__((ext_vector_type(4)));
            const int kk = lane >> 4, jj = lane & 15;
            constexpr bool DIRECT = C::DIRECT_STORE && !COND;      // condensed mode: always through the LDS image
            constexpr bool EARLY_OUT = PA_EARLY_OUT && !DIRECT && !COND && C::CPW > 1;
            constexpr int OS = COND ? C::LDI : MS;                 // stride of the image
            // Through the image with an even stride: the columns of Z are fed to the tiles in the order pi(i) = 4 (i & 3) + (i >> 2),
            // both operands alike, so that the lane's four accumulators D[kk + 4 r][jj] are lc[16 I + 4 kk + r][16 J + pi(jj)], r = 0..3:
            // FOUR CONSECUTIVE rows of one column of the image -- two 16-byte LDS writes instead of four 8-byte ones, and the 16 lanes
            // of a write cycle (kk fixed, pi(jj) = all 16 columns) fall on 16 different bank quads where the 8-byte column writes at
            // stride 22 met two by two.  The mirror copy of an off-diagonal tile stays 8-byte writes.
            constexpr bool ACC_PERM = PA_ACC_PERM && !DIRECT && !COND && OS % 2 == 0;      // (condensed k = 2: 8 spilled registers with it, and no gain)
            const int pj = ACC_PERM ? 4 * (jj & 3) + (jj >> 2) : jj;
            static_assert(!T11_SHORT || ACC_PERM == ACC_PERM_, "the per-lane constants of tile (1,1) assume this tile's lane -> (row, column) map");
            const bool want_image = COND || a.lc != nullptr;
            // no barrier is needed between the cells: the wavefront reads a cell's Z and then overwrites
            // it with the same cell's output image in program order
            wave_sync();      // Z complete (all columns written)
            // corner tile on the vector pipe: every group for its own cell, before the image may overwrite Z
            double corner = 0.0;
            if (C::CORNER_VALU && l < C::NCORNER * (C::NCORNER + 1) / 2 && !(a.ablate & 64u)) {
                const double *zi = S + C::oZ + (16 + cic) * ZS, *zj = S + C::oZ + (16 + cjc) * ZS;
                constexpr int NROW = CORNER_SHORT ? NRP : C::ZR;             // (row NR of Z is a zero pad)
                typedef double v2d_ __attribute__((ext_vector_type(2)));
                v2d_ pz[NROW / 2], qz[NROW / 2];                             // every read first: one LDS round trip
#pragma unroll
                for (int k = 0; k + 1 < NROW; k += 2) {
                    pz[k / 2] = *reinterpret_cast<const v2d_ *>(zi + k);
                    qz[k / 2] = *reinterpret_cast<const v2d_ *>(zj + k);
                }
                double s0 = cornerU, s1 = 0.0;
#pragma unroll
                for (int k = 0; k + 1 < NROW; k += 2) {
                    s0 = __builtin_fma(pz[k / 2].x, qz[k / 2].x, s0);
                    s1 = __builtin_fma(pz[k / 2].y, qz[k / 2].y, s1);
                }
                if (NROW & 1) s0 = __builtin_fma(zi[NROW - 1], zj[NROW - 1], s0);
                corner = s0 + s1;
            }
#pragma unroll
            for (int gi = 0; gi < C::CPW; ++gi) {
                v4d acc[NPAIRS];
                const double *Zg = smem + gi * C::LDS_PER_CELL + C::oZ;
                double *Og = smem + gi * C::LDS_PER_CELL + C::oOUT;
                // (the accumulators start from the zero operand of each tile's first instruction; on the profiling path that
                // skips the product they are left undefined -- zeroing them ahead of the branch cost 8 moves per cell)
                if (a.ablate & 64u) {
#pragma unroll
                    for (int t = 0; t < NPAIRS; ++t) asm volatile("" : "=v"(acc[t]));
                } else {
#pragma unroll
                    for (int t = 0; t < NPAIRS; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int k = 4 * ks + kk;
                        double z[NTL];
#pragma unroll
                        for (int t = 0; t < NTL; ++t) {
                            if (UPERM_ON && t == 1 && ks >= C::KS1) { z[t] = 0.0; continue; }
                            const int col = 16 * t + pj;
                            // (decided at compile time wherever the whole tile row / k-step is inside Z)
                            const bool rok = 4 * ks + 3 < C::ZR || k < C::ZR, cok = 16 * t + 15 < MS || col < MS;
                            // (a volatile read stays ONE ds_read_b64 -- 64 banks, two groups of 32 lanes, 2 cycles --; the compiler otherwise pairs the
                            // reads of two k-steps into ds_read2_b64: 32 banks, groups of 16 lanes, 8 cycles, and columns j, j + 8 on one bank at this stride)
                            const double v = PA_LDS_NOPAIR ? lds_single(Zg + ((rok && cok) ? k + col * ZS : 0)) : Zg[(rok && cok) ? k + col * ZS : 0];
                            // A lane beyond the last COLUMN of Z feeds only the rows / columns >= MS of the tiles (D[i][j] takes row i of
                            // A and column j of B), which are never stored: whatever it read (element 0 of Z) may stay.  A lane beyond the
                            // last ROW of Z would add to entries that are.
                            z[t] = (PA_ZCOL_NOSEL ? rok : (rok && cok)) ? v : 0.0;
                        }
                        // tile (1,1), short form: rows of Y only (the last of their k-steps may also hold rows of U: masked)
                        constexpr int KSY = cdiv(NRP, 4);
                        double z1y = z[NTL - 1];
                        if (T11_SHORT && ks == KSY - 1 && NRP % 4 != 0) z1y = k < NRP ? z1y : 0.0;
                        int t = 0;
#pragma unroll
                        for (int I = 0; I < NTL; ++I)
#pragma unroll
                            for (int J = I; J < NTL; ++J, ++t) {
                                if (T11_SHORT && I == 1 && J == 1) {
                                    if (ks < KSY) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(z1y, z1y, acc[t], 0, 0, 0);
                                } else if (!(C::CORNER_VALU && I == 1 && J == 1) && !(UPERM_ON && J == 1 && ks >= C::KS1))
                                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[I], z[J], acc[t], 0, 0, 0);
                            }
                    }
                    if (T11_SHORT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[NPAIRS - 1][r] = __builtin_fma(t11c[r], t11s[gi], acc[NPAIRS - 1][r]);
                    }
                }
                PA_TICK(6 + 2 * (gi & 1));
                PA_MARK("S8m");
                if (DIRECT) {
                    // ---- S8 (lc only): straight from the accumulators to HBM, no LDS image.  lc is symmetric:
                    // the lane's D[row][col] is written at (col, row), where the 16 lanes of a group
                    // (jj = 0..15) cover 16 consecutive rows of one column = one 128-byte run; the second copy
                    // of an off-diagonal tile is 32-byte runs (kk = 0..3).  All 64 lanes work on cell gi.
                    if (a.lc != nullptr && !(a.ablate & 128u)) {
                        const bool valid_g = base + gi < a.n;
                        double *o = a.lc + (valid_g ? base + gi : 0) * (size_t)(MS * MS);
                        int t = 0;
#pragma unroll
                        for (int I = 0; I < NTL; ++I)
#pragma unroll
                            for (int J = I; J < NTL; ++J, ++t) {
                                const int colj = 16 * J + jj;
                                if (C::CORNER_VALU && I == 1 && J == 1) continue;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * I + kk + 4 * r;
                                    if (16 * I + 4 * r < MS && valid_g && (16 * J + 15 < MS || colj < MS) &&
                                        (16 * I + 4 * r + 3 < MS || row < MS)) {
                                        const double v = acc[t][r];
                                        o[colj + row * MS] = v;                 // entry (colj, row)
                                        if (I != J) o[row + colj * MS] = v;     // entry (row, colj)
                                    }
                                }
                            }
                    }
                } else {
                if (want_image && !(a.ablate & 128u)) {
                    int t = 0;
#pragma unroll
                    for (int I = 0; I < NTL; ++I)
#pragma unroll
                        for (int J = I; J < NTL; ++J, ++t) {
                            const int colj = 16 * J + pj;
                            if ((16 * J + 15 < MS || colj < MS) && !(C::CORNER_VALU && I == 1 && J == 1)) {
                                if (ACC_PERM) {
                                    const int row0 = 16 * I + 4 * kk;
                                    typedef double v2d_ __attribute__((ext_vector_type(2)));
                                    if (16 * I + 15 < MS || row0 + 3 < MS) {
                                        *reinterpret_cast<v2d_ *>(Og + row0 + colj * OS) = v2d_{acc[t][0], acc[t][1]};
                                        *reinterpret_cast<v2d_ *>(Og + row0 + 2 + colj * OS) = v2d_{acc[t][2], acc[t][3]};
                                    } else {
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                            if (row0 + r < MS) Og[row0 + r + colj * OS] = acc[t][r];
                                    }
                                    if (I != J) {
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                            if (16 * I + 15 < MS || row0 + r < MS) Og[colj + (row0 + r) * OS] = acc[t][r];
                                    }
                                } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int row = 16 * I + kk + 4 * r;
                                    if (16 * I + 4 * r < MS && (16 * I + 4 * r + 3 < MS || row < MS)) {
                                        const double v = acc[t][r];
                                        Og[row + colj * OS] = v;
                                        if (I != J) Og[colj + row * OS] = v;
                                    }
                                }
                                }
                            }
                        }
                }
                // lc only: the image of cell gi goes out at once, all 64 lanes on it (1 KB per store instruction), while the
                // matrix instructions of the next cell run -- not both cells in one burst of stores at the end of the pass
                if (EARLY_OUT && a.lc != nullptr && !(a.ablate & 128u)) {
                    if (C::CORNER_VALU && g == gi && l < C::NCORNER * (C::NCORNER + 1) / 2) {
                        Og[(16 + cic) + (16 + cjc) * OS] = corner;
                        Og[(16 + cjc) + (16 + cic) * OS] = corner;
                    }
                    wave_sync();
                    if (base + gi < a.n) {
                        double *o = a.lc + ((a.ablate & 1024u) ? lblock * C::CPW + gi : base + gi) * (size_t)(MS * MS);
                        constexpr int NPAIR = MS * MS / 2, NIT = cdiv(NPAIR, 64);
                        typedef double v2d_ __attribute__((ext_vector_type(2)));
                        v2d_ img[NIT];
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const int e = it * 64 + lane;
                            img[it] = *reinterpret_cast<const v2d_ *>(Og + 2 * (((it + 1) * 64 <= NPAIR || e < NPAIR) ? e : 0));
                        }
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const int e = it * 64 + lane;
                            if ((it + 1) * 64 <= NPAIR || e < NPAIR) *reinterpret_cast<v2d_ *>(o + 2 * e) = img[it];
                        }
                        if (((MS * MS) & 1) && lane == 0) o[MS * MS - 1] = Og[MS * MS - 1];
                    }
                }
                }
                PA_TICK(7 + 2 * (gi & 1));
            }
            if (C::CORNER_VALU && want_image && !EARLY_OUT && !(a.ablate & 128u) && l < C::NCORNER * (C::NCORNER + 1) / 2) {
                if (DIRECT) {
                    if (valid) {
                        double *o = a.lc + rel(MS * MS);
                        o[(16 + cic) + (16 + cjc) * MS] = corner;
                        o[(16 + cjc) + (16 + cic) * MS] = corner;
                    }
                } else {
                    S[C::oOUT + (16 + cic) + (16 + cjc) * OS] = corner;
                    S[C::oOUT + (16 + cjc) + (16 + cic) * OS] = corner;
                }
            }
            if (COND) {
                // ================= S9 (condensed mode): eliminate the cell unknowns in the LDS image ==================
                // Row i of the symmetric (MS + 1) x (MS + 1) matrix  M = [A_TT A_TF f_T; A_FT A_FF 0; f_T^T 0 0]  belongs to
                // lane i.  The first CBS steps of its Cholesky factorization, row by row (left-looking, Eigen's LLT order on
                // A_TT), leave  L_TT = chol(A_TT)  in the cell rows,  W^T = A_FT L_TT^-T  in the face rows and
                // w0^T = f_T^T L_TT^-T  in the last one -- the forward substitutions ride on the factorization's chain --
                // and the Schur complement of the pivots is what is asked for:
                //   S = A_FF - W^T W,   g = -W^T w0,   uT = L_TT^-T (w0 - W uF).
                PA_MARK("S9");
                constexpr int LDI = C::LDI, NPV = CBS;
                double *A = S + C::oOUT;
                if (l < CBS) A[MS * LDI + l] = fT_l;                            // row MS: f_T
                if (a.uF != nullptr && l < NF) A[MS * LDI + CBS + l] = uF_l;       // (behind it: the face unknowns)
                wave_sync();
                const int i = l <= MS ? l : 0;                                  // (lanes beyond the last row mirror row 0)
                double row[NPV];
#pragma unroll
                for (int k = 0; k + 1 < NPV; k += 2) {
                    const double2 v = lds_pair(A + i * LDI + k);
                    row[k] = v.x; row[k + 1] = v.y;
                }
                if (NPV & 1) row[NPV - 1] = A[i * LDI + NPV - 1];
                int badc = 0;
                constexpr bool COND_DPP = PA_COND_DPP && NPV <= 16 && MS + 1 <= 32 && G <= 32;
                if (COND_DPP) {
                    // The chain in registers: lane q of each row of 16 lanes holds image rows q and q + 16 (both rows of 16 lanes
                    // of a 32-lane group hold the same pairs), so every pivot row -- there are at most 16 -- sits in lane j of the
                    // row of 16 and its prefix reaches the other lanes through the DPP operand of the FMAs (row_newbcast:j):
                    // no LDS read of the pivot row, no broadcast of the pivot by v_readlane, no LDS write and wait per step.
                    if (!(a.ablate & 256u)) {
                    constexpr bool TWO = MS + 1 > 16;
                    const int q = lane & 15;
                    const int iA = q <= MS ? q : 0, iB = (TWO && q + 16 <= MS) ? q + 16 : 0;
                    double rA[NPV], rB[TWO ? NPV : 1], rdiag = 0.0;
#pragma unroll
                    for (int k = 0; k + 1 < NPV; k += 2) {
                        const double2 va = lds_pair(A + iA * LDI + k);
                        rA[k] = va.x; rA[k + 1] = va.y;
                        if (TWO) { const double2 vb = lds_pair(A + iB * LDI + k); rB[k] = vb.x; rB[k + 1] = vb.y; }
                    }
                    if (NPV & 1) { rA[NPV - 1] = A[iA * LDI + NPV - 1]; if (TWO) rB[NPV - 1] = A[iB * LDI + NPV - 1]; }
                    if (NPV >= PA_COND_RL_MIN) cond_dpp_chain_rl<NPV, TWO>(rA, rB, rdiag, badc, q);
                    else cond_dpp_chain<NPV, TWO>(rA, rB, rdiag, badc, q);
                    // back to the image (the diagonal holds 1 / L[j][j]); one row of 16 lanes writes
                    if ((G == 16 || l < 16)) {
#pragma unroll
                        for (int k = 0; k < NPV; ++k) {
                            if (q <= MS) A[iA * LDI + k] = (k == q) ? rdiag : rA[k];
                            if (TWO && q + 16 <= MS) A[iB * LDI + k] = rB[k];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NPV; ++k) row[k] = (TWO && l >= 16) ? rB[TWO ? k : 0] : rA[k];
                    wave_sync();
                    }
                } else if (PA_COND_NB > 1) {
                    if (!(a.ablate & 256u)) badc = lds_partial_cholesky<NPV, MS + 1, PA_COND_NB, LDI>(A, l, row);
                } else if (!(a.ablate & 256u)) {
#pragma unroll
                for (int j = 0; j < NPV; ++j) {
                    // M[i][j] - sum_{k<j} L[i][k] L[j][k]; row j's prefix is in LDS already
                    const double s = j == 0 ? row[0] : lds_dotsub_n(row[j], A + j * LDI, row, j);
                    const double d = group_broadcast<G>(s, j);
                    if (!(d > 0.0) && !badc) badc = j + 1;
                    const double r = fast_rsqrt<1>(d);
                    row[j] = s * r;
                    if (l <= MS && l >= j) A[i * LDI + j] = (l == j) ? r : row[j];      // the diagonal holds 1 / L[j][j]
                    wave_sync();
                }
                }
                PA_TICK(13);
                if (badc && !bad) bad = 200 + badc;
                if (a.uF == nullptr) {
                    // Schur complement, row i' = l - CBS of it by lane l: entries (m', i'), m' <= i', are the run
                    // i'(i'+1)/2 .. of the column-packed upper triangle
                    double *stg = S + C::oOUT + C::oSTG;
                    const int ip = l - CBS;
                    const bool frow = l >= CBS && l < MS;
                    constexpr bool SCHUR_MFMA = NF >= PA_COND_SCHUR_MFMA_MIN && NF <= 16;
                    if (a.ablate & 512u) {
                    } else if (SCHUR_MFMA) {
                        // W^T W on the matrix pipe, all 64 lanes on one cell at a time: W[k][j] = L[CBS + j][k] is both
                        // operands of the one 16 x 16 tile (one LDS read per lane and k-step); the lane's P[kk + 4r][jj]
                        // meets A_FF(kk + 4r, jj) of the image's lower triangle and goes to the packed upper one
                        const double gi = lds_dotsub<NPV>(0.0, A + MS * LDI, row);
#pragma unroll
                        for (int gi_ = 0; gi_ < C::CPW; ++gi_) {
                            const double *Ag = smem + gi_ * C::LDS_PER_CELL + C::oOUT;
                            double *sg = smem + gi_ * C::LDS_PER_CELL + C::oOUT + C::oSTG;
                            v4d p = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int ks = 0; ks < (NPV + 3) / 4; ++ks) {
                                const int k = 4 * ks + kk;
                                const bool ok = (4 * ks + 3 < NPV || k < NPV) && (NF == 16 || jj < NF);
                                const double w = Ag[ok ? (CBS + jj) * LDI + k : 0];
                                const double wz = ok ? w : 0.0;
                                p = __builtin_amdgcn_mfma_f64_16x16x4f64(wz, wz, p, 0, 0, 0);
                            }
                            double aff[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int ir = kk + 4 * r;
                                const bool up = ir <= jj && (NF == 16 || jj < NF);
                                aff[r] = Ag[up ? (CBS + jj) * LDI + CBS + ir : 0];
                            }
                            // (the staging area lies on the cell rows: every read of W above has been issued, and LDS
                            // operations of a wavefront execute in order)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int ir = kk + 4 * r;
                                if (4 * r < NF && ir <= jj && (NF == 16 || jj < NF)) sg[jj * (jj + 1) / 2 + ir] = aff[r] - p[r];
                            }
                        }
                        if (frow) stg[C::NSP + ip] = gi;
                    } else {
#pragma unroll
                    for (int mp = 0; mp < NF; ++mp) {
                        const double s = lds_dotsub<NPV>(A[i * LDI + CBS + mp], A + (CBS + mp) * LDI, row);
                        if (frow && mp <= ip) stg[ip * (ip + 1) / 2 + mp] = s;
                    }
                    const double gi = lds_dotsub<NPV>(0.0, A + MS * LDI, row);
                    if (frow) stg[C::NSP + ip] = gi;
                    }
                    wave_sync();
                    PA_TICK(14);
                    if (valid && a.cond != nullptr) {
                        double *o = a.cond + rel(C::NCOND);
                        static_assert(C::NCOND % 2 == 0, "16-byte stores of the packed record");
#pragma unroll
                        for (int e0 = 0; e0 < C::NCOND / 2; e0 += G) {
                            const int e = e0 + l;
                            if (e < C::NCOND / 2)
                                *reinterpret_cast<double2 *>(o + 2 * e) = *reinterpret_cast<const double2 *>(stg + 2 * e);
                        }
                    }
                } else {
                    // recovery: t = w0 - W uF (lane k < CBS: column k of the face rows), then L_TT^T uT = t column by column
                    double t = 0.0;
                    const int k = l < CBS ? l : 0;
                    t = A[MS * LDI + k];
#pragma unroll
                    for (int ip = 0; ip < NF; ++ip) t = __builtin_fma(-A[(CBS + ip) * LDI + k], A[MS * LDI + CBS + ip], t);
                    const double rdk = A[k * LDI + k];
#pragma unroll
                    for (int m = NPV - 1; m >= 0; --m) {
                        const double um = group_broadcast<G>(t * rdk, m);
                        if (l == m) t = um;
                        else if (l < m) t = __builtin_fma(-A[m * LDI + k], um, t);
                    }
                    if (valid && a.uT != nullptr && l < CBS) a.uT[rel(CBS) + l] = t;
                }
                wave_sync();      // every read of the image is done
                if (C::USE_PRE && !C::COND_OWN_P && (!SELF || ib + 1 < nbc)) { rec_deposit(); wave_sync(); }      // the next cell's record, into its place in region P
            } else if (DIRECT) {
                wave_sync();      // the next cell's tables overwrite Z
            } else if (EARLY_OUT) {
                wave_sync();      // the next cell's tables overwrite the images
            } else {
                if (a.lc != nullptr && !(a.ablate & 128u)) {
                    wave_sync();
                    if (valid) {
                        // (profiling, bit 1024: every pass of a block writes the same two matrices -- the stores without their HBM traffic)
                        // (the block's part of the address is wave-uniform: scalar arithmetic; the lane adds its group's matrix)
                        const size_t cb = (a.ablate & 1024u) ? lblock * C::CPW : base;
                        double *o = a.lc + cb * (size_t)(MS * MS) + (uint32_t)g * (uint32_t)(MS * MS);
                        constexpr int NPAIR = MS * MS / 2, NIT = cdiv(NPAIR, G);
                        // every read of the image first, then the stores: with a read and its store per step the compiler
                        // reuses one register quad and the steps become a chain of LDS round trips
                        typedef double v2d_ __attribute__((ext_vector_type(2)));
                        v2d_ img[NIT];
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const int e = it * G + l;
                            img[it] = *reinterpret_cast<const v2d_ *>(S + C::oOUT + 2 * (((it + 1) * G <= NPAIR || e < NPAIR) ? e : 0));
                        }
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const int e = it * G + l;
                            if ((it + 1) * G <= NPAIR || e < NPAIR) *reinterpret_cast<v2d_ *>(o + 2 * e) = img[it];
                        }
                        if ((MS * MS) & 1) {
                            if (l == 0) o[MS * MS - 1] = S[C::oOUT + MS * MS - 1];
                        }
                    }
                }
                wave_sync();
            }
        } else {
        // ================= S7: lc = Z^T Z, entries (c, c + d mod MS) ==================
        PA_MARK("S7");
        double acc_d[ND], acc_s[SPLIT ? ND : 1];
#pragma unroll
        for (int d = 0; d < ND; ++d) { acc_d[d] = 1.0; if (SPLIT) acc_s[d] = 1.0; }
        if (!(a.ablate & 64u)) {
            int cp = c;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double *zc = S + C::oZ + cp * ZS;
                double s = lds_dotadd<NR>(0.0, zc, ycol);
                if (SPLIT) {
                    double u = 0.0;
                    if (C::HAS_STAB) u = lds_dotadd<NF>(0.0, zc + NRP, ucol);
                    asm volatile("" : "+v"(s), "+v"(u));     // pin: keep the FMAs next to their LDS reads
                    acc_d[d] = s; acc_s[d] = u;
                } else {
                    if (C::HAS_STAB) s = lds_dotadd<NF>(s, zc + NRP, ucol);
                    asm volatile("" : "+v"(s));
                    acc_d[d] = s;
                }
                cp = (cp + 1 == MS) ? 0 : cp + 1;
            }
        }
        wave_sync();      // every read of L and Z is done: the output image may overwrite Z

        // ================= S8: mirror through LDS, stream to HBM ======================
        PA_MARK("S8");
#pragma unroll
        for (int which = 0; which < (SPLIT ? 3 : 1); ++which) {
            double *dst = which == 0 ? a.lc : which == 1 ? a.data : a.stab;
            if (dst == nullptr || (a.ablate & 128u)) continue;
            if (l < MS) {
                int cp = c;
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    double v = acc_d[d];
                    if (SPLIT) v = which == 0 ? acc_d[d] + acc_s[d] : which == 1 ? acc_d[d] : acc_s[d];
                    S[C::oOUT + c + cp * MS] = v;
                    S[C::oOUT + cp + c * MS] = v;
                    cp = (cp + 1 == MS) ? 0 : cp + 1;
                }
            }
            wave_sync();
            if (valid) {
                double *o = dst + rel(MS * MS);
                constexpr int NPAIR = MS * MS / 2, NIT = cdiv(NPAIR, G);
                typedef double v2d_ __attribute__((ext_vector_type(2)));
                v2d_ img[NIT];                             // reads first, then the stores (see the lc-only path)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int e = it * G + l;
                    img[it] = *reinterpret_cast<const v2d_ *>(S + C::oOUT + 2 * (((it + 1) * G <= NPAIR || e < NPAIR) ? e : 0));
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int e = it * G + l;
                    if ((it + 1) * G <= NPAIR || e < NPAIR) *reinterpret_cast<v2d_ *>(o + 2 * e) = img[it];
                }
                if ((MS * MS) & 1) {
                    if (l == 0) o[MS * MS - 1] = S[C::oOUT + MS * MS - 1];
                }
            }
            wave_sync();
        }
        }
        if (valid && a.info != nullptr && l == 0) a.info[rel(1)] = bad;
        PA_TICK(10);
    }
#if PA_SELF_PRE
    }
#endif
#ifdef PA_STAGE_CLOCK
    if (a.dbg != nullptr && lane == 0)
        for (int i = 0; i < PA_NSTAGE; ++i) a.dbg[(size_t)blockIdx.x * PA_NSTAGE + i] = tk_sum[i];
#endif
}

}  // namespace pa
