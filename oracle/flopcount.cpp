// flopcount.cpp -- the oracle's restatement compiled with every double-precision operation counted (test
// infrastructure).  `double` becomes a counting wrapper over the unchanged C source hho_oracle.c: the result is the EXACT
// number of FP64 adds / multiplies / divisions / square roots the reference's algorithm (hho.hpp:32-237, utils.hpp:153-174)
// executes per cell, for the degree / quadrature / stabilization combinations of BASELINE.json.  Prints one JSON line per
// combination; `make -C oracle` keeps the binary, tests/test_oracle_flops.py checks oracle/flops_per_cell.json against it.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
static unsigned long long g_add, g_mul, g_div, g_sqrt, g_trans, g_cmp;
struct cd {
    double v;
    cd() {}
    cd(double x) : v(x) {}
    cd(int x) : v(x) {}
    cd(size_t x) : v((double)x) {}
    explicit operator int() const { return (int)v; }
    explicit operator double() const { return v; }
    cd &operator+=(cd o) { ++g_add; v += o.v; return *this; }
    cd &operator-=(cd o) { ++g_add; v -= o.v; return *this; }
    cd &operator*=(cd o) { ++g_mul; v *= o.v; return *this; }
    cd &operator/=(cd o) { ++g_div; v /= o.v; return *this; }
};
static inline cd operator+(cd a, cd b) { ++g_add; return cd(a.v + b.v); }
static inline cd operator-(cd a, cd b) { ++g_add; return cd(a.v - b.v); }
static inline cd operator*(cd a, cd b) { ++g_mul; return cd(a.v * b.v); }
static inline cd operator/(cd a, cd b) { ++g_div; return cd(a.v / b.v); }
static inline cd operator-(cd a) { return cd(-a.v); }
static inline bool operator>(cd a, cd b) { ++g_cmp; return a.v > b.v; }
static inline bool operator<(cd a, cd b) { ++g_cmp; return a.v < b.v; }
static inline bool operator>=(cd a, cd b) { ++g_cmp; return a.v >= b.v; }
static inline bool operator<=(cd a, cd b) { ++g_cmp; return a.v <= b.v; }
static inline bool operator==(cd a, cd b) { return a.v == b.v; }
static inline bool operator!=(cd a, cd b) { return a.v != b.v; }
static inline cd cd_sqrt(cd a) { ++g_sqrt; return cd(sqrt(a.v)); }
static inline cd cd_fabs(cd a) { return cd(fabs(a.v)); }
static inline cd cd_sin(cd a) { ++g_trans; return cd(sin(a.v)); }
static inline cd cd_fmax(cd a, cd b) { ++g_cmp; return cd(fmax(a.v, b.v)); }
#define sqrt cd_sqrt
#define fabs cd_fabs
#define sin cd_sin
#define fmax cd_fmax
#define double cd
#define HHO_FLOPCOUNT 1
extern "C" {
#include "hho_oracle.c"
}
#undef double
#undef sqrt
#undef fabs
#undef sin
#undef fmax
int main(int argc, char **argv)
{
    int cfgs[][4] = {{2,1,0,2},{2,1,1,1},{3,2,1,1},{3,2,0,2},{0,1,0,2},{4,3,0,2},{3,3,0,2}};
    hho_mesh_params mp; mp.Nx = 4; mp.Ny = 4; mp.min_x = 0; mp.max_x = 1; mp.min_y = 0; mp.max_y = 1;
    size_t np = hho_mesh_num_points(&mp), nc = hho_mesh_num_cells(&mp);
    cd *pts = (cd*)malloc(sizeof(cd)*2*np); uint64_t *ids = (uint64_t*)malloc(8*4*nc);
    hho_mesh_generate(&mp, pts, ids);
    for (auto &c : cfgs) {
        hho_degrees di = hho_degree_info2(c[0], c[1], 0);
        int ms = hho_cell_basis_size(di.cell_deg) + 4*hho_face_basis_size(di.face_deg);
        cd *lc = (cd*)malloc(sizeof(cd)*ms*ms*nc), *rhs=(cd*)malloc(sizeof(cd)*32*nc);
        g_add=g_mul=g_div=g_sqrt=g_trans=g_cmp=0;
        int st = hho_local_ops_batch(pts, ids, 0, nc, di, c[2], c[3], 0, 0, 0, 0,0,0, lc, 0);
        unsigned long long ops_add=g_add, ops_mul=g_mul, ops_div=g_div, ops_sqrt=g_sqrt;
        g_add=g_mul=g_div=g_sqrt=g_trans=0;
        hho_local_ops_batch(pts, ids, 0, nc, di, c[2], c[3], hho_builtin_fn(c[0]==0?3:1), 0, c[0]==0?1:0, 0,0,0, lc, rhs);
        printf("{\"cd\":%d,\"fd\":%d,\"quad\":%d,\"stab\":%d,\"status\":%d,\"ops\":{\"add\":%.1f,\"mul\":%.1f,\"div\":%.1f,\"sqrt\":%.1f},"
               "\"ops_with_rhs\":{\"add\":%.1f,\"mul\":%.1f,\"div\":%.1f,\"sqrt\":%.1f,\"sin_etc\":%.1f}}\n", c[0],c[1],c[2],c[3],st,
               (double)ops_add/nc,(double)ops_mul/nc,(double)ops_div/nc,(double)ops_sqrt/nc,
               (double)g_add/nc,(double)g_mul/nc,(double)g_div/nc,(double)g_sqrt/nc,(double)g_trans/nc);
    }
}
