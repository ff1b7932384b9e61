// Issue rate of VALU opcodes on gfx950, relative to v_add_u32 (4 cycles per wave64 instruction): one wavefront per SIMD,
// 8 independent chains per lane, wall time by HIP events.   hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, int n, unsigned seed)
{
    unsigned a[8]; unsigned long long q[8]; double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 8 + i; q[i] = a[i]; d[i] = 1.0 + 1e-9 * a[i]; }
    unsigned m = seed | 3; double dm = 1.0000001; const unsigned long long smask = __ballot(threadIdx.x & 1);
    for (int it = 0; it < n; ++it) {
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(m) : "vcc");
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dm));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(q[i]) : "v"(q[(i + 1) & 7]));
#define MOVDPP(i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(q[i]) : "v"(q[(i + 1) & 7]));
#define FMADPP(i) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(d[(i + 4) & 7]), "v"(dm));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
#define RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
#define RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define CVT(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(smask));
#define CND0(i) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[i]) : "s"(smask));
#define NOP1(i) asm volatile("s_nop 0");
        if (OP == 0) { REP8(ADD) } else if (OP == 1) { REP8(MULLO) } else if (OP == 2) { REP8(MULHI) } else if (OP == 3) { REP8(MUL24) }
        else if (OP == 4) { REP8(MAD24) } else if (OP == 5) { REP8(MAD64) } else if (OP == 6) { REP8(FMA64) } else if (OP == 7) { REP8(CND) }
        else if (OP == 8) { REP8(MOV64) } else if (OP == 9) { REP8(MOVDPP) } else if (OP == 10) { REP8(FMADPP) } else if (OP == 11) { REP8(LSHLADD) }
        else if (OP == 12) { REP8(ADD3) } else if (OP == 13) { REP8(LSHLADD64) } else if (OP == 14) { REP8(MUL64) } else if (OP == 15) { REP8(ADD64) }
        else if (OP == 16) { REP8(RSQ64) } else if (OP == 17) { REP8(RCP64) } else if (OP == 18) { REP8(CVT) }
        else if (OP == 19) { REP8(CNDS) } else if (OP == 20) { REP8(CND0) } else if (OP == 21) { REP8(NOP1) }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (unsigned)q[i] + (unsigned)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static int W = 1;
template <int OP> float run(unsigned *out, int n)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<256 * W, 256>>>(out, n / 10, 1);
    hipEventRecord(e0); k<OP><<<256 * W, 256>>>(out, n, 1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    unsigned *out; hipMalloc(&out, 8 * 256 * 256 * 4);
    const int n = 200000;
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_u64_u32", "v_fma_f64", "v_cndmask_b32",
                           "v_mov_b64", "v_mov_b64_dpp", "v_fmac_f64_dpp", "v_lshl_add_u32", "v_add3_u32", "v_lshl_add_u64", "v_mul_f64", "v_add_f64",
                           "v_rsq_f64", "v_rcp_f64", "v_cvt_f64_i32", "v_cndmask_e64 s", "v_cndmask 0,s", "s_nop 0"};
    for (W = 1; W <= 4; W *= 2) {
    printf("== %d wavefront(s) per SIMD\n", W);
    float t[22];
    t[0] = run<0>(out, n); t[1] = run<1>(out, n); t[2] = run<2>(out, n); t[3] = run<3>(out, n); t[4] = run<4>(out, n); t[5] = run<5>(out, n);
    t[6] = run<6>(out, n); t[7] = run<7>(out, n); t[8] = run<8>(out, n); t[9] = run<9>(out, n); t[10] = run<10>(out, n); t[11] = run<11>(out, n);
    t[12] = run<12>(out, n); t[13] = run<13>(out, n); t[14] = run<14>(out, n); t[15] = run<15>(out, n); t[16] = run<16>(out, n); t[17] = run<17>(out, n);
    t[18] = run<18>(out, n); t[19] = run<19>(out, n); t[20] = run<20>(out, n); t[21] = run<21>(out, n);
    for (int i = 0; i < 22; ++i)
        printf("%-16s %8.3f ms  %5.2f x v_add_u32  (%.2f ns per SIMD per wave instruction)\n", names[i], t[i], t[i] / t[0], t[i] * 1e6 / (n * 8.0 * W));
    }
    return 0;
}
