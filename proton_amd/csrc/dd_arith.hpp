// dd_arith.hpp -- double-double arithmetic (an unevaluated sum hi + lo of two doubles, |lo| <= ulp(hi) / 2: about 106 bits) for the
// few places where the conditioning of the problem, not the kernel, decides how many digits survive: the Nitsche-penalised
// reconstruction system of a sliver cut cell (cut_device.hpp; 1-norm condition numbers to 1.9e9 on the 512 x 512 mesh of
// BASELINE.json's config 3).  Error-free transformations of Dekker / Knuth with the hardware FMA; the "accurate" addition (two
// two-sums), so that cancellation between large terms keeps its low part.  No -ffast-math: the compiler must not reassociate --
// and no CONTRACTION inside these functions (#pragma clang fp contract(off)): hipcc's default -ffp-contract=fast fuses the rounded
// product p = a * b of a two_prod with the addition that follows it (p + e -> fma(a, b, e)), after which the "error" term no longer
// is the error of anything: on the device dd_mul was accurate to 1e-16, not 1e-32 (tools/probe/dd_check.hip found it).
#pragma once
#include <hip/hip_runtime.h>

namespace pa {

struct dd { double hi, lo; };

__device__ __forceinline__ dd dd_from(double a) { return dd{a, 0.0}; }
__device__ __forceinline__ dd two_sum(double a, double b)
{
#pragma clang fp contract(off)
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd quick_two_sum(double a, double b)      // |a| >= |b|
{
#pragma clang fp contract(off)
    const double s = a + b;
    return dd{s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b)
{
#pragma clang fp contract(off)
    const double p = a * b;
    return dd{p, __builtin_fma(a, b, -p)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b)
{
#pragma clang fp contract(off)
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
// the cheaper addition (one two-sum): absolute error <= 2^-104 (|a| + |b|) -- what a long accumulation of terms of one scale needs
// (11 operations instead of 20); the accurate form above where a result that has cancelled is divided by or multiplied on
__device__ __forceinline__ dd dd_add_fast(dd a, dd b)
{
#pragma clang fp contract(off)
    dd s = two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
__device__ __forceinline__ dd dd_sub_fast(dd a, dd b) { return dd_add_fast(a, dd{-b.hi, -b.lo}); }
__device__ __forceinline__ dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
__device__ __forceinline__ dd dd_mul(dd a, dd b)
{
#pragma clang fp contract(off)
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b)
{
#pragma clang fp contract(off)
    dd p = two_prod(a.hi, b);
    p.lo = __builtin_fma(a.lo, b, p.lo);
    return quick_two_sum(p.hi, p.lo);
}
// 1 / sqrt(a): a double seed and Newton steps x <- x + x (1 - a x^2) / 2 carried out in double-double.  Each step squares the
// relative error (x 1.5): TWO steps, so that the result does not depend on how good the seed is -- with `1.0 / sqrt(a.hi)` as the
// seed and one step the results on the device carried ~1e-20, the square of a 1e-10 seed: the compiler had taken the expression for
// the hardware's approximate reciprocal square root (measured: errors of data at 1e-19 x condition number; two steps: 1e-32).
__device__ __forceinline__ dd dd_rsqrt(dd a)
{
#pragma clang fp contract(off)
    double x = __builtin_amdgcn_rsq(a.hi);
    {
        const double t = a.hi * x, e = __builtin_fma(-t, x, 1.0);      // one step in double first (seed ~ 2^-26 -> ~2^-50)
        x = __builtin_fma(0.5 * x, e, x);
    }
    dd y = dd_from(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const dd r = dd_sub(dd_from(1.0), dd_mul(dd_mul(a, y), y));
        y = dd_add(y, dd_mul(dd_mul_d(r, 0.5), y));
    }
    return y;
}
// The same from a seed that is already a correctly rounded double (relative error e0 <= 2^-50): ONE Newton step whose correction
// y0 (1 - a y0^2) / 2 needs only double accuracy -- 1 - a y0^2 is formed exactly (its high part cancels by Sterbenz) and is ~e0, so
// rounding it to double leaves 2^-53 e0 ~ 1e-31, the size of the step's own quadratic term 3/8 e0^2.  25 dependent operations
// instead of the ~140 of two full double-double steps: the pivot chain of the cut cells' factorization is made of these.
__device__ __forceinline__ dd dd_rsqrt_1(dd a)
{
#pragma clang fp contract(off)
    double x = __builtin_amdgcn_rsq(a.hi);
    {
        const double t = a.hi * x, e = __builtin_fma(-t, x, 1.0);
        x = __builtin_fma(0.5 * x, e, x);
    }
    {   // a second step in double: the hardware seed is good to ~2^-26 only on paper; this makes e0 <= 2^-50 whatever it was
        const double t = a.hi * x, e = __builtin_fma(-t, x, 1.0);
        x = __builtin_fma(0.5 * x, e, x);
    }
    const dd t = dd_mul_d(dd_mul_d(a, x), x);            // a x^2 = 1 - r
    const double r = (1.0 - t.hi) - t.lo;
    return quick_two_sum(x, (0.5 * x) * r);
}
// ---- cross-lane helpers (wave64) ----------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane)      // `lane` wave-uniform: the value lands in scalar registers
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, lane), hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ dd dd_readlane(dd v, int lane) { return dd{readlane_f64(v.hi, lane), readlane_f64(v.lo, lane)}; }
__device__ __forceinline__ dd dd_shfl_xor(dd v, int off) { return dd{__shfl_xor(v.hi, off), __shfl_xor(v.lo, off)}; }

// Sum over the 64 lanes of N values per lane, ONE result per lane pair instead of every result on every lane: at each level of the
// butterfly a lane keeps one half of its values and sends the other half to its partner (N/2, N/4, ... exchanges instead of N per
// level: 29 instead of 168 for the 28 moments of k = 2).  On return `out` is the full sum of entry `index` (the return value) --
// valid only if `ok`: the halves of an odd count are padded.
template <int N, int OFF, typename T, typename Add, typename Shfl>
__device__ __forceinline__ int lanes_transpose_reduce(const T (&v)[N], int lane, T zero, Add add, Shfl shfl, T &out, bool &ok)
{
    if constexpr (N == 1) {
        T s = v[0];
#pragma unroll
        for (int off = OFF; off >= 1; off >>= 1) s = add(s, shfl(s, off));
        out = s;
        return 0;
    } else {
        static_assert(OFF >= 1, "more values than lanes");
        constexpr int H = (N + 1) / 2;
        const bool up = (lane & OFF) != 0;
        T w[H];
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const T lo = v[m], hi = m + H < N ? v[m + H < N ? m + H : 0] : zero;
            const T send = up ? lo : hi, keep = up ? hi : lo;
            w[m] = add(keep, shfl(send, OFF));
        }
        const int m = lanes_transpose_reduce<H, OFF / 2>(w, lane, zero, add, shfl, out, ok);
        if (up && m + H >= N) ok = false;
        return m + (up ? H : 0);
    }
}

__device__ __forceinline__ double dd_round(dd a) { return a.hi + a.lo; }
__device__ __forceinline__ dd dd_load(const double *p) { return dd{p[0], p[1]}; }
__device__ __forceinline__ void dd_store(double *p, dd v) { p[0] = v.hi; p[1] = v.lo; }

}  // namespace pa
