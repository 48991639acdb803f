// rm_math_pow.h -- pow(x, y), bit-exact restatement of glibc 2.35 __pow_fma, branch-free.
//
// Algorithm: glibc sysdeps/ieee754/dbl-64/e_pow.c (Szabolcs Nagy's pow from ARM
// optimized-routines): log(x) as a double-double from a 128-entry table and a
// degree-7 polynomial, y*log(x) as a double-double, exp() from a 128-entry 2^(i/128)
// table and a degree-5 polynomial.  The x86-64 multiarch build compiles that file
// with -mfma -mavx2 (so __FP_FAST_FMA paths are taken and gcc contracts a*b+c);
// the operation order and the exact set of fused operations below were taken
// from the machine code of the FMA variant, which is the one CPython reaches on
// every FMA+AVX2 host (the build container and the GPU box alike).
//
// Shape: straight-line code, special cases resolved by selects at the end, so a
// 64-lane wavefront never diverges inside pow and independent calls can be
// interleaved by the instruction scheduler (the kernels are latency-bound on the
// longest ray -- DESIGN.md).
//
// STATUS: EXACT for x >= +0 (zero, subnormal, normal, +inf), NaN, and y > 0 with
// 2^-65 <= y < 2^63 whose result neither overflows nor is subnormal -- every call site of
// the path: `** 0.5` (vec3.py:46-47, primitives.py:24-32,49-50), `** 2` (primitives.py:26),
// `r ** 7.0`, `r ** 8.0` (catalog.py:280,283).  Not claimed: negative x (NaN is returned;
// CPython returns a complex number or raises), results below 2^-1022 (0 is returned;
// CPython raises OverflowError for a subnormal result), overflow (+inf).
#pragma once

#include "rm_tables.h"

namespace rm {

RM_MATH_HD uint64_t rm_asuint64(double x) { return __builtin_bit_cast(uint64_t, x); }
RM_MATH_HD double rm_asdouble(uint64_t u) { return __builtin_bit_cast(double, u); }

// log(x) for positive normal-range bits `ix`, returned as hi + *tail (log_inline, e_pow.c).
RM_MATH_HD double rm_pow_log_inline(uint64_t ix, double* tail)
{
    const double Ln2hi = rm_pow_log_head[0], Ln2lo = rm_pow_log_head[1];
    const double A0 = rm_pow_log_head[2], A1 = rm_pow_log_head[3], A2 = rm_pow_log_head[4],
                 A3 = rm_pow_log_head[5], A4 = rm_pow_log_head[6], A5 = rm_pow_log_head[7],
                 A6 = rm_pow_log_head[8];
    // x = 2^k z; where z is in range [OFF, 2*OFF) and exact.
    uint64_t tmp = ix - 0x3fe6955500000000ull;
    int i = (int)((tmp >> 45) & 127);
    int k = (int)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    double z = rm_asdouble(iz);
    double kd = (double)k;
    double invc = tab_pow_log(i, 0);
    double logc = tab_pow_log(i, 1);
    double logctail = tab_pow_log(i, 2);

    double r = rm_fma(z, invc, -1.0);
    // k*Ln2 + log(c) + r.
    double t1 = rm_fma(kd, Ln2hi, logc);
    double t2 = t1 + r;
    double lo1 = rm_fma(kd, Ln2lo, logctail);
    double lo2 = (t1 - t2) + r;
    double ar = A0 * r;
    double ar2 = r * ar;
    double ar3 = r * ar2;
    // k*Ln2 + log(c) + r + A[0]*r*r.
    double hi = t2 + ar2;
    double lo3 = rm_fma(ar, r, -ar2);
    double lo4 = (t2 - hi) + ar2;
    // p = log1p(r) - r - A[0]*r*r.
    double p12 = rm_fma(r, A2, A1);
    double p34 = rm_fma(r, A4, A3);
    double p56 = rm_fma(r, A6, A5);
    double q = rm_fma(p56, ar2, p34);
    double pp = rm_fma(ar2, q, p12);
    double lo = rm_fma(ar3, pp, ((lo1 + lo2) + lo3) + lo4);
    double y = hi + lo;
    *tail = (hi - y) + lo;
    return y;
}

// bits of x brought into the normal range (glibc: ix = asuint64(x * 0x1p52) & ~sign; ix -= 52 << 52)
RM_MATH_HD uint64_t rm_pow_norm_bits(double x)
{
    uint64_t ix = rm_asuint64(x);
    uint64_t sx = (rm_asuint64(x * 0x1p52) & 0x7fffffffffffffffull) - (52ull << 52);
    return (ix < 0x0010000000000000ull) ? sx : ix;
}

// exp(ehi + elo) for the double-double y*log(x) (exp_inline, e_pow.c); `bad` results are fixed by the caller
RM_MATH_HD double rm_pow_exp_inline(double ehi, double elo)
{
    const double InvLn2N = rm_exp_head[0], Shift = rm_exp_head[1], NegLn2hiN = rm_exp_head[2],
                 NegLn2loN = rm_exp_head[3], C2 = rm_exp_head[4], C3 = rm_exp_head[5],
                 C4 = rm_exp_head[6], C5 = rm_exp_head[7];
    double kd = rm_fma(ehi, InvLn2N, Shift);
    uint64_t ki = rm_asuint64(kd);
    kd -= Shift;
    double r = rm_fma(kd, NegLn2hiN, ehi);
    r = rm_fma(kd, NegLn2loN, r);
    r = elo + r;
    const int idx = (int)(ki & 127);
    uint64_t top = ki << 45;
    double tail = rm_asdouble(tab_exp(idx, 0));
    uint64_t sbits = tab_exp(idx, 1) + top;
    double r2 = r * r;
    double a = rm_fma(r, C3, C2);
    double b = rm_fma(r, C5, C4);
    double c = rm_fma(a, r2, r + tail);
    double r4 = r2 * r2;
    double tmp = rm_fma(b, r4, c);
    double scale = rm_asdouble(sbits);
    double res = rm_fma(tmp, scale, scale);
    // |y log x| < 2^-54: 1 + ehi;  |y log x| >= 512: the result leaves the normal range
    uint32_t abstop = (uint32_t)(rm_asuint64(ehi) >> 52) & 0x7ff;
    const double tiny = 1.0 + ehi;
    const double out_of_range = (ehi < 0.0) ? 0.0 : __builtin_inf();
    res = (abstop < 0x3c9) ? tiny : res;
    res = (abstop >= 0x409) ? out_of_range : res;
    return res;
}

RM_MATH_HD double rm_pow_fix_special(double x, double res)
{
    res = (x == 0.0) ? 0.0 : res;                       // pow(+0, y > 0) = +0
    res = (x == __builtin_inf()) ? x : res;             // pow(+inf, y > 0) = +inf
    res = ((x != x) | (x < 0.0)) ? __builtin_nan("") : res;  // NaN in -> NaN; negative base unclaimed
    return res;
}

// pow(x, y) for x >= 0, y > 0 (see STATUS)
RM_MATH_HD double rm_pow(double x, double y)
{
    double lo;
    double hi = rm_pow_log_inline(rm_pow_norm_bits(x), &lo);
    double ehi = y * hi;
    double elo = rm_fma(y, lo, rm_fma(hi, y, -ehi));
    return rm_pow_fix_special(x, rm_pow_exp_inline(ehi, elo));
}

// pow(x, 0.5) for a wavefront with FEW live lanes: the square root decides.
//   s = sqrt(x) rounded, e = x - s*s (exact in one fma): the true root is s + e / (2 s).  glibc's pow returns the true
//   value of x ** 0.5 with an error below 0.01 ulp BEFORE its final rounding (e_pow.c: exp 0.509 ulp after rounding, the
//   log term contributes |y log x| 2^-15 ulp; measured on 1.4e9 arguments: pow(x, 0.5) != sqrt(x) only where the true
//   root lies within 0.0088 ulp of a rounding midpoint, tests/test_math_exact.py), so it rounds to s whenever the true
//   root is further than that from both midpoints around s.  The guard asks for 1/32 ulp (and a root that is not a
//   power of two, where the spacing changes); it also rejects any s that is not the correctly rounded root, so
//   nothing is assumed about the device's sqrt.  (x ** 2 yields to the same treatment -- product, fma residual -- and
//   was measured: the cylinder scenes got slower, so it is not built.)  Lanes that fail the guard (6.25 % of spread-out
//   arguments) take the full pow -- a WAVE-UNIFORM branch, which is why this form only pays when few lanes are live:
//   the tail of a frame, where a handful of long rays march on in nearly empty wavefronts and every instruction is
//   on the frame's critical chain (30 dependent instructions instead of 130).  Full wavefronts go straight to pow.
// rm_pow_half_guard is the lane-local part, shared with the host check build.
//
// The guard band (round 3: 1/64 ulp, was 1/32).  e_pow.c states its error budget: the exponential is within 0.509 ulp
// AFTER the final rounding, i.e. 0.009 ulp before it, and the logarithm adds |y log x| * 1.3 * 2^-68 relative =
// |y log x| * 1.3 * 2^-15 ulp -- below 0.001 ulp for |y log x| < 25, i.e. every squared length and every square the
// path forms (the guards below also refuse arguments outside 2^-60 .. 2^60).  pow therefore returns the correctly
// rounded value whenever the true value is further than 0.01 ulp from a rounding midpoint; the band asks for 1/64
// (0.0156).  tests/test_math_exact.py: on > 10^8 spread and adversarial (near-midpoint) arguments every accepted value
// equals libm's pow bit for bit, and the arguments on which pow and the rounded root differ all lie within 0.009 ulp
// of a midpoint.
// (Round 3 also built the guarded root -- and the same for x ** 2 -- into the DENSE render kernels of the algebraic
// scenes: an evaluation whose arguments were all proven was used at once, the others waited for a masked exact block run
// every few turns.  Bit-identical frames, but slower everywhere -- Sphere 0.35 -> 0.41 ms, Cube 0.18 -> 0.24, Pillar
// Forest 1.7 -> 2.2: with 64 lanes x 3 % two lanes wait on every turn, so either the exact block runs nearly every turn
// or the waiting lanes idle for a third of a 10-turn ray.  Removed; the guard stays where few lanes are live.)
constexpr double kGuardBand = 1.0 / 64.0;
RM_MATH_HD double rm_pow_half_guard(double x, bool* safe)
{
    const double s = __builtin_sqrt(x);
    const double e = rm_fma(-s, s, x);
    const uint64_t sb = rm_asuint64(s);
    const double u = rm_asdouble((sb & 0x7ff0000000000000ull) - (52ull << 52));     // ulp(s); x is far inside the normal range
    // the true root is s + e / (2 s): at least kGuardBand ulp away from both midpoints  <=>  |e| <= (1 - 2 band) s u
    // x = +0 (a point inside a box's slabs: length of the zero vector): pow(+0, 0.5) = +0, as the square root
    const bool proven = (rm_fabs(e) < (1.0 - 2.0 * kGuardBand) * (s * u)) & ((sb & 0x000fffffffffffffull) != 0) & (x > 0x1p-60) & (x < 0x1p60);
    *safe = (rm_asuint64(x) == 0ull) ? true : proven;
    return s;
}
constexpr int kSparseLanes = 16;      // live lanes up to which the short forms are tried (fallback odds 1 - 0.969^n: 40 % at 16)
template <bool SPARSE>
RM_MATH_HD double rm_pow_half(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (SPARSE) {
        if (__popcll(__ballot(true)) <= kSparseLanes) {        // live lanes of this wave (wave-uniform)
            bool safe;
            double r = rm_pow_half_guard(x, &safe);
            if (__any(!safe)) {
                const double full = rm_pow(x, 0.5);
                r = safe ? r : full;
            }
            return r;
        }
    }
#endif
    return rm_pow(x, 0.5);
}
// pow(x, ya) and pow(x, yb) sharing the one log(x) they both start from (catalog.py:280,283:
// r ** 7.0 and r ** 8.0 of the same r); each result is bit-identical to a separate rm_pow call.
RM_MATH_HD void rm_pow2(double x, double ya, double yb, double* ra, double* rb)
{
    double lo;
    double hi = rm_pow_log_inline(rm_pow_norm_bits(x), &lo);
    double ehia = ya * hi, ehib = yb * hi;
    double eloa = rm_fma(ya, lo, rm_fma(hi, ya, -ehia));
    double elob = rm_fma(yb, lo, rm_fma(hi, yb, -ehib));
    *ra = rm_pow_fix_special(x, rm_pow_exp_inline(ehia, eloa));
    *rb = rm_pow_fix_special(x, rm_pow_exp_inline(ehib, elob));
}

}  // namespace rm
