// rm_math_trig.h -- sin, cos, log (and acos / atan2, rm_math_atan.h) of the path
// (catalog.py:277-293 Mandelbulb, :510-513 Gyroid), restating glibc 2.35.
//
// sin / cos: glibc sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library,
// after the 2.28 removal of the slow paths): table-driven do_sin / do_cos on
// __sincostab (x = k/128), TAYLOR_SIN below 0.126, 4-part Cody-Waite reduction
// (reduce_sincos) up to |x| < 105414350.  log: sysdeps/ieee754/dbl-64/e_log.c (ARM
// optimized-routines).  As with pow, the x86-64 FMA multiarch build is what CPython
// reaches, so the exact fused/unfused operation sequence below was read from that
// variant's machine code (gcc contracted a*b+c there; every rm_fma is one of its
// vfmadd/vfnmadd/vfmsub instructions, every plain * + - stays unfused).
//
// Shape: branch-free (every range/band is evaluated and the result selected) so a wavefront
// never diverges inside these routines and independent calls interleave -- DESIGN.md section 3.
//
// STATUS: rm_sin / rm_cos / rm_sincos EXACT for |x| < 105414350 (the path needs |x| <= 8*pi for
// Mandelbulb, <= ~320 for Gyroid), NaN for larger or non-finite arguments (unclaimed: glibc uses
// __branred there); rm_log EXACT for positive finite x (normal or subnormal).  Verified against
// libm by tests/test_math_exact.py.
#pragma once

namespace rm {

// ---- s_sin.c ----------------------------------------------------------------------

struct SinCosK {
    static constexpr double big = 0x1.8p45, toint = 0x1.8p52;
    static constexpr double hpinv = 0x1.45f306dc9c883p-1;
    static constexpr double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
    static constexpr double mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27;
    static constexpr double pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
    static constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
    static constexpr double cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
    static constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
                            s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
};

RM_MATH_HD double rm_fnma(double a, double b, double c) { return rm_fma(-a, b, c); }   // -(a*b) + c, one rounding

// Wave-uniform band skipping (template flag U of rm_acos / rm_atan2 / rm_sincos).  The routines evaluate every
// band and select, which is what a full wavefront of unrelated lanes wants (no divergence).  A wavefront TEAM
// (rm_kernels.h) is the opposite case: one wave alone on its SIMD, a handful of live lanes, and every fp64
// instruction it issues costs 8 cycles of the frame's critical chain whether its result is selected or not.
// With U a band is evaluated only if some live lane of the wave selects it (one ballot per band); the selected
// value is computed by the same instructions either way, so results are bit-identical.
template <bool U>
RM_MATH_HD bool rm_band_needed(bool lane_selects)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (U) return __any(lane_selects) != 0;
#endif
    (void)lane_selects;
    return true;
}

// TAYLOR_SIN(xx, x, dx): x + ((POLYNOMIAL(xx)*x - 0.5*dx)*xx + dx)
RM_MATH_HD double rm_taylor_sin(double x, double dx)
{
    typedef SinCosK K;
    double xx = x * x;
    double p = rm_fma(xx, K::s5, K::s4);
    p = rm_fma(xx, p, K::s3);
    p = rm_fma(xx, p, K::s2);
    p = rm_fma(xx, p, K::s1);
    double q = rm_fma(p, x, -(dx * 0.5));
    double t = rm_fma(xx, q, dx);
    return x + t;
}

// table row of __sincostab for u = big + |a| (|a| < 0.855469 on every claimed input; the clamp only
// keeps the gather in bounds for unclaimed / garbage arguments)
RM_MATH_HD int rm_sincos_row(double u)
{
    uint32_t i = (uint32_t)rm_asuint64(u);
    return (int)((i > 109u) ? 0u : i);
}

// do_sin(x, dx), branch-free: Taylor below 0.126, table otherwise (both evaluated, one selected)
template <bool U = false>
RM_MATH_HD double rm_do_sin(double a, double da)
{
    typedef SinCosK K;
    const double aa = rm_fabs(a);
    double taylor = 0.0;
    if (rm_band_needed<U>(aa < 0.126)) taylor = rm_taylor_sin(a, da);   // uses dx before the sign flip, as s_sin.c does
    const double dx = (a <= 0.0) ? -da : da;
    const double u = K::big + aa;
    const double x = aa - (u - K::big);
    const int k = rm_sincos_row(u);
    const double sn = tab_sincos(k, 0), ssn = tab_sincos(k, 1), cs = tab_sincos(k, 2), ccs = tab_sincos(k, 3);
    const double xx = x * x;
    const double s = x + rm_fma(x * xx, rm_fma(xx, K::sn5, K::sn3), dx);
    const double c = rm_fma(x, dx, xx * rm_fma(xx, rm_fma(xx, K::cs6, K::cs4), K::cs2));
    const double cor = rm_fma(s, cs, rm_fnma(c, sn, rm_fma(s, ccs, ssn)));
    const double tab = __builtin_copysign(sn + cor, a);
    return (aa < 0.126) ? taylor : tab;
}

RM_MATH_HD double rm_do_cos(double a, double da)
{
    typedef SinCosK K;
    const double aa = rm_fabs(a);
    const double dx = (a < 0.0) ? -da : da;
    const double u = K::big + aa;
    const double x = (aa - (u - K::big)) + dx;
    const int k = rm_sincos_row(u);
    const double sn = tab_sincos(k, 0), ssn = tab_sincos(k, 1), cs = tab_sincos(k, 2), ccs = tab_sincos(k, 3);
    const double xx = x * x;
    const double s = rm_fma(x * xx, rm_fma(xx, K::sn5, K::sn3), x);
    const double c = xx * rm_fma(xx, rm_fma(xx, K::cs6, K::cs4), K::cs2);
    const double cor = rm_fnma(s, sn, rm_fnma(c, cs, rm_fnma(s, ssn, ccs)));
    return cs + cor;
}

// sin(x) and cos(x) together, branch-free.  __sin and __cos (s_sin.c) pick one of four argument
// ranges; every range ends in do_sin on one (a, da) pair and do_cos on another:
//   |x| < 0.855469          sin = do_sin(x, 0)                         cos = do_cos(x, 0)
//   |x| < 2.426265          sin = copysign(do_cos(hp0-|x|, hp1), x)    cos = do_sin(a, da), a = y+hp1, da = (y-a)+hp1
//   |x| < 105414350         reduce_sincos -> (a, da, n): sin = do_sincos(a, da, n), cos = do_sincos(a, da, n+1)
// so the two kernels are evaluated once each on selected inputs and the outputs are routed by selects.
template <bool U = false>
RM_MATH_HD void rm_sincos(double x, double* sin_out, double* cos_out)
{
    typedef SinCosK K;
    const uint32_t k = (uint32_t)(rm_asuint64(x) >> 32) & 0x7fffffffu;
    const double ax = rm_fabs(x);
    const bool r1 = k < 0x3feb6000u;
    const bool r2 = !r1 & (k < 0x400368fdu);
    // range 3: reduce_sincos.  Range 1 (|x| < 0.855469: no reduction, a = x, da = 0, n = 0) is folded into it by
    // forcing the multiple of pi/2 to zero: every product with xn = 0 vanishes exactly, so b == x and db == +0.
    const double t = rm_fma(x, K::hpinv, K::toint);
    const double xn = r1 ? 0.0 : (t - K::toint);
    const uint32_t n = r1 ? 0u : (uint32_t)rm_asuint64(t);
    const double y = rm_fnma(xn, K::mp2, rm_fnma(xn, K::mp1, x));
    const double t2 = rm_fnma(xn, K::pp3, y);
    double db = rm_fnma(K::pp3, xn, y - t2);
    const double b = rm_fnma(xn, K::pp4, t2);
    db = db + rm_fnma(xn, K::pp4, t2 - b);
    // range 2
    const double tt = K::hp0 - ax;
    const double a2 = tt + K::hp1;
    const double da2 = (tt - a2) + K::hp1;

    // single-level selects only (a nested ?: comes back from the compiler as a branch)
    const double aS = r2 ? a2 : b, daS = r2 ? da2 : db, aC = r2 ? tt : b, daC = r2 ? K::hp1 : db;
    const double dS = rm_do_sin<U>(aS, daS);
    const double dC = rm_do_cos(aC, daC);

    // range 3 (and 1, n = 0) routing: do_sincos(a, da, n) = (n & 1 ? do_cos : do_sin), negated when n & 2
    const uint32_t m = n + 1u;
    double s3 = (n & 1u) ? dC : dS;
    s3 = (n & 2u) ? -s3 : s3;
    double c3 = (m & 1u) ? dC : dS;
    c3 = (m & 2u) ? -c3 : c3;
    const double s2 = __builtin_copysign(dC, x);
    double s = r2 ? s2 : s3, c = r2 ? dS : c3;
    s = (k < 0x3e500000u) ? x : s;                      // |x| < 2^-26
    c = (k < 0x3e400000u) ? 1.0 : c;                    // |x| < 2^-27
    const double bad = __builtin_nan("");                // inf / nan -> nan; __branred range unclaimed
    const bool huge = k >= 0x419921fbu;
    *sin_out = huge ? bad : s;
    *cos_out = huge ? bad : c;
}

RM_MATH_HD double rm_sin(double x)
{
    double s, c;
    rm_sincos(x, &s, &c);
    return s;
}

RM_MATH_HD double rm_cos(double x)
{
    double s, c;
    rm_sincos(x, &s, &c);
    return c;
}

// ---- e_log.c ------------------------------------------------------------------------

RM_MATH_HD double rm_log(double x)
{
    const double Ln2hi = rm_log_head[0], Ln2lo = rm_log_head[1];
    const double* A = rm_log_head + 2;    // poly[5]
    const double* B = rm_log_head + 7;    // poly1[11]
    uint64_t ix = rm_asuint64(x);
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {
        // x is close to 1.0: log1p polynomial with a double-double leading term
        if (ix == 0x3ff0000000000000ull) return 0.0;
        double r = x - 1.0;
        double r2 = r * r;
        double r3 = r * r2;
        double p1 = rm_fma(r2, B[3], rm_fma(r, B[2], B[1]));
        double p2 = rm_fma(r2, B[6], rm_fma(r, B[5], B[4]));
        double p3 = rm_fma(r3, B[10], rm_fma(r2, B[9], rm_fma(r, B[8], B[7])));
        double y = rm_fma(rm_fma(p3, r3, p2), r3, p1);
        double w = rm_fma(r, 0x1p27, r);
        double rhi = rm_fnma(0x1p27, r, w);
        double rlo = r - rhi;
        double rhi2 = rhi * rhi;
        double hi = rm_fma(rhi2, B[0], r);
        double lo = rm_fma(rhi2, B[0], r - hi);
        lo = rm_fma(B[0] * rlo, r + rhi, lo);
        y = rm_fma(y, r3, lo);
        return hi + y;
    }
    uint32_t top = (uint32_t)(ix >> 48);
    if (top - 0x0010 >= 0x7ff0 - 0x0010) {
        if (ix * 2 == 0) return -__builtin_inf();                  // log(+-0) = -inf
        if (ix == 0x7ff0000000000000ull) return x;                 // log(inf) = inf
        if ((top & 0x8000) || (top & 0x7ff0) == 0x7ff0) return (x - x) / (x - x);   // x < 0 or nan
        ix = rm_asuint64(x * 0x1p52);                               // subnormal: normalise
        ix -= 52ull << 52;
    }
    uint64_t tmp = ix - 0x3fe6000000000000ull;
    int i = (int)((tmp >> 45) & 127);
    int k = (int)((int64_t)tmp >> 52);
    uint64_t iz = ix - (tmp & 0xfff0000000000000ull);
    double invc = tab_log(i, 0), logc = tab_log(i, 1);
    double z = rm_asdouble(iz);
    double r = rm_fma(z, invc, -1.0);
    double kd = (double)k;
    double w = rm_fma(kd, Ln2hi, logc);
    double hi = w + r;
    double lo = (w - hi) + r;
    lo = rm_fma(kd, Ln2lo, lo);
    double r2 = r * r;
    double r3 = r * r2;
    lo = rm_fma(r2, A[0], lo);
    double q = rm_fma(rm_fma(r, A[4], A[3]), r2, rm_fma(r, A[2], A[1]));
    double y = rm_fma(r3, q, lo);
    return y + hi;
}

}  // namespace rm

#include "rm_math_atan.h"
