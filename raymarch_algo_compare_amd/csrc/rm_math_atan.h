// rm_math_atan.h -- acos and atan2 (catalog.py:277-278), restating glibc 2.35.
//
// acos: sysdeps/ieee754/dbl-64/e_asin.c (__ieee754_acos, IBM Accurate Mathematical Library
// after the 2.33 removal of the slow paths): eight magnitude bands, a Taylor band below 1/8,
// five table bands on asncs (centre point, degree 5..9 polynomial, asin value) and a
// 1/sqrt band (inroot seed + one Newton step + double-double correction) up to 1.
// atan2: sysdeps/ieee754/dbl-64/e_atan2.c (__ieee754_atan2 after the 2.34 removal of the slow
// paths): u = min/max as a double-double quotient, degree-13 series below 1/16, otherwise the
// 241-row cij table, four quadrant cases.
// Operation order and the fused operations are those of the x86-64 FMA variants' machine code.
//
// STATUS: rm_acos EXACT on [-1, 1] (NaN outside); rm_atan2 EXACT for every finite pair, zeros and
// the huge-ratio shortcuts included (for an exponent gap >= 57 with |x| or |y| outside
// [2^-500, 2^500] the quotient is formed after glibc's rescaling: identical unless it is subnormal).
// NaN / infinite arguments of atan2 return NaN (not reachable from the path).
#pragma once

namespace rm {

struct AtanK {
    static constexpr double hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54;
    static constexpr double opi = 0x1.921fb54442d18p+1, opi1 = 0x1.1a62633145c07p-53;
    static constexpr double d3 = -0x1.5555555555555p-2, d5 = 0x1.99999999997fdp-3, d7 = -0x1.24924923f7603p-3,
                            d9 = 0x1.c71c6e5129a3bp-4, d11 = -0x1.7458022b13c25p-4, d13 = 0x1.375f08b31cbcep-4;
    static constexpr double f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4, f3 = 0x1.6db6dae42c0e4p-5,
                            f4 = 0x1.f1c7e04f4ad99p-6, f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6;
    static constexpr double rt0 = 0x1.fffffffecc1ddp-1, rt1 = 0x1.fffffff757304p-2, rt2 = 0x1.800496769c91ap-2,
                            rt3 = 0x1.4006318d1dab9p-2;
    static constexpr double t27 = 0x1p27;
};

// ---- e_asin.c: __ieee754_acos -----------------------------------------------------------
//
// Branch-free over the common bands: the Taylor band (|x| < 1/8) and ONE unified table band are
// always evaluated and selected.  The five table bands of e_asin.c differ only in row stride and
// polynomial degree (5..9); a band of degree D is evaluated here as a degree-9 Horner chain whose
// leading coefficients are +0 (stored that way in the re-laid-out table, tools/extract_libm_tables.py)
// -- fma(xx, +0, c) == c exactly, so the value is bit-identical to the shorter chain.  The 1/sqrt band (|x| >= 0.96875) is evaluated as well and selected: no branch anywhere.

RM_MATH_HD double rm_acos_sqrt_band(double x, int m)          // 0.96875 <= |x| < 1
{
    typedef AtanK K;
    const double zp = 1.0 - x, zn = x + 1.0;
    double z = ((m > 0) ? zp : zn) * 0.5;
    const int kz = (int)(rm_asuint64(z) >> 32);
    // powtwo[511 - (kz >> 21)] = 2^(511 - (kz >> 21)); masked so that out-of-band arguments (the
    // routine is evaluated for every x and selected afterwards) still form a finite double
    double t = rm_inroot[(kz >> 14) & 0x7f] * rm_asdouble((uint64_t)((1023 + 511 - (kz >> 21)) & 0x7ff) << 52);
    double r = rm_fnma(t * t, z, 1.0);
    double q = rm_fma(r, K::rt3, K::rt2);
    q = rm_fma(r, q, K::rt1);
    q = rm_fma(r, q, K::rt0);
    t = q * t;
    double c = z * t;
    double h = rm_fnma(c, t * 0.5, 1.5);
    double y = rm_fnma(K::t27, c, rm_fma(c, K::t27, c));
    double ty = rm_fma(h, c, y);                             // t + y with t = c*(1.5 - 0.5*t*c)
    double cc = rm_fnma(y, y, z) / ty;
    double p = rm_fma(z, K::f6, K::f5);
    p = rm_fma(z, p, K::f4);
    p = rm_fma(z, p, K::f3);
    p = rm_fma(z, p, K::f2);
    p = rm_fma(z, p, K::f1);
    double pz = (p * z) * (y + cc);
    const double res_neg = ((K::hpi1 - cc) - pz) + (K::hpi - y);
    const double res_pos = (cc + pz) + y;
    const double res = (m < 0) ? res_neg : res_pos;
    return res + res;
}

template <bool U = false>
RM_MATH_HD double rm_acos(double x)
{
    typedef AtanK K;
    const uint64_t bits = rm_asuint64(x);
    const int m = (int)(bits >> 32);
    const int k = m & 0x7fffffff;
    const bool pos = m > 0;
    const double ax = rm_fabs(x);

    // |x| < 1/8: Taylor
    double res_taylor = 0.0;
    if (rm_band_needed<U>(k < 0x3fc00000)) {
        const double x2 = x * x;
        double pt = rm_fma(x2, K::f6, K::f5);
        pt = rm_fma(x2, pt, K::f4);
        pt = rm_fma(x2, pt, K::f3);
        pt = rm_fma(x2, pt, K::f2);
        pt = rm_fma(x2, pt, K::f1);
        const double rt = K::hpi - x;
        res_taylor = rt + rm_fnma(pt, x * x2, ((K::hpi - rt) - x) + K::hpi1);
    }

    // 1/8 <= |x| < 0.96875: table band.  rm_asncs holds 216 uniform rows of 13 doubles (the generator
    // pads the shorter polynomials of e_asin.c's lower bands with zero leading coefficients):
    //   rows   0.. 31  |x| < 0.25 (step 2^-5 in the high word's bits 15..19), rows 32..95  |x| < 0.5,
    //   rows 96..215   0.5 <= |x| < 0.96875, index (k >> 13) & 0x7f
    double res_band = 0.0;
    if (rm_band_needed<U>(k >= 0x3fc00000 && k < 0x3fef0000)) {
        const int r1 = (k >> 15) & 0x1f, r2 = 32 + ((k >> 14) & 0x3f), r3 = 96 + ((k >> 13) & 0x7f);
        int row = (k >= 0x3fd00000) ? r2 : r1;
        row = (k >= 0x3fe00000) ? r3 : row;
        row = (row > 215) ? 215 : row;                 // keeps the gather in bounds for out-of-band arguments
        const double* a = rm_asncs + 13 * row;
        const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5], a6 = a[6], a7 = a[7], a8 = a[8],
                     a9 = a[9], a10 = a[10], c0 = a[11], cv = a[12];
        const double xx = ax - a0;
        double p = rm_fma(xx, a10, a9);                // degree-9 Horner; zero-padded coefficients leave the value unchanged
        p = rm_fma(xx, p, a8);
        p = rm_fma(xx, p, a7);
        p = rm_fma(xx, p, a6);
        p = rm_fma(xx, p, a5);
        p = rm_fma(xx, p, a4);
        p = rm_fma(xx, p, a3);
        p = rm_fma(xx, p, a2);
        p = rm_fma(xx * xx, p, c0);
        const double t = rm_fma(xx, a1, p);
        const double yb = pos ? (K::hpi - cv) : (cv + K::hpi);
        const double tb = pos ? (K::hpi1 - t) : (t + K::hpi1);
        res_band = tb + yb;
    }

    double res = (k < 0x3fc00000) ? res_taylor : res_band;
    res = (k < 0x3c880000) ? K::hpi : res;                        // |x| < 2^-55
    // |x| >= 0.96875: 1/sqrt band, |x| == 1, invalid -- without U evaluated unconditionally and selected, so
    // the routine is one straight-line block (the z argument is clamped to stay in the table)
    double res_sqrt = 0.0;
    if (rm_band_needed<U>(k >= 0x3fef0000 && k < 0x3ff00000)) res_sqrt = rm_acos_sqrt_band(x, m);
    res = (k >= 0x3fef0000) ? res_sqrt : res;
    const double at_one = pos ? 0.0 : K::opi;                                    // |x| == 1
    const bool is_one = (k == 0x3ff00000) & ((uint32_t)bits == 0);
    const double beyond = is_one ? at_one : __builtin_nan("");                   // |x| > 1 or NaN: invalid
    res = (k >= 0x3ff00000) ? beyond : res;
    return res;
}

// ---- e_atan2.c: __ieee754_atan2 ------------------------------------------------------------
//
// Branch-free main path: the quotient u = min/max as a double-double, then the degree-13 series
// (u < 1/16) and the cij-table form are both evaluated for the two structural cases -- (i) x > 0,
// |y| < |x| and the three "pi/2 or pi plus/minus atan" cases, which differ only in a base constant
// and a sign (a - b == a + (-b) bit-for-bit) -- and one result is selected; zeros and exponent
// gaps >= 57 are selects as well (no branch anywhere).  NaN / infinite operands yield NaN.

RM_MATH_HD double rm_atan_series(double v)   // d3 + v*(d5 + v*(d7 + v*(d9 + v*(d11 + v*d13))))
{
    typedef AtanK K;
    double p = rm_fma(v, K::d13, K::d11);
    p = rm_fma(v, p, K::d9);
    p = rm_fma(v, p, K::d7);
    p = rm_fma(v, p, K::d5);
    return rm_fma(v, p, K::d3);
}

template <bool U = false>
RM_MATH_HD double rm_atan2(double y, double x)
{
    typedef AtanK K;
    const uint32_t ux = (uint32_t)(rm_asuint64(x) >> 32), uy = (uint32_t)(rm_asuint64(y) >> 32);
    const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);

    double ax = rm_fabs(x), ay = rm_fabs(y);
    const double up = ((ax < 0x1p-500) | (ay < 0x1p-500)) ? 0x1p500 : 1.0;
    ax *= up; ay *= up;
    const double dn = ((ax > 0x1p500) | (ay > 0x1p500)) ? 0x1p-500 : 1.0;
    ax *= dn; ay *= dn;

    const bool y_lt_x = ay < ax;
    const double num = y_lt_x ? ay : ax, den = y_lt_x ? ax : ay;
    const double u = num / den;
    const double vq = den * u;
    const double du = ((num - vq) - rm_fma(den, u, -vq)) / den;

    const bool xpos = x > 0.0;
    const bool case_i = xpos & y_lt_x;
    // cases (ii) x>0,|x|<=|y|: pi/2 - atan;  (iii) x<0,|x|<|y|: pi/2 + atan;  (iv) x<0,|y|<=|x|: pi - atan
    const bool case_iv = !xpos & !(ax < ay);
    const bool plus = !xpos & (ax < ay);
    const double B = case_iv ? K::opi : K::hpi, B1 = case_iv ? K::opi1 : K::hpi1;

    const bool small = u < 0.0625;
    // series forms (u < 1/16)
    double z_is = 0.0, z_os = 0.0;
    if (rm_band_needed<U>(small)) {
        const double v2 = u * u;
        const double ser = rm_atan_series(v2);
        const double uv = u * v2;
        z_is = u + rm_fma(uv, ser, du);
        const double su = plus ? u : -u, sdu = plus ? du : -du;
        const double zz_s = uv * ser;
        const double t2 = B + su;
        const double cor = (B - t2) + su;
        z_os = (((cor + B1) + sdu) + (plus ? zz_s : -zz_s)) + t2;
    }

    // table forms
    double z_it = 0.0, z_ot = 0.0;
    if (rm_band_needed<U>(!small)) {
        int i = (int)(rm_fma(u, 256.0, 0x1p52) - 0x1p52) - 16;
        i = (i < 0) ? 0 : ((i > 240) ? 240 : i);      // in range whenever the table form is selected
        const double* c = rm_cij + 7 * i;
        const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6];
        const double t3 = u - c0;
        // (i): EADD(t3, du, v, dv), zz = v*c2 + (dv*c2 + v*v*(c3 + v*(c4 + v*(c5 + v*c6))))
        if (rm_band_needed<U>(!small & case_i)) {
            const double vi = du + t3;
            const double dv = (rm_fabs(t3) > rm_fabs(du)) ? ((t3 - vi) + du) : ((du - vi) + t3);
            double pi_ = rm_fma(vi, c6, c5);
            pi_ = rm_fma(vi, pi_, c4);
            pi_ = rm_fma(vi, pi_, c3);
            pi_ = (vi * vi) * pi_;
            pi_ = rm_fma(dv, c2, pi_);
            z_it = rm_fma(vi, c2, pi_) + c1;
        }
        // (ii)-(iv): v = (u - c0) + du, zz = B1 -+ v*(c2 + v*(c3 + v*(c4 + v*(c5 + v*c6)))), z = (B -+ c1) + zz
        if (rm_band_needed<U>(!small & !case_i)) {
            const double vo = t3 + du;
            double po = rm_fma(vo, c6, c5);
            po = rm_fma(vo, po, c4);
            po = rm_fma(vo, po, c3);
            po = rm_fma(vo, po, c2);
            z_ot = (B + (plus ? c1 : -c1)) + rm_fma(plus ? vo : -vo, po, B1);
        }
    }

    const double z_i = small ? z_is : z_it, z_o = small ? z_os : z_ot;
    double z = case_i ? z_i : z_o;
    // special operands (e_atan2.c:66-133), resolved by selects in the reference's priority order
    const bool xneg = (ux & 0x80000000u) != 0;
    const double z_tiny = xpos ? u : K::opi;
    z = (de <= -59768832) ? z_tiny : z;                          // |y/x| tiny: y/x itself, or pi
    z = (de >= 59768832) ? K::hpi : z;                           // |y/x| huge
    z = (x == 0.0) ? K::hpi : z;                                 // x = +-0, y != 0
    const double z_y0 = xneg ? K::opi : 0.0;
    z = (y == 0.0) ? z_y0 : z;                                   // y = +-0: +-0 or +-pi by the sign of x
    const bool nonfinite = ((ux & 0x7ff00000u) == 0x7ff00000u) | ((uy & 0x7ff00000u) == 0x7ff00000u);
    z = nonfinite ? __builtin_nan("") : z;                       // inf / nan operands: unclaimed
    return __builtin_copysign(z, y);
}

}  // namespace rm
