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
// STATUS: rm_acos EXACT on [-1, 1]; rm_atan2 EXACT for finite arguments whose exponents differ
// by < 57*2^20 ... i.e. every finite pair (the huge-ratio shortcuts are included), zeros included.
// NaN / infinity arguments take the platform fallback (not reachable from the path).
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

// one table band: row at asncs[n], polynomial degree DEG in xx = |x| - centre
template <int DEG>
RM_MATH_HD double rm_acos_band(double x, int m, int n)
{
    typedef AtanK K;
    const double* a = rm_asncs + n;
    double xx = ((m > 0) ? x : -x) - a[0];
    double p = a[DEG + 1];
#pragma unroll
    for (int j = DEG; j >= 2; --j) p = rm_fma(xx, p, a[j]);
    p = rm_fma(xx * xx, p, a[DEG + 2]);
    double t = rm_fma(xx, a[1], p);
    double c = a[DEG + 3];
    if (m > 0) {
        double y = K::hpi - c;
        t = K::hpi1 - t;
        return t + y;
    }
    double y = c + K::hpi;
    t = t + K::hpi1;
    return t + y;
}

RM_MATH_HD double rm_acos(double x)
{
    typedef AtanK K;
    const uint64_t bits = rm_asuint64(x);
    const int m = (int)(bits >> 32);
    const int k = m & 0x7fffffff;
    if (k < 0x3c880000) return K::hpi;                           // |x| < 2^-55
    if (k < 0x3fc00000) {                                        // |x| < 1/8
        double x2 = x * x;
        double p = rm_fma(x2, K::f6, K::f5);
        p = rm_fma(x2, p, K::f4);
        p = rm_fma(x2, p, K::f3);
        p = rm_fma(x2, p, K::f2);
        p = rm_fma(x2, p, K::f1);
        double r = K::hpi - x;
        double cor = rm_fnma(p, x * x2, ((K::hpi - r) - x) + K::hpi1);
        return r + cor;
    }
    if (k < 0x3fe00000) {                                        // 1/8 <= |x| < 1/2
        int n = (k < 0x3fd00000) ? 11 * ((k >> 15) & 0x1f) : 11 * ((k >> 14) & 0x3f) + 352;
        return rm_acos_band<5>(x, m, n);
    }
    if (k < 0x3fe80000) return rm_acos_band<6>(x, m, 1056 + 12 * ((k >> 13) & 0x7f));   // < 0.75
    if (k < 0x3fed8000) return rm_acos_band<7>(x, m, 992 + 13 * ((k >> 13) & 0x7f));     // < 0.921875
    if (k < 0x3fee8000) return rm_acos_band<8>(x, m, 884 + 14 * ((k >> 13) & 0x7f));     // < 0.953125
    if (k < 0x3fef0000) return rm_acos_band<9>(x, m, 768 + 15 * ((k >> 13) & 0x7f));     // < 0.96875
    if (k < 0x3ff00000) {                                        // 0.96875 <= |x| < 1
        double z = ((m > 0) ? (1.0 - x) : (x + 1.0)) * 0.5;
        const uint64_t zb = rm_asuint64(z);
        const int kz = (int)(zb >> 32);
        double t = rm_inroot[(kz >> 14) & 0x7f] * rm_asdouble((uint64_t)(1023 + (511 - (kz >> 21))) << 52);  // powtwo[]
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
        if (m < 0) {
            double cor = (K::hpi1 - cc) - pz;
            double res1 = K::hpi - y;
            double res = cor + res1;
            return res + res;
        }
        double res = (cc + pz) + y;
        return res + res;
    }
    if (k == 0x3ff00000 && (uint32_t)bits == 0) return (m > 0) ? 0.0 : K::opi;   // |x| == 1
    if (x != x) return x + x;
    return (x - x) / (x - x);                                    // |x| > 1: invalid
}

// ---- e_atan2.c: __ieee754_atan2 ------------------------------------------------------------

RM_MATH_HD double rm_atan_series(double v)   // d3 + v*(d5 + v*(d7 + v*(d9 + v*(d11 + v*d13))))
{
    typedef AtanK K;
    double p = rm_fma(v, K::d13, K::d11);
    p = rm_fma(v, p, K::d9);
    p = rm_fma(v, p, K::d7);
    p = rm_fma(v, p, K::d5);
    return rm_fma(v, p, K::d3);
}

RM_MATH_HD const double* rm_atan_row(double u)
{
    int i = (int)(rm_fma(u, 256.0, 0x1p52) - 0x1p52) - 16;
    return rm_cij + 7 * i;
}

RM_MATH_HD double rm_atan_row_poly(const double* c, double v)   // c2 + v*(c3 + v*(c4 + v*(c5 + v*c6)))
{
    double p = rm_fma(v, c[6], c[5]);
    p = rm_fma(v, p, c[4]);
    p = rm_fma(v, p, c[3]);
    return rm_fma(v, p, c[2]);
}

RM_MATH_HD double rm_atan2(double y, double x)
{
    typedef AtanK K;
    const uint64_t bx = rm_asuint64(x), by = rm_asuint64(y);
    const uint32_t ux = (uint32_t)(bx >> 32), uy = (uint32_t)(by >> 32);
    if ((ux & 0x7ff00000u) == 0x7ff00000u || (uy & 0x7ff00000u) == 0x7ff00000u) return ::atan2(y, x);   // inf / nan
    if ((by << 1) == 0)                                              // y = +-0
        return (ux & 0x80000000u) ? ((uy & 0x80000000u) ? -K::opi : K::opi) : ((uy & 0x80000000u) ? -0.0 : 0.0);
    if (x == 0.0) return (uy & 0x80000000u) ? -K::hpi : K::hpi;      // x = +-0

    double ax = (x < 0.0) ? -x : x;
    double ay = (y < 0.0) ? -y : y;
    const int de = (int)(uy & 0x7ff00000u) - (int)(ux & 0x7ff00000u);
    if (de >= 59768832) return (y > 0.0) ? K::hpi : -K::hpi;          // |y/x| huge
    if (de <= -59768832) {                                           // |y/x| tiny
        if (x > 0.0) return __builtin_copysign(ay / ax, y);
        return (y > 0.0) ? K::opi : -K::opi;
    }
    if (ax < 0x1p-500 || ay < 0x1p-500) { ax *= 0x1p500; ay *= 0x1p500; }
    if (ax > 0x1p500 || ay > 0x1p500) { ax *= 0x1p-500; ay *= 0x1p-500; }

    double u, du;
    const bool y_lt_x = ay < ax;
    if (y_lt_x) {
        u = ay / ax;
        double v = ax * u;
        double vv = rm_fma(ax, u, -v);
        du = ((ay - v) - vv) / ax;
    } else {
        u = ax / ay;
        double v = ay * u;
        double vv = rm_fma(ay, u, -v);
        du = ((ax - v) - vv) / ay;
    }

    double z;
    if (x > 0.0) {
        if (y_lt_x) {                                                // (i) atan(ay/ax)
            if (u < 0.0625) {
                double v = u * u;
                double zz = rm_fma(u * v, rm_atan_series(v), du);
                z = u + zz;
            } else {
                const double* c = rm_atan_row(u);
                double t3 = u - c[0];
                double v = du + t3;
                double dv = (rm_fabs(t3) > rm_fabs(du)) ? ((t3 - v) + du) : ((du - v) + t3);
                double p = rm_fma(v, c[6], c[5]);
                p = rm_fma(v, p, c[4]);
                p = rm_fma(v, p, c[3]);
                p = (v * v) * p;
                p = rm_fma(dv, c[2], p);
                double zz = rm_fma(v, c[2], p);
                z = zz + c[1];
            }
        } else {                                                     // (ii) pi/2 - atan(ax/ay)
            if (u < 0.0625) {
                double v = u * u;
                double zz = (u * v) * rm_atan_series(v);
                double t2 = K::hpi - u;
                double cor = (K::hpi - t2) - u;
                double t3 = ((cor + K::hpi1) - du) - zz;
                z = t3 + t2;
            } else {
                const double* c = rm_atan_row(u);
                double v = (u - c[0]) + du;
                double zz = rm_fnma(v, rm_atan_row_poly(c, v), K::hpi1);
                double t1 = K::hpi - c[1];
                z = t1 + zz;
            }
        }
    } else {
        if (ax < ay) {                                               // (iii) pi/2 + atan(ax/ay)
            if (u < 0.0625) {
                double v = u * u;
                double zz = (v * u) * rm_atan_series(v);
                double t2 = u + K::hpi;
                double cor = (K::hpi - t2) + u;
                double t3 = ((cor + K::hpi1) + du) + zz;
                z = t3 + t2;
            } else {
                const double* c = rm_atan_row(u);
                double v = (u - c[0]) + du;
                double zz = rm_fma(v, rm_atan_row_poly(c, v), K::hpi1);
                double t1 = K::hpi + c[1];
                z = t1 + zz;
            }
        } else {                                                     // (iv) pi - atan(ay/ax)
            if (u < 0.0625) {
                double v = u * u;
                double zz = (v * u) * rm_atan_series(v);
                double t2 = K::opi - u;
                double cor = (K::opi - t2) - u;
                double t3 = ((cor + K::opi1) - du) - zz;
                z = t3 + t2;
            } else {
                const double* c = rm_atan_row(u);
                double v = (u - c[0]) + du;
                double zz = rm_fnma(v, rm_atan_row_poly(c, v), K::opi1);
                double t1 = K::opi - c[1];
                z = t1 + zz;
            }
        }
    }
    return __builtin_copysign(z, y);
}

}  // namespace rm
