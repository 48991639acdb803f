// rm_scenes.h -- the 20 catalogue scenes as inlinable functors (one per scene id).
//
// Scene ids follow get_all_scenes() (scenes/catalog.py:640-663).  Each sdf() keeps
// the reference's evaluation order (IEEE binary64, no contraction).  `x ** 0.5`,
// `x ** 2`, `r ** 7.0` ... are float_pow in the reference and therefore rm_pow here.
#pragma once

#include "rm_core.h"

namespace rm {

// Which gathered libm tables a scene's SDF reaches (the kernels mirror exactly those in LDS).
// Every scene needs the pow tables: camera normalisation and length() are `** 0.5`.
enum : unsigned { TB_POW = 1u, TB_SINCOS = 2u, TB_ACOS = 4u, TB_ATAN = 8u, TB_LOG = 16u };
template <class Scene> struct SceneTables { static constexpr unsigned value = TB_POW; };
// Scenes whose SDF is a data-dependent loop expose it as a resumable evaluation (Scene::Eval with
// begin / trip / value); the render kernel then interleaves trips of different evaluations.
template <class Scene> struct SceneIterative { static constexpr bool value = false; };

// ---- scenes/primitives.py -----------------------------------------------------

RM_HD double sd_sphere(vec3 p, double radius) { return length(p) - radius; }            // :11-12

RM_HD double sd_box(vec3 p, vec3 b)                                                      // :14-18
{
    vec3 q = v3(rm_fabs(p.x) - b.x, rm_fabs(p.y) - b.y, rm_fabs(p.z) - b.z);
    double outside = length(v3(py_max(q.x, 0.0), py_max(q.y, 0.0), py_max(q.z, 0.0)));
    double inside = py_min(py_max(q.x, py_max(q.y, q.z)), 0.0);
    return outside + inside;
}

RM_HD double sd_plane(vec3 p, vec3 n, double offset) { return dot(p, n) - offset; }      // :20-21

RM_HD double sd_cylinder(vec3 p, double radius, double half_height)                      // :23-28
{
    double d_radial = pow_half(p.x * p.x + p.z * p.z) - radius;
    double d_height = rm_fabs(p.y) - half_height;
    double outside = pow_half(rm_pow(py_max(d_radial, 0.0), 2.0) + rm_pow(py_max(d_height, 0.0), 2.0));
    double inside = py_min(py_max(d_radial, d_height), 0.0);
    return outside + inside;
}

RM_HD double sd_torus(vec3 p, double major_radius, double minor_radius)                  // :30-32
{
    double q_xz = pow_half_a(p.x * p.x + p.z * p.z) - major_radius;
    return pow_half_a(q_xz * q_xz + p.y * p.y) - minor_radius;
}

RM_HD double sd_capped_torus(vec3 p, double sc0, double sc1, double ra, double rb)       // :41-50
{
    double px = rm_fabs(p.x);
    double k;
    if (sc1 * px > sc0 * p.y)
        k = px * sc0 + p.y * sc1;
    else
        k = pow_half_a(px * px + p.y * p.y);
    return pow_half_a(p.x * p.x + p.y * p.y + p.z * p.z + ra * ra - 2.0 * ra * k) - rb;
}

RM_HD double op_smooth_union(double d1, double d2, double k)                             // :80-86
{
    double h = py_max(0.0, py_min(1.0, 0.5 + 0.5 * (d2 - d1) / k));
    return (d2 * (1.0 - h) + d1 * h) - k * h * (1.0 - h);
}

// op_repeat (:102-108), one axis, spacing > 0 and a power of two
RM_HD double repeat_axis(double x, double spacing)
{
    return py_mod_pow2(x + spacing * 0.5, spacing) - spacing * 0.5;
}

// ---- scenes/catalog.py ----------------------------------------------------------

struct SceneSphere {                                                                     // :25-26
    static RM_HD double sdf(vec3 p) { return sd_sphere(p, 1.0); }
};
struct SceneGrazingPlane {                                                               // :44-45
    static RM_HD double sdf(vec3 p) { return sd_plane(p, v3(0.0, 1.0, 0.0), -0.5); }
};
struct SceneCube {                                                                       // :68-69
    static RM_HD double sdf(vec3 p) { return sd_box(p, v3(1.0, 1.0, 1.0)); }
};
struct SceneThinTorus {                                                                  // :87-88
    static RM_HD double sdf(vec3 p) { return sd_torus(p, 1.5, 0.05); }
};
struct SceneCylinder {                                                                   // :105-106
    static RM_HD double sdf(vec3 p) { return sd_cylinder(p, 1.0, 1.5); }
};
struct SceneNearMiss {                                                                   // :124-127
    static RM_HD double sdf(vec3 p)
    {
        double d1 = sd_sphere(p - v3(-1.01, 0.0, 0.0), 1.0);
        double d2 = sd_sphere(p - v3(1.01, 0.0, 0.0), 1.0);
        return py_min(d1, d2);
    }
};
struct SceneHollowCube {                                                                 // :145-148
    static RM_HD double sdf(vec3 p)
    {
        double d_box = sd_box(p, v3(1.0, 1.0, 1.0));
        double d_sphere = sd_sphere(p, 1.3);
        return py_max(d_box, -d_sphere);
    }
};
struct SceneSmoothBlend {                                                                // :166-169
    static RM_HD double sdf(vec3 p)
    {
        double d1 = sd_sphere(p - v3(-0.5, 0.0, 0.0), 0.8);
        double d2 = sd_box(p - v3(0.5, 0.0, 0.0), v3(0.6, 0.6, 0.6));
        return op_smooth_union(d1, d2, 0.5);
    }
};
struct SceneOnionShell {                                                                 // :187-191
    static RM_HD double sdf(vec3 p)
    {
        double d = sd_sphere(p, 2.0);
        d = rm_fabs(d) - 0.1;
        d = rm_fabs(d) - 0.05;
        return d;
    }
};
struct SceneMenger {                                                                     // :212-241 (iterations=3)
    static RM_HD double sdf(vec3 p)
    {
        double d = sd_box(p, v3(1.0, 1.0, 1.0));
        double s = 1.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            vec3 a = v3(py_mod_pow2(p.x * s, 2.0) - 1.0, py_mod_pow2(p.y * s, 2.0) - 1.0,
                        py_mod_pow2(p.z * s, 2.0) - 1.0);
            s *= 3.0;
            vec3 r = v3(rm_fabs(1.0 - 3.0 * rm_fabs(a.x)), rm_fabs(1.0 - 3.0 * rm_fabs(a.y)),
                        rm_fabs(1.0 - 3.0 * rm_fabs(a.z)));
            double da = py_max(r.x, r.y);
            double db = py_max(r.y, r.z);
            double dc = py_max(r.z, r.x);
            double c = (py_min(da, py_min(db, dc)) - 1.0) / s;
            d = py_max(d, c);
        }
        return d;
    }
};
struct SceneMandelbulb {                                                                 // :266-293 (power 8, 8 iterations)
    // The SDF is a data-dependent loop (1..8 trips, 1.8 on average on the default view, the lanes of a
    // wavefront rarely agree).  It is written as a resumable evaluation -- begin / trip / value -- so the
    // render kernel can run ONE trip per turn for every lane and let a lane whose value is ready go on
    // to its next evaluation (rm_kernels.h); sdf() below is the same three pieces run to completion.
    struct Eval {
        vec3 p, z;
        double dr, r;
        int32_t i;
    };
    // start of an evaluation: the first `r = z.length(); if r > bailout: break` (:272-274).
    // true when the value is ready at once (the point lies outside the bailout radius)
    static RM_HD bool begin(Eval& e, vec3 p)
    {
        e.p = p; e.z = p; e.dr = 1.0; e.i = 0;
        e.r = length_a(p);
        return e.r > 4.0;
    }
    // iterations the evaluation ran (8 = no bail-out: the point is close to the set; rm_pipeline.h EARLY HAND-OVER)
    static RM_HD int eval_trips(const Eval& e) { return e.i; }
    // the body of one trip of `for i in range(8)` (:276-290) followed by the loop test and the next
    // trip's length / bailout test; needs r <= 4 and i < 8.  true when the loop has ended.  The loop
    // is rotated (test at the bottom) so that every trip a lane takes is a full update: an
    // evaluation with u updates costs u trips, none for a point outside the bailout radius.
    static RM_HD bool trip(Eval& e)
    {
        const double power = 8.0;
        const vec3 z = e.z;
        const double r = e.r;
        // acos / atan2 / the two pows are independent of one another, as are the two sincos
        // pairs: written back to back (all branch-free) so their dependency chains interleave.
        double theta = rm_acos(py_max(-1.0, py_min(1.0, z.z / py_max(r, 1e-12))));
        double phi = rm_atan2(z.y, z.x);
        double r7, zr;
        rm_pow2(r, power - 1.0, power, &r7, &zr);       // r ** 7.0 and r ** 8.0 share log(r)
        e.dr = r7 * power * e.dr + 1.0;
        theta *= power;
        phi *= power;
        double st, ct, sp, cp;
        rm_sincos(theta, &st, &ct);
        rm_sincos(phi, &sp, &cp);
        e.z = v3(zr * st * cp, zr * st * sp, zr * ct) + e.p;
        if (++e.i >= 8) return true;                    // r keeps the length measured before this update
        e.r = length_a(e.z);
        return e.r > 4.0;
    }
    // ---- the trip split by function, for a TEAM of wavefronts (rm_kernels.h) ----------------------
    // The three transcendental chains of a trip are independent of one another:
    //   part 0:  sin, cos of 8 * acos(clamp(z.z / max(r, 1e-12)))            (:277, :286-290)
    //   part 1:  sin, cos of 8 * atan2(z.y, z.x)                              (:278, :287-289)
    //   part 2:  r ** 7.0 and r ** 8.0                                        (:280, :283)
    // trip_part evaluates one of them, trip_join does the rest of the trip from all six values.
    // trip(e) == trip_join(e, parts 0..2): same expressions, same order, same bits.
    // (wave-uniform band skipping inside acos / atan2 / sincos: a team wave is alone on its SIMD with few live
    // lanes, every instruction it does not issue shortens the frame's critical chain -- rm_math_trig.h)
    static RM_HD void trip_part(const Eval& e, int part, double& o0, double& o1)
    {
        const double power = 8.0;
        if (part == 0) {
            double theta = rm_acos<true>(py_max(-1.0, py_min(1.0, e.z.z / py_max(e.r, 1e-12))));
            theta *= power;
            rm_sincos<true>(theta, &o0, &o1);
        } else if (part == 1) {
            double phi = rm_atan2<true>(e.z.y, e.z.x);
            phi *= power;
            rm_sincos<true>(phi, &o0, &o1);
        } else {
            rm_pow2(e.r, power - 1.0, power, &o0, &o1);
        }
    }
    static RM_HD bool trip_join(Eval& e, double st, double ct, double sp, double cp, double r7, double zr)
    {
        const double power = 8.0;
        e.dr = r7 * power * e.dr + 1.0;
        e.z = v3(zr * st * cp, zr * st * sp, zr * ct) + e.p;
        if (++e.i >= 8) return true;
        e.r = length_a(e.z);
        return e.r > 4.0;
    }
    static RM_HD double value(const Eval& e)
    {
        return 0.5 * rm_log(py_max(e.r, 1e-12)) * e.r / py_max(e.dr, 1e-12);
    }
    static RM_HD double sdf(vec3 p)
    {
        Eval e;
        bool done = begin(e, p);
        while (!done) done = trip(e);
        return value(e);
    }
};
struct SceneBadLipschitz {                                                               // :320-321
    static RM_HD double sdf(vec3 p) { return (length(p) - 1.0) * 2.0; }
};
struct ScenePillars {                                                                    // :339-344
    static RM_HD double sdf(vec3 p)
    {
        vec3 q = v3(repeat_axis(p.x, 2.0), p.y, repeat_axis(p.z, 2.0));
        double d_pillar = sd_cylinder(q, 0.15, 3.0);
        double d_floor = sd_plane(p, v3(0.0, 1.0, 0.0), -3.0);
        return py_min(d_pillar, d_floor);
    }
};
struct SceneThinPlanes {                                                                 // :368-377
    static RM_HD double sdf(vec3 p)
    {
        const double spacing = 0.5;
        double py_m = py_mod_pow2(p.y + spacing * 0.5, spacing) - spacing * 0.5;
        vec3 q = v3(p.x, py_m, p.z);
        return rm_fabs(sd_plane(q, v3(0.0, 1.0, 0.0), 0.0)) - 0.01;
    }
};

// (center, radius) tables of Sphere Cloud (:401-414) and Bumpy Sphere (:450-461).
// Written as unrolled literal lists so they live in the instruction stream / SGPRs.
#define RM_CLOUD_LIST_A(X)                                                                 \
    X(0.4253, 1.3505, 0.9373, 0.4723) X(-0.9343, -0.6794, 1.2701, 0.4257)                 \
    X(-1.6821, 1.0922, 1.0100, 0.3090) X(-0.1090, -0.6697, -0.7534, 0.4659)               \
    X(-0.8334, -0.1867, 0.0155, 0.4879) X(0.1819, 1.6847, 0.9951, 0.4789)                 \
    X(0.4154, 1.6625, -0.9680, 0.4053) X(-1.1553, 0.3826, -1.5506, 0.3120)
#define RM_CLOUD_LIST_B(X)                                                                 \
    X(-1.5787, 0.0506, -0.1149, 0.3223) X(1.4184, 0.4394, 0.0480, 0.4841)                 \
    X(-0.0106, -0.8584, -1.6599, 0.4015) X(-1.0458, 0.6529, -1.0179, 0.3197)              \
    X(-0.4436, -1.6873, 1.1222, 0.4745) X(-1.1748, -0.7902, 1.2931, 0.4211)               \
    X(0.0333, 1.1803, 0.4750, 0.4053) X(0.8220, -1.3889, 0.1399, 0.3628)
#define RM_CLOUD_LIST_C(X)                                                                 \
    X(0.0264, 1.2626, -0.4717, 0.3704) X(0.3338, -1.4985, -0.3821, 0.3327)                \
    X(-0.6017, -1.1893, 1.0755, 0.2884) X(-0.4099, 1.6277, 0.3060, 0.4728)                \
    X(0.3572, 0.4692, 0.5999, 0.3829) X(-1.1873, -0.2029, -0.8855, 0.4005)                \
    X(-0.3315, -1.3712, 1.5906, 0.3509) X(-0.9690, 0.5840, -0.6786, 0.4453)
#define RM_CLOUD_LIST(X) RM_CLOUD_LIST_A(X) RM_CLOUD_LIST_B(X) RM_CLOUD_LIST_C(X)

// A union of many primitives is a flat evaluation, but its terms are independent: for a TEAM of wavefronts
// (rm_kernels.h) it is written like a one-trip resumable SDF whose trip splits three ways -- every wave takes
// the minimum over a third of the list, trip_join takes the minimum of the three.  min is exact and
// associative on these values (no NaN, and length(...) - r never yields -0.0), so the result has the same bits.
struct UnionEval {
    vec3 p;
    double d;
};

struct SceneSphereCloud {                                                                // :424-428
    static RM_HD double sdf(vec3 p)
    {
        double d = 1e10;
#define RM_X(cx, cy, cz, r) d = py_min(d, sd_sphere(p - v3(cx, cy, cz), r));
        RM_CLOUD_LIST(RM_X)
#undef RM_X
        return d;
    }
    using Eval = UnionEval;
    static RM_HD bool begin(Eval& e, vec3 p) { e.p = p; return false; }
    static RM_HD bool trip(Eval& e) { e.d = sdf(e.p); return true; }
    static RM_HD double value(const Eval& e) { return e.d; }
    static RM_HD void trip_part(const Eval& e, int part, double& o0, double& o1)
    {
        const vec3 p = e.p;
        double d = 1e10;
#define RM_X(cx, cy, cz, r) d = py_min(d, sd_sphere(p - v3(cx, cy, cz), r));
        if (part == 0) { RM_CLOUD_LIST_A(RM_X) } else if (part == 1) { RM_CLOUD_LIST_B(RM_X) } else { RM_CLOUD_LIST_C(RM_X) }
#undef RM_X
        o0 = d; o1 = 0.0;
    }
    static RM_HD bool trip_join(Eval& e, double a, double, double b, double, double c, double)
    {
        e.d = py_min(py_min(a, b), c);
        return true;
    }
};

#define RM_BUMP_LIST_A(X)                                                                  \
    X(0.3841, 1.4500, 0.0000) X(-0.4821, 1.3500, 0.4417) X(0.0725, 1.2500, -0.8260)       \
    X(0.5860, 1.1500, 0.7643) X(-1.0548, 1.0500, -0.1866) X(0.9794, 0.9500, -0.6230)      \
    X(-0.3209, 0.8500, 1.1935) X(-0.5987, 0.7500, -1.1528) X(1.2698, 0.6500, 0.4637)      \
    X(-1.2900, 0.5500, 0.5325)
#define RM_BUMP_LIST_B(X)                                                                  \
    X(0.6065, 0.4500, -1.2960) X(0.4365, 0.3500, 1.3917)                                  \
    X(-1.2797, 0.2500, -0.7416) X(1.4577, 0.1500, -0.3205) X(-0.8622, 0.0500, 1.2264)     \
    X(-0.1927, -0.0500, -1.4867) X(1.1412, -0.1500, 0.9618) X(-1.4778, -0.2500, 0.0611)   \
    X(1.0339, -0.3500, -1.0289) X(-0.0661, -0.4500, 1.4294)
#define RM_BUMP_LIST_C(X)                                                                  \
    X(-0.8941, -0.5500, -1.0715)                                                          \
    X(1.3398, -0.6500, 0.1803) X(-1.0663, -0.7500, 0.7419) X(0.2713, -0.8500, -1.2058)    \
    X(0.5771, -0.9500, 1.0072) X(-1.0205, -1.0500, -0.3256) X(0.8743, -1.1500, -0.4039)   \
    X(-0.3201, -1.2500, 0.7649) X(-0.2213, -1.3500, -0.6152) X(0.3400, -1.4500, 0.1787)
#define RM_BUMP_LIST(X) RM_BUMP_LIST_A(X) RM_BUMP_LIST_B(X) RM_BUMP_LIST_C(X)

struct SceneBumpySphere {                                                                // :471-475
    static RM_HD double sdf(vec3 p)
    {
        double d = sd_sphere(p, 1.4);
#define RM_X(cx, cy, cz) d = py_min(d, sd_sphere(p - v3(cx, cy, cz), 0.18));
        RM_BUMP_LIST(RM_X)
#undef RM_X
        return d;
    }
    using Eval = UnionEval;
    static RM_HD bool begin(Eval& e, vec3 p) { e.p = p; return false; }
    static RM_HD bool trip(Eval& e) { e.d = sdf(e.p); return true; }
    static RM_HD double value(const Eval& e) { return e.d; }
    static RM_HD void trip_part(const Eval& e, int part, double& o0, double& o1)
    {
        const vec3 p = e.p;
        double d = 1e10;
#define RM_X(cx, cy, cz) d = py_min(d, sd_sphere(p - v3(cx, cy, cz), 0.18));
        if (part == 0) { d = sd_sphere(p, 1.4); RM_BUMP_LIST_A(RM_X) } else if (part == 1) { RM_BUMP_LIST_B(RM_X) } else { RM_BUMP_LIST_C(RM_X) }
#undef RM_X
        o0 = d; o1 = 0.0;
    }
    static RM_HD bool trip_join(Eval& e, double a, double, double b, double, double c, double)
    {
        e.d = py_min(py_min(a, b), c);
        return true;
    }
};

struct SceneGyroid {                                                                     // :496-517
    static constexpr double FREQ = 3.0;
    static constexpr double LIP = 0x1.4c8dc2e423980p+3;  // 3.0 * 2.0 * (3.0 ** 0.5) as CPython evaluates it
    static RM_HD double combine(vec3 p, double sx, double cx, double sy, double cy, double sz, double cz)
    {
        double g = sx * cy + sy * cz + sz * cx;
        double sheet = g / LIP;
        double ball = sd_sphere(p, 2.2);
        return py_max(sheet, ball);
    }
    static RM_HD double sdf(vec3 p)
    {
        double qx = FREQ * p.x, qy = FREQ * p.y, qz = FREQ * p.z;
        double sx, cx, sy, cy, sz, cz;
        rm_sincos(qx, &sx, &cx);
        rm_sincos(qy, &sy, &cy);
        rm_sincos(qz, &sz, &cz);
        return combine(p, sx, cx, sy, cy, sz, cz);
    }
};

struct SceneCappedTorus {                                                                // :533-548
    static RM_HD double sdf(vec3 p)
    {
        // SC = (math.sin(2.0), math.cos(2.0)) as glibc evaluates them
        return sd_capped_torus(p, 0x1.d18f6ead1b446p-1, -0x1.aa22657537205p-2, 1.2, 0.2);
    }
};

struct SceneBoxLattice {                                                                 // :581-590
    static RM_HD double cell(double x)
    {
        double r = rm_floor(x / 1.0 + 0.5);
        r = py_max(-2.0, py_min(2.0, r));
        return r;
    }
    static RM_HD double sdf(vec3 p)
    {
        vec3 q = v3(p.x - 1.0 * cell(p.x), p.y - 1.0 * cell(p.y), p.z - 1.0 * cell(p.z));
        return sd_box(q, v3(0.3, 0.3, 0.3));
    }
};

struct SceneMetaballs {                                                                  // :608-633
    static RM_HD double sdf(vec3 p)
    {
        const double K = 0.45;
        double d = sd_sphere(p - v3(0.0, 0.0, 0.0), 0.8);
        d = op_smooth_union(d, sd_sphere(p - v3(1.0, 0.0, 0.0), 0.6), K);
        d = op_smooth_union(d, sd_sphere(p - v3(-1.0, 0.0, 0.0), 0.6), K);
        d = op_smooth_union(d, sd_sphere(p - v3(0.0, 1.0, 0.0), 0.6), K);
        d = op_smooth_union(d, sd_sphere(p - v3(0.0, -1.0, 0.0), 0.6), K);
        d = op_smooth_union(d, sd_sphere(p - v3(0.0, 0.0, 1.0), 0.6), K);
        return d;
    }
};

template <> struct SceneTables<SceneMandelbulb> { static constexpr unsigned value = TB_POW | TB_SINCOS | TB_ACOS | TB_ATAN | TB_LOG; };
template <> struct SceneIterative<SceneMandelbulb> { static constexpr bool value = true; };
template <> struct SceneIterative<SceneSphereCloud> { static constexpr bool value = true; };    // one trip, splits three ways
template <> struct SceneIterative<SceneBumpySphere> { static constexpr bool value = true; };
template <> struct SceneTables<SceneGyroid> { static constexpr unsigned value = TB_POW | TB_SINCOS; };

#define RM_NUM_SCENES 20

// X(id, functor) in registry order
#define RM_SCENE_LIST(X)                                                                   \
    X(0, SceneSphere) X(1, SceneGrazingPlane) X(2, SceneCube) X(3, SceneThinTorus)         \
    X(4, SceneCylinder) X(5, SceneNearMiss) X(6, SceneHollowCube) X(7, SceneSmoothBlend)   \
    X(8, SceneOnionShell) X(9, SceneMenger) X(10, SceneMandelbulb) X(11, SceneBadLipschitz)\
    X(12, ScenePillars) X(13, SceneThinPlanes) X(14, SceneSphereCloud) X(15, SceneBumpySphere) \
    X(16, SceneGyroid) X(17, SceneCappedTorus) X(18, SceneBoxLattice) X(19, SceneMetaballs)

}  // namespace rm
