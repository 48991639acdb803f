/*
 * rm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's per-ray SDF sphere-tracing path
 * (kylegrover/raymarch-algo-compare, CPython + glibc libm, IEEE binary64).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (raymarch_algo_compare_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file
 * bit-for-bit (iterations, hit, t, final_sdf as raw doubles) against fixtures
 * under tests/golden/ that oracle/gen_golden.py produced by importing the
 * reference itself in the build container.
 *
 * Every arithmetic expression follows the reference's evaluation order, and
 * every `x ** y`, `%`, math.* call goes to the same libm entry point CPython
 * uses (float_pow -> pow, float_rem -> fmod, math.sin -> sin ...).  Build with
 * -fno-builtin -ffp-contract=off (see oracle/Makefile): gcc would otherwise
 * fold pow(x, 2.0) into x*x, which is NOT what CPython computes.
 *
 * File:line citations are relative to /root/reference/raymarching_benchmark/.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct { double x, y, z; } v3;

/* ---- Python scalar semantics --------------------------------------------- */

/* float.__pow__ -> libm pow (CPython Objects/floatobject.c float_pow). */
static double py_pow(double a, double b) { return pow(a, b); }

/* float.__mod__ (float_rem): fmod, then sign-of-divisor fix-up. */
static double py_mod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* builtin max/min keep the FIRST argument on ties / unordered compares. */
static double py_max(double a, double b) { return (b > a) ? b : a; }
static double py_min(double a, double b) { return (b < a) ? b : a; }

/* ---- core/vec3.py --------------------------------------------------------- */

static v3 v(double x, double y, double z) { v3 r = { x, y, z }; return r; }
static v3 v_add(v3 a, v3 b) { return v(a.x + b.x, a.y + b.y, a.z + b.z); }     /* vec3.py:17-18 */
static v3 v_sub(v3 a, v3 b) { return v(a.x - b.x, a.y - b.y, a.z - b.z); }     /* vec3.py:20-21 */
static v3 v_mul(v3 a, double s) { return v(a.x * s, a.y * s, a.z * s); }       /* vec3.py:23-24 */
static double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* vec3.py:36-37 */
static double v_len(v3 a)                                                       /* vec3.py:46-47 */
{
    return py_pow(a.x * a.x + a.y * a.y + a.z * a.z, 0.5);
}
static v3 v_normalized(v3 a)                                                    /* vec3.py:52-56, 32-34 */
{
    double l = v_len(a);
    if (l < 1e-12) return v(0.0, 0.0, 0.0);
    double inv = 1.0 / l;
    return v(a.x * inv, a.y * inv, a.z * inv);
}

/* core/ray.py:6-17 */
typedef struct { v3 o, d; } ray_t;
static v3 ray_at(const ray_t *r, double t) { return v_add(r->o, v_mul(r->d, t)); }

/* ---- scenes/primitives.py ------------------------------------------------- */

static double sd_sphere(v3 p, double radius) { return v_len(p) - radius; }       /* :11-12 */

static double sd_box(v3 p, v3 b)                                                 /* :14-18 */
{
    v3 q = v(fabs(p.x) - b.x, fabs(p.y) - b.y, fabs(p.z) - b.z);
    double outside = v_len(v(py_max(q.x, 0.0), py_max(q.y, 0.0), py_max(q.z, 0.0)));
    double inside = py_min(py_max(q.x, py_max(q.y, q.z)), 0.0);
    return outside + inside;
}

static double sd_plane(v3 p, v3 n, double offset) { return v_dot(p, n) - offset; } /* :20-21 */

static double sd_cylinder(v3 p, double radius, double half_height)              /* :23-28 */
{
    double d_radial = py_pow(p.x * p.x + p.z * p.z, 0.5) - radius;
    double d_height = fabs(p.y) - half_height;
    double outside = py_pow(py_pow(py_max(d_radial, 0.0), 2.0) + py_pow(py_max(d_height, 0.0), 2.0), 0.5);
    double inside = py_min(py_max(d_radial, d_height), 0.0);
    return outside + inside;
}

static double sd_torus(v3 p, double major, double minor)                        /* :30-32 */
{
    double q_xz = py_pow(p.x * p.x + p.z * p.z, 0.5) - major;
    return py_pow(q_xz * q_xz + p.y * p.y, 0.5) - minor;
}

static double sd_capped_torus(v3 p, double sc0, double sc1, double ra, double rb) /* :41-50 */
{
    double px = fabs(p.x);
    double k;
    if (sc1 * px > sc0 * p.y) k = px * sc0 + p.y * sc1;
    else k = py_pow(px * px + p.y * p.y, 0.5);
    return py_pow(p.x * p.x + p.y * p.y + p.z * p.z + ra * ra - 2.0 * ra * k, 0.5) - rb;
}

static double op_smooth_union(double d1, double d2, double k)                   /* :80-86 */
{
    double h = py_max(0.0, py_min(1.0, 0.5 + 0.5 * (d2 - d1) / k));
    return (d2 * (1.0 - h) + d1 * h) - k * h * (1.0 - h);
}

/* op_repeat (:102-108) for one axis with spacing > 0 */
static double repeat_axis(double x, double spacing)
{
    return py_mod(x + spacing * 0.5, spacing) - spacing * 0.5;
}

/* ---- scenes/catalog.py ---------------------------------------------------- */

static double sc_sphere(v3 p) { return sd_sphere(p, 1.0); }                               /* :25-26 */
static double sc_grazing_plane(v3 p) { return sd_plane(p, v(0.0, 1.0, 0.0), -0.5); }     /* :44-45 */
static double sc_cube(v3 p) { return sd_box(p, v(1.0, 1.0, 1.0)); }                       /* :68-69 */
static double sc_thin_torus(v3 p) { return sd_torus(p, 1.5, 0.05); }                      /* :87-88 */
static double sc_cylinder(v3 p) { return sd_cylinder(p, 1.0, 1.5); }                      /* :105-106 */
static double sc_near_miss(v3 p)                                                          /* :124-127 */
{
    double d1 = sd_sphere(v_sub(p, v(-1.01, 0.0, 0.0)), 1.0);
    double d2 = sd_sphere(v_sub(p, v(1.01, 0.0, 0.0)), 1.0);
    return py_min(d1, d2);
}
static double sc_hollow_cube(v3 p)                                                        /* :145-148 */
{
    double d_box = sd_box(p, v(1.0, 1.0, 1.0));
    double d_sphere = sd_sphere(p, 1.3);
    return py_max(d_box, -d_sphere);
}
static double sc_smooth_blend(v3 p)                                                       /* :166-169 */
{
    double d1 = sd_sphere(v_sub(p, v(-0.5, 0.0, 0.0)), 0.8);
    double d2 = sd_box(v_sub(p, v(0.5, 0.0, 0.0)), v(0.6, 0.6, 0.6));
    return op_smooth_union(d1, d2, 0.5);
}
static double sc_onion(v3 p)                                                              /* :187-191 */
{
    double d = sd_sphere(p, 2.0);
    d = fabs(d) - 0.1;
    d = fabs(d) - 0.05;
    return d;
}
static double sc_menger(v3 p)                                                             /* :212-241 */
{
    double d = sd_box(p, v(1.0, 1.0, 1.0));
    double s = 1.0;
    for (int i = 0; i < 3; ++i) {
        v3 a = v(py_mod(p.x * s, 2.0) - 1.0, py_mod(p.y * s, 2.0) - 1.0, py_mod(p.z * s, 2.0) - 1.0);
        s *= 3.0;
        v3 r = v(fabs(1.0 - 3.0 * fabs(a.x)), fabs(1.0 - 3.0 * fabs(a.y)), fabs(1.0 - 3.0 * fabs(a.z)));
        double da = py_max(r.x, r.y);
        double db = py_max(r.y, r.z);
        double dc = py_max(r.z, r.x);
        double c = (py_min(da, py_min(db, dc)) - 1.0) / s;
        d = py_max(d, c);
    }
    return d;
}
static double sc_mandelbulb(v3 p)                                                         /* :266-293 */
{
    const double power = 8.0;
    v3 z = p;
    double dr = 1.0, r = 0.0;
    for (int i = 0; i < 8; ++i) {
        r = v_len(z);
        if (r > 4.0) break;
        double theta = acos(py_max(-1.0, py_min(1.0, z.z / py_max(r, 1e-12))));
        double phi = atan2(z.y, z.x);
        dr = py_pow(r, power - 1.0) * power * dr + 1.0;
        double zr = py_pow(r, power);
        theta *= power;
        phi *= power;
        z = v_add(v(zr * sin(theta) * cos(phi), zr * sin(theta) * sin(phi), zr * cos(theta)), p);
    }
    return 0.5 * log(py_max(r, 1e-12)) * r / py_max(dr, 1e-12);
}
static double sc_bad_lipschitz(v3 p) { return (v_len(p) - 1.0) * 2.0; }                   /* :320-321 */
static double sc_pillars(v3 p)                                                            /* :339-344 */
{
    v3 q = v(repeat_axis(p.x, 2.0), p.y, repeat_axis(p.z, 2.0));
    double d_pillar = sd_cylinder(q, 0.15, 3.0);
    double d_floor = sd_plane(p, v(0.0, 1.0, 0.0), -3.0);
    return py_min(d_pillar, d_floor);
}
static double sc_thin_planes(v3 p)                                                        /* :368-377 */
{
    double spacing = 0.5;
    double py_m = py_mod(p.y + spacing * 0.5, spacing) - spacing * 0.5;
    v3 q = v(p.x, py_m, p.z);
    return fabs(sd_plane(q, v(0.0, 1.0, 0.0), 0.0)) - 0.01;
}

static const double CLOUD[24][4] = {                                                      /* :401-414 */
    { 0.4253, 1.3505, 0.9373, 0.4723 }, { -0.9343, -0.6794, 1.2701, 0.4257 },
    { -1.6821, 1.0922, 1.0100, 0.3090 }, { -0.1090, -0.6697, -0.7534, 0.4659 },
    { -0.8334, -0.1867, 0.0155, 0.4879 }, { 0.1819, 1.6847, 0.9951, 0.4789 },
    { 0.4154, 1.6625, -0.9680, 0.4053 }, { -1.1553, 0.3826, -1.5506, 0.3120 },
    { -1.5787, 0.0506, -0.1149, 0.3223 }, { 1.4184, 0.4394, 0.0480, 0.4841 },
    { -0.0106, -0.8584, -1.6599, 0.4015 }, { -1.0458, 0.6529, -1.0179, 0.3197 },
    { -0.4436, -1.6873, 1.1222, 0.4745 }, { -1.1748, -0.7902, 1.2931, 0.4211 },
    { 0.0333, 1.1803, 0.4750, 0.4053 }, { 0.8220, -1.3889, 0.1399, 0.3628 },
    { 0.0264, 1.2626, -0.4717, 0.3704 }, { 0.3338, -1.4985, -0.3821, 0.3327 },
    { -0.6017, -1.1893, 1.0755, 0.2884 }, { -0.4099, 1.6277, 0.3060, 0.4728 },
    { 0.3572, 0.4692, 0.5999, 0.3829 }, { -1.1873, -0.2029, -0.8855, 0.4005 },
    { -0.3315, -1.3712, 1.5906, 0.3509 }, { -0.9690, 0.5840, -0.6786, 0.4453 },
};
static double sc_sphere_cloud(v3 p)                                                       /* :424-428 */
{
    double d = 1e10;
    for (int i = 0; i < 24; ++i)
        d = py_min(d, sd_sphere(v_sub(p, v(CLOUD[i][0], CLOUD[i][1], CLOUD[i][2])), CLOUD[i][3]));
    return d;
}

static const double BUMPS[30][3] = {                                                      /* :450-461 */
    { 0.3841, 1.4500, 0.0000 }, { -0.4821, 1.3500, 0.4417 }, { 0.0725, 1.2500, -0.8260 },
    { 0.5860, 1.1500, 0.7643 }, { -1.0548, 1.0500, -0.1866 }, { 0.9794, 0.9500, -0.6230 },
    { -0.3209, 0.8500, 1.1935 }, { -0.5987, 0.7500, -1.1528 }, { 1.2698, 0.6500, 0.4637 },
    { -1.2900, 0.5500, 0.5325 }, { 0.6065, 0.4500, -1.2960 }, { 0.4365, 0.3500, 1.3917 },
    { -1.2797, 0.2500, -0.7416 }, { 1.4577, 0.1500, -0.3205 }, { -0.8622, 0.0500, 1.2264 },
    { -0.1927, -0.0500, -1.4867 }, { 1.1412, -0.1500, 0.9618 }, { -1.4778, -0.2500, 0.0611 },
    { 1.0339, -0.3500, -1.0289 }, { -0.0661, -0.4500, 1.4294 }, { -0.8941, -0.5500, -1.0715 },
    { 1.3398, -0.6500, 0.1803 }, { -1.0663, -0.7500, 0.7419 }, { 0.2713, -0.8500, -1.2058 },
    { 0.5771, -0.9500, 1.0072 }, { -1.0205, -1.0500, -0.3256 }, { 0.8743, -1.1500, -0.4039 },
    { -0.3201, -1.2500, 0.7649 }, { -0.2213, -1.3500, -0.6152 }, { 0.3400, -1.4500, 0.1787 },
};
static double sc_bumpy_sphere(v3 p)                                                       /* :471-475 */
{
    double d = sd_sphere(p, 1.4);
    for (int i = 0; i < 30; ++i)
        d = py_min(d, sd_sphere(v_sub(p, v(BUMPS[i][0], BUMPS[i][1], BUMPS[i][2])), 0.18));
    return d;
}

static double sc_gyroid(v3 p)                                                             /* :496-517 */
{
    const double FREQ = 3.0;
    const double LIP = 0x1.4c8dc2e423980p+3; /* 3.0 * 2.0 * (3.0 ** 0.5), evaluated by CPython */
    double qx = FREQ * p.x, qy = FREQ * p.y, qz = FREQ * p.z;
    double g = sin(qx) * cos(qy) + sin(qy) * cos(qz) + sin(qz) * cos(qx);
    double sheet = g / LIP;
    double ball = sd_sphere(p, 2.2);
    return py_max(sheet, ball);
}

static double sc_capped_torus(v3 p)                                                       /* :533-548 */
{
    /* SC = (math.sin(2.0), math.cos(2.0)) evaluated by CPython/glibc */
    return sd_capped_torus(p, 0x1.d18f6ead1b446p-1, -0x1.aa22657537205p-2, 1.2, 0.2);
}

static double lattice_cell(double x)                                                      /* :582-585 */
{
    double r = floor(x / 1.0 + 0.5);
    r = py_max(-2.0, py_min(2.0, r));
    return r;
}
static double sc_box_lattice(v3 p)                                                        /* :581-590 */
{
    v3 q = v(p.x - 1.0 * lattice_cell(p.x), p.y - 1.0 * lattice_cell(p.y), p.z - 1.0 * lattice_cell(p.z));
    return sd_box(q, v(0.3, 0.3, 0.3));
}

static const double BALLS[6][4] = {                                                       /* :608-615 */
    { 0.0, 0.0, 0.0, 0.8 }, { 1.0, 0.0, 0.0, 0.6 }, { -1.0, 0.0, 0.0, 0.6 },
    { 0.0, 1.0, 0.0, 0.6 }, { 0.0, -1.0, 0.0, 0.6 }, { 0.0, 0.0, 1.0, 0.6 },
};
static double sc_metaballs(v3 p)                                                          /* :628-633 */
{
    double d = sd_sphere(v_sub(p, v(BALLS[0][0], BALLS[0][1], BALLS[0][2])), BALLS[0][3]);
    for (int i = 1; i < 6; ++i)
        d = op_smooth_union(d, sd_sphere(v_sub(p, v(BALLS[i][0], BALLS[i][1], BALLS[i][2])), BALLS[i][3]), 0.45);
    return d;
}

typedef double (*sdf_fn)(v3);
#define RMO_NUM_SCENES 20
/* order == get_all_scenes() (catalog.py:640-663) */
static const sdf_fn SCENES[RMO_NUM_SCENES] = {
    sc_sphere, sc_grazing_plane, sc_cube, sc_thin_torus, sc_cylinder, sc_near_miss,
    sc_hollow_cube, sc_smooth_blend, sc_onion, sc_menger, sc_mandelbulb, sc_bad_lipschitz,
    sc_pillars, sc_thin_planes, sc_sphere_cloud, sc_bumpy_sphere, sc_gyroid, sc_capped_torus,
    sc_box_lattice, sc_metaballs,
};

/* ---- strategies ------------------------------------------------------------ */

typedef struct {
    int32_t max_iterations;   /* config.py:21 */
    double hit_threshold;     /* config.py:22 */
    double max_distance;      /* config.py:23 */
    double lipschitz;         /* SegmentTracing.lipschitz, wired by main.py:58-61 */
    /* Constructor arguments of the reference's strategies (defaults in the comments), then four
     * constants their march() bodies hold as literals.  Same order as RmStrategyParams (include/rm_hip.h). */
    double omega;                   /* RelaxedSphereTracing(omega=1.2)                 relaxed_sphere.py:17 */
    double ar_omega_min;            /* AutoRelaxedSphereTracing(omega_min=1.0,          auto_relaxed.py:21-23 */
    double ar_omega_max;            /*   omega_max=2.0, */
    double ar_smoothing;            /*   smoothing=0.7, */
    double ar_growth_rate;          /*   growth_rate=1.05, */
    double ar_decay_rate;           /*   decay_rate=0.7) */
    double beta;                    /* SlopeAutoRelaxed(beta=0.3)                      slope_auto_relaxed.py:25 */
    double overstep_min_step;       /* OverstepBisectTracing(min_step_factor=0.01,      overstep_bisect.py:18 */
    double hybrid_stuck_step_ratio; /* AdaptiveHybridTracing(stuck_step_ratio=0.001,    adaptive_hybrid.py:17-19 */
    double hybrid_min_step;         /*   min_step_factor=0.005, */
    double margin;                  /* `margin = 0.05`, a literal of SkippingSpheresTracing.march    skipping_spheres.py:30 */
    double ar_omega_init;           /* `omega = 1.2`, a literal of AutoRelaxedSphereTracing.march    auto_relaxed.py:41 */
    int32_t overstep_bisection_steps; /* bisection_steps=16)                            overstep_bisect.py:18 */
    int32_t hybrid_stuck_threshold;   /* stuck_threshold=5)                             adaptive_hybrid.py:17 */
    int32_t segment_bisection_steps;  /* `range(8)`, a literal of SegmentTracing.march  segment_tracing.py:79 */
    int32_t revaa_bisection_steps;    /* `range(8)`, a literal of RevAAApproxTracing.march  rev_affine.py:70 */
    /* uniforms only the reference's fragment shader has */
    double step_scale;                /* `stepScale` (1.0): standard() and dense_march()   gpu/shaders/strategies.glsl:24,47,570 */
    double dense_min_step;            /* `minStep` as dense_march reads it                 gpu/shaders/strategies.glsl:570 */
} rmo_cfg;

typedef struct { int hit; double t; int32_t iterations; double final_sdf; } result_t;

static result_t mk(int hit, double t, int32_t it, double fs)
{
    result_t r; r.hit = hit; r.t = t; r.iterations = it; r.final_sdf = fs; return r;
}

/* strategies/standard_sphere.py:24-49 */
static result_t st_standard(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0; int32_t iterations = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        t += d * c->step_scale;      /* standard_sphere.py:41 `t += d`; the shader's stepScale (strategies.glsl:47), 1.0 by default */
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/relaxed_sphere.py:28-70 */
static result_t st_relaxed(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0, prev_d = 0.0, omega = c->omega; int32_t iterations = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (d < 0.0) { t += d; omega = 1.0; prev_d = fabs(d); continue; }
        double step = d * omega;
        if (i > 0 && (prev_d + d) < prev_d * omega) { step = d; omega = 1.0; }
        t += step;
        prev_d = d;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/auto_relaxed.py:38-90 */
static result_t st_auto_relaxed(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    const double omega_min = c->ar_omega_min, omega_max = c->ar_omega_max, smoothing = c->ar_smoothing,
                 growth = c->ar_growth_rate, decay = c->ar_decay_rate;
    double t = 0.0, omega = c->ar_omega_init, prev_d = INFINITY, ema = 1.0; int32_t iterations = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (prev_d > 1e-10 && i > 0) {
            double ratio = d / prev_d;
            ema = smoothing * ema + (1.0 - smoothing) * ratio;
            if (ema < 0.8) omega = py_max(omega_min, omega * decay);
            else if (ema > 1.0) omega = py_min(omega_max, omega * growth);
        }
        double step = d * omega;
        if (d < 0.0) { t += d; omega = omega_min; prev_d = fabs(d); continue; }
        t += step;
        prev_d = d;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/slope_auto_relaxed.py:41-115 */
static result_t st_slope(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    const double beta = c->beta;
    double t = 0.0; int32_t iterations = 0;
    double r = sdf(ray_at(ray, t));
    double z = r, m = -1.0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        if (fabs(r) < c->hit_threshold) return mk(1, t, iterations, r);
        if (t > c->max_distance) break;
        double T = t + z;
        double R = sdf(ray_at(ray, T));
        if (z <= r + fabs(R)) {
            double denom = T - t;
            double M = (denom > 1e-12) ? (R - r) / denom : -1.0;
            m = (1.0 - beta) * m + beta * M;
            t = T;
            r = R;
        } else {
            m = -1.0;
        }
        double denom = 1.0 - m;
        if (denom < 1e-6) denom = 1e-6;
        z = (2.0 * r) / denom;
        if (z < 0.0) z = r;
    }
    return mk(0, t, iterations, r);
}

/* strategies/enhanced_sphere.py:32-86 */
static result_t st_enhanced(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0, prev_t = 0.0, prev_d = INFINITY; int32_t iterations = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        double step = d;
        if (i > 0 && prev_d > d && d > 0.0 && (prev_d - d) > 1e-10) {
            double dt = t - prev_t;
            double predicted = d * dt / (prev_d - d);
            if (0.0 < predicted && predicted < d * 3.0) step = predicted;
        }
        if (d < 0.0) { t = (prev_t + t) * 0.5; prev_d = fabs(d); continue; }
        prev_t = t;
        prev_d = d;
        t += step;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/curvature_auto_relaxed.py:23-89 */
static result_t st_curvature(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0; int32_t iterations = 0;
    double th[3] = { 0.0, 0.0, 0.0 }, dh[3] = { 0.0, 0.0, 0.0 };
    int hist = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        th[0] = th[1]; th[1] = th[2]; th[2] = t;
        dh[0] = dh[1]; dh[1] = dh[2]; dh[2] = d;
        hist = (hist + 1 < 3) ? hist + 1 : 3;
        double step = d;
        if (hist == 3) {
            double t1 = th[0], t2 = th[1], t3 = th[2];
            double d1 = dh[0], d2 = dh[1], d3 = dh[2];
            if (fabs(t3 - t2) > 1e-5 && fabs(t2 - t1) > 1e-5) {
                /* Python float division raises ZeroDivisionError on a zero divisor
                 * (`except: pass` keeps step = d).  The six divisors are +-(d1-d2),
                 * +-(d1-d3), +-(d2-d3); term1 is evaluated first and already touches
                 * (d1-d2) and (d1-d3); term2 adds (d2-d3).  Any zero -> no prediction. */
                if ((d1 - d2) != 0.0 && (d1 - d3) != 0.0 && (d2 - d1) != 0.0 && (d2 - d3) != 0.0 &&
                    (d3 - d1) != 0.0 && (d3 - d2) != 0.0) {
                    double term1 = t1 * ((0 - d2) / (d1 - d2)) * ((0 - d3) / (d1 - d3));
                    double term2 = t2 * ((0 - d1) / (d2 - d1)) * ((0 - d3) / (d2 - d3));
                    double term3 = t3 * ((0 - d1) / (d3 - d1)) * ((0 - d2) / (d3 - d2));
                    double t_pred = term1 + term2 + term3;
                    double pred_step = t_pred - t;
                    if (d < dh[1] && 0.0 < pred_step && pred_step < 3.0 * d) step = pred_step;
                }
            }
        }
        t += step;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/overstep_bisect.py:30-121 */
static result_t st_overstep_bisect(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    const double min_step = c->overstep_min_step; const int bis = c->overstep_bisection_steps;
    double t = 0.0, t_near = 0.0, t_far = -1.0; int32_t iterations = 0;
    int32_t budget = c->max_iterations - bis;
    for (int32_t i = 0; i < budget; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (d > 0.0) { t_near = t; t += py_max(d, min_step); }
        else { t_far = t; break; }
        if (t > c->max_distance) return mk(0, t, iterations, sdf(ray_at(ray, t)));
    }
    if (t_far > 0.0) {
        for (int j = 0; j < bis; ++j) {
            iterations += 1;
            double t_mid = (t_near + t_far) * 0.5;
            double d = sdf(ray_at(ray, t_mid));
            if (fabs(d) < c->hit_threshold) return mk(1, t_mid, iterations, d);
            if (d > 0.0) t_near = t_mid; else t_far = t_mid;
            if ((t_far - t_near) < c->hit_threshold) {
                t_mid = (t_near + t_far) * 0.5;
                return mk(1, t_mid, iterations, sdf(ray_at(ray, t_mid)));
            }
        }
        double t_final = (t_near + t_far) * 0.5;
        double d = sdf(ray_at(ray, t_final));
        return mk(fabs(d) < c->hit_threshold * 10, t_final, iterations, d);
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/adaptive_hybrid.py:35-149 */
static result_t st_hybrid(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    const int stuck_threshold = c->hybrid_stuck_threshold;
    const double stuck_ratio = c->hybrid_stuck_step_ratio, min_step = c->hybrid_min_step;
    enum { SPHERE, OVERSTEP, BISECT } mode = SPHERE;
    double t = 0.0, t_near = 0.0, t_far = -1.0; int32_t iterations = 0; int small = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (mode == SPHERE) {
            double step = d;
            if (d > 0.0 && d < stuck_ratio * py_max(t, 1.0)) small += 1; else small = 0;
            if (small >= stuck_threshold) { mode = OVERSTEP; t_near = t; t_far = -1.0; small = 0; continue; }
            if (d < 0.0) { t_far = t; t_near = py_max(0.0, t + d); mode = BISECT; continue; }
            t += step;
        } else if (mode == OVERSTEP) {
            if (d > 0.0) { t_near = t; t += py_max(d, min_step); }
            else { t_far = t; mode = BISECT; continue; }
        } else {
            if (t_far < 0.0) { mode = SPHERE; continue; }
            double t_mid = (t_near + t_far) * 0.5;
            d = sdf(ray_at(ray, t_mid));
            if (fabs(d) < c->hit_threshold) return mk(1, t_mid, iterations, d);
            if (d > 0.0) t_near = t_mid; else t_far = t_mid;
            if ((t_far - t_near) < c->hit_threshold) {
                t = (t_near + t_far) * 0.5;
                return mk(1, t, iterations, sdf(ray_at(ray, t)));
            }
            t = t_mid;
            continue;
        }
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/segment_tracing.py:37-113 */
static result_t st_segment(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0, L = c->lipschitz; int32_t iterations = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (d < 0.0) { t -= fabs(d) * 0.5; continue; }
        double candidate = d / L;
        double d_end = sdf(ray_at(ray, t + candidate));
        iterations += 1;
        if (fabs(d_end) < c->hit_threshold) { t += candidate; return mk(1, t, iterations, d_end); }
        if (d_end < 0.0) {
            double t_lo = t, t_hi = t + candidate;
            for (int k = 0; k < c->segment_bisection_steps; ++k) {
                iterations += 1;
                double t_mid = (t_lo + t_hi) * 0.5;
                double d_mid = sdf(ray_at(ray, t_mid));
                if (fabs(d_mid) < c->hit_threshold) return mk(1, t_mid, iterations, d_mid);
                if (d_mid > 0.0) t_lo = t_mid; else t_hi = t_mid;
            }
            t = (t_lo + t_hi) * 0.5;
            return mk(1, t, iterations, sdf(ray_at(ray, t)));
        }
        double extended = candidate + d_end / L;
        t += extended;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/skipping_spheres.py:25-72 */
static result_t st_skipping(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    const double margin = c->margin;
    double t = 0.0; int32_t iterations = 0;
    int32_t coarse = (c->max_iterations * 2) / 3;
    int32_t fine = c->max_iterations - coarse;
    for (int32_t i = 0; i < coarse; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t)) - margin;
        if (d < c->hit_threshold) break;
        t += d;
        if (t > c->max_distance) break;
    }
    for (int32_t j = 0; j < fine; ++j) {
        iterations += 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (d < 0.0) { t = py_max(0.0, t + d); continue; }
        t += d;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* strategies/rev_affine.py:37-97 */
static result_t st_revaa(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0; int32_t iterations = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        iterations = i + 1;
        double d = sdf(ray_at(ray, t));
        if (fabs(d) < c->hit_threshold) return mk(1, t, iterations, d);
        if (d < 0.0) { t = py_max(0.0, t + d); continue; }
        double next_t = t + py_max(d, c->hit_threshold);
        double d_hi = sdf(ray_at(ray, next_t));
        double lo = py_min(d, d_hi), hi = py_max(d, d_hi);
        if (lo <= 0.0 && hi >= 0.0) {
            double a = t, b = next_t;
            for (int j = 0; j < c->revaa_bisection_steps; ++j) {
                double mid = 0.5 * (a + b);
                double dm = sdf(ray_at(ray, mid));
                iterations += 1;
                if (fabs(dm) < c->hit_threshold) return mk(1, mid, iterations, dm);
                if (dm > 0.0) a = mid; else b = mid;
            }
            double tm = 0.5 * (a + b);
            return mk(1, tm, iterations, sdf(ray_at(ray, tm)));
        }
        t = next_t;
        if (t > c->max_distance) break;
    }
    return mk(0, t, iterations, sdf(ray_at(ray, t)));
}

/* ---- the two strategies that exist only in the reference's fragment shader ----------------------------------
 * PARITY UNPINNED: there is no Python statement of them and the shader computes in fp32 (moderngl is absent here), so
 * no fixture of the reference pins these two functions.  They restate the shader text in binary64 on this camera;
 * what the tests check with them is that the HIP state machines follow the same text. */

/* gpu/shaders/strategies.glsl:508-541 safe_relaxed */
static result_t st_safe_relaxed(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    double t = 0.0, omega_eff = c->omega, prev_radius = 0.0, step_length = 0.0, d = 0.0;
    int32_t it = 0;
    for (int32_t i = 0; i < c->max_iterations; ++i) {
        it = i + 1;
        d = sdf(ray_at(ray, t));
        double radius = fabs(d);
        int sor_fail = (omega_eff > 1.0) && ((radius + prev_radius) < step_length);
        if (sor_fail) { step_length -= omega_eff * step_length; omega_eff = 1.0; }
        else step_length = d * omega_eff;
        prev_radius = radius;
        if (!sor_fail && radius < c->hit_threshold) return mk(1, t, it, d);
        t += step_length;
        if (t > c->max_distance) break;
        if (t < 0.0) t = 0.0;
    }
    return mk(0, t, it, d);
}

/* gpu/shaders/strategies.glsl:559-593 dense_march */
static result_t st_dense_march(const ray_t *ray, sdf_fn sdf, const rmo_cfg *c)
{
    int32_t it = 1;
    double t = 0.0;
    double d = sdf(ray_at(ray, 0.0));
    if (fabs(d) < c->hit_threshold) return mk(1, t, it, d);
    for (int32_t i = 1; i < c->max_iterations; ++i) {
        it = i + 1;
        double prev_t = t, prev_d = d;
        double sc = d * c->step_scale;
        double step = (sc < c->dense_min_step) ? c->dense_min_step : sc;     /* max(d * stepScale, minStep) */
        t += step;
        if (t > c->max_distance) break;
        d = sdf(ray_at(ray, t));
        if (prev_d > 0.0 && d <= 0.0) {
            double a = prev_t, b = t;
            for (int j = 0; j < 30; ++j) {
                double mid = 0.5 * (a + b);
                double dm = sdf(ray_at(ray, mid));
                if (dm > 0.0) a = mid; else b = mid;
            }
            double tm = 0.5 * (a + b);
            return mk(1, tm, it, sdf(ray_at(ray, tm)));
        }
        if (fabs(d) < c->hit_threshold) return mk(1, t, it, d);
    }
    return mk(0, t, it, d);
}

typedef result_t (*strat_fn)(const ray_t *, sdf_fn, const rmo_cfg *);
#define RMO_NUM_STRATEGIES 13
/* [0, 11): order == STRATEGIES dict (strategies/__init__.py:16-28); 11, 12: the shader-only two (unpinned, above) */
static const strat_fn STRATS[RMO_NUM_STRATEGIES] = {
    st_standard, st_relaxed, st_auto_relaxed, st_slope, st_enhanced, st_curvature,
    st_overstep_bisect, st_skipping, st_revaa, st_hybrid, st_segment,
    st_safe_relaxed, st_dense_march,
};

/* ---- camera (core/camera.py:35-41) + frame loop (metrics/collector.py:40-44) */

/* cam[13] = position(3) forward(3) right(3) up(3)... no: pos, fwd, right, up = 12, + half_w, half_h = 14 */
static ray_t camera_ray(const double *cam, int W, int H, int px, int py)
{
    v3 pos = v(cam[0], cam[1], cam[2]), fwd = v(cam[3], cam[4], cam[5]);
    v3 right = v(cam[6], cam[7], cam[8]), up = v(cam[9], cam[10], cam[11]);
    double half_w = cam[12], half_h = cam[13];
    double u = (2.0 * (px + 0.5) / W - 1.0) * half_w;
    double w = (1.0 - 2.0 * (py + 0.5) / H) * half_h;
    v3 dir = v_add(v_add(fwd, v_mul(right, u)), v_mul(up, w));
    ray_t r; r.o = pos; r.d = v_normalized(dir);            /* ray.py:11-13 */
    return r;
}

int rmo_num_scenes(void) { return RMO_NUM_SCENES; }
int rmo_num_strategies(void) { return RMO_NUM_STRATEGIES; }

int rmo_sdf_eval(int scene, const double *xyz, size_t n, double *out)
{
    if (scene < 0 || scene >= RMO_NUM_SCENES) return -1;
    for (size_t i = 0; i < n; ++i) out[i] = SCENES[scene](v(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    return 0;
}

/* Rays given explicitly; directions are normalised the way Ray.__init__ does. */
int rmo_march_rays(int scene, int strategy, const rmo_cfg *cfg, const double *origins, const double *dirs,
                   size_t n, uint8_t *hit, double *t, int32_t *iters, double *final_sdf)
{
    if (scene < 0 || scene >= RMO_NUM_SCENES) return -1;
    if (strategy < 0 || strategy >= RMO_NUM_STRATEGIES) return -2;
    for (size_t i = 0; i < n; ++i) {
        ray_t r;
        r.o = v(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        r.d = v_normalized(v(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
        result_t res = STRATS[strategy](&r, SCENES[scene], cfg);
        hit[i] = (uint8_t)res.hit; t[i] = res.t; iters[i] = res.iterations; final_sdf[i] = res.final_sdf;
    }
    return 0;
}

/* Rows [row0, row0+rows) of a W x H frame, row-major, row 0 = top.  Outputs hold
 * rows*W elements.  `t` is the raw termination parameter of EVERY ray (the
 * reference's depth_map is t if hit else 0.0, types.py:93).  nthreads > 1 splits
 * rows with OpenMP (used only for the multi-core cpu_baseline figure). */
int rmo_render(int scene, int strategy, const rmo_cfg *cfg, const double *cam14, int W, int H, int row0,
               int rows, int nthreads, uint8_t *hit, double *t, int32_t *iters, double *final_sdf)
{
    if (scene < 0 || scene >= RMO_NUM_SCENES) return -1;
    if (strategy < 0 || strategy >= RMO_NUM_STRATEGIES) return -2;
    if (W <= 0 || H <= 0 || row0 < 0 || rows < 0 || row0 + rows > H) return -3;
    sdf_fn sdf = SCENES[scene];
    strat_fn st = STRATS[strategy];
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int r = 0; r < rows; ++r) {
        int py = row0 + r;
        for (int px = 0; px < W; ++px) {
            ray_t ray = camera_ray(cam14, W, H, px, py);
            result_t res = st(&ray, sdf, cfg);
            size_t k = (size_t)r * (size_t)W + (size_t)px;
            hit[k] = (uint8_t)res.hit; t[k] = res.t; iters[k] = res.iterations;
            if (final_sdf) final_sdf[k] = res.final_sdf;
        }
    }
    return 0;
}
