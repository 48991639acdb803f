// rm_core.h -- scalar semantics + vec3 shared by the gfx950 kernels.
//
// Everything on the hot path is IEEE binary64 evaluated in the reference's
// order (CPython floats); the translation units are built with
// -ffp-contract=off so `o + d*t` stays a multiply and an add, never an FMA.
// Citations are to /root/reference/raymarching_benchmark/.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RM_HD __host__ __device__ __forceinline__
#else
// Host-only build: used exclusively by tests/native (a CPU check of these
// headers against the oracle; never loaded by the product).
#define RM_HD inline __attribute__((always_inline))
#endif

#include "rm_math.h"

namespace rm {

// builtin max()/min(): keep the FIRST argument unless the second compares
// strictly greater/less (ties, signed zeros and NaNs resolve like CPython).
RM_HD double py_max(double a, double b) { return (b > a) ? b : a; }
RM_HD double py_min(double a, double b) { return (b < a) ? b : a; }

// float.__mod__ for a positive power-of-two divisor (2.0 and 0.5 are the only
// divisors on the path: catalog.py:221-225, :374, primitives.py:102-108).
// fmod(a, 2^k) == a - trunc(a / 2^k) * 2^k exactly (every step is exact), then
// CPython's float_rem moves a negative remainder up by b and returns +0.0 for
// a zero remainder.
RM_HD double py_mod_pow2(double a, double b)
{
    double m = a - rm_trunc(a / b) * b;
    if (m != 0.0) {
        if (m < 0.0) m += b;
    } else {
        m = 0.0;
    }
    return m;
}

struct vec3 {
    double x, y, z;
};

RM_HD vec3 v3(double x, double y, double z)
{
    vec3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}
RM_HD vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   // vec3.py:17-18
RM_HD vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   // vec3.py:20-21
RM_HD vec3 operator*(vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }       // vec3.py:23-24
RM_HD double dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }         // vec3.py:36-37
// vec3.py:46-47: (x*x + y*y + z*z) ** 0.5 -- float_pow, i.e. libm pow(s, 0.5)
RM_HD double pow_half(double x) { return rm_pow(x, 0.5); }
RM_HD double length(vec3 a) { return pow_half(a.x * a.x + a.y * a.y + a.z * a.z); }
// The same values for call sites on a frame's critical chain: the full pow in a busy wavefront, the guarded square
// root in a nearly empty one (rm_math_pow.h: rm_pow_half) -- same bits either way.  Only where it measured faster
// (MI355X, 1920x1080): Mandelbulb (single launch 10.1 -> 9.6 ms: every wave of a team repeats the trip's length), the
// two torus primitives (Thin Torus -4 %, Capped Torus -6 %).  Elsewhere the wave-uniform branch it brings costs more
// than the sparse tail saves: it separates the independent pow chains of multi-primitive scenes (Sphere Cloud and
// Bumpy Sphere +34 %, Pillar Forest +7 %, Menger +4 %).
RM_HD double pow_half_a(double x) { return rm_pow_half<true>(x); }
RM_HD double length_a(vec3 a) { return pow_half_a(a.x * a.x + a.y * a.y + a.z * a.z); }
RM_HD vec3 normalized(vec3 a)                                                          // vec3.py:52-56
{
    double l = length(a);
    if (l < 1e-12) return v3(0.0, 0.0, 0.0);
    double inv = 1.0 / l;                                                              // vec3.py:32-34
    return v3(a.x * inv, a.y * inv, a.z * inv);
}

// Constructor arguments of the reference's strategies (defaults in the comments) and four constants their
// march() bodies hold as literals; RmStrategyParams of the C ABI (include/rm_hip.h), same order.
struct StratParams {
    double omega;                    // RelaxedSphereTracing(omega=1.2)              relaxed_sphere.py:17
    double ar_omega_min;             // AutoRelaxedSphereTracing(omega_min=1.0,       auto_relaxed.py:21-23
    double ar_omega_max;             //   omega_max=2.0,
    double ar_smoothing;             //   smoothing=0.7,
    double ar_growth_rate;           //   growth_rate=1.05,
    double ar_decay_rate;            //   decay_rate=0.7)
    double beta;                     // SlopeAutoRelaxed(beta=0.3)                   slope_auto_relaxed.py:25
    double overstep_min_step;        // OverstepBisectTracing(min_step_factor=0.01,   overstep_bisect.py:18
    double hybrid_stuck_step_ratio;  // AdaptiveHybridTracing(stuck_step_ratio=0.001, adaptive_hybrid.py:17-19
    double hybrid_min_step;          //   min_step_factor=0.005,
    double margin;                   // `margin = 0.05` in SkippingSpheresTracing.march   skipping_spheres.py:30
    double ar_omega_init;            // `omega = 1.2` in AutoRelaxedSphereTracing.march   auto_relaxed.py:41
    int32_t overstep_bisection_steps;   //   bisection_steps=16)                      overstep_bisect.py:18
    int32_t hybrid_stuck_threshold;     //   stuck_threshold=5,                       adaptive_hybrid.py:17
    int32_t segment_bisection_steps;    // `range(8)` in SegmentTracing.march         segment_tracing.py:79
    int32_t revaa_bisection_steps;      // `range(8)` in RevAAApproxTracing.march     rev_affine.py:70
    double step_scale;               // `stepScale` uniform: standard / dense_march (1.0)   gpu/shaders/strategies.glsl:24,47,570
    double dense_min_step;           // `minStep` uniform as dense_march reads it (1e-4)    gpu/shaders/strategies.glsl:570
};

RM_HD StratParams default_strat_params()
{
    StratParams p;
    p.omega = 1.2;
    p.ar_omega_min = 1.0; p.ar_omega_max = 2.0; p.ar_smoothing = 0.7; p.ar_growth_rate = 1.05; p.ar_decay_rate = 0.7;
    p.beta = 0.3;
    p.overstep_min_step = 0.01;
    p.hybrid_stuck_step_ratio = 0.001; p.hybrid_min_step = 0.005;
    p.margin = 0.05;
    p.ar_omega_init = 1.2;
    p.overstep_bisection_steps = 16; p.hybrid_stuck_threshold = 5; p.segment_bisection_steps = 8; p.revaa_bisection_steps = 8;
    p.step_scale = 1.0; p.dense_min_step = 1e-4;
    return p;
}

// MarchConfig (config.py:19-29): only these three are read by the CPU strategies.
// `lipschitz` is SegmentTracing.lipschitz as wired by main.py:58-61; `prm` the strategy's own constants.
// A kernel that renders ONE frame reads this record from its arguments (scalar registers); a batch
// kernel keeps the fields its strategy reads per lane (rm_kernels.h).
struct MarchCfg {
    double hit_threshold;
    double max_distance;
    double lipschitz;
    int32_t max_iterations;
    int32_t full;  // 1: also produce final_sdf (costs the reference's tail evaluations)
    StratParams prm;
};

}  // namespace rm
