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
RM_HD double length(vec3 a) { return rm_pow(a.x * a.x + a.y * a.y + a.z * a.z, 0.5); }
RM_HD vec3 normalized(vec3 a)                                                          // vec3.py:52-56
{
    double l = length(a);
    if (l < 1e-12) return v3(0.0, 0.0, 0.0);
    double inv = 1.0 / l;                                                              // vec3.py:32-34
    return v3(a.x * inv, a.y * inv, a.z * inv);
}

// MarchConfig (config.py:19-29): only these three are read by the CPU strategies.
// `lipschitz` is SegmentTracing.lipschitz as wired by main.py:58-61.
struct MarchCfg {
    double hit_threshold;
    double max_distance;
    double lipschitz;
    int32_t max_iterations;
    int32_t full;  // 1: also produce final_sdf (costs the reference's tail evaluations)
};

}  // namespace rm
