// rm_math.h -- the fp64 elementary functions of the path.
//
// The reference reaches libm (glibc 2.35 via CPython) at exactly these call
// sites: float_pow -> pow (vec3.py:46-47 `** 0.5`, primitives.py:24-26 `** 2`,
// catalog.py:280,283 `r ** 7.0`, `r ** 8.0`), math.acos / atan2 / sin / cos /
// log (catalog.py:277-293, :510-513).  Each rm_* below is the device-side
// restatement of that entry point.  Status per function is recorded in
// DESIGN.md ("math parity"); a function marked EXACT reproduces glibc 2.35's
// x86-64 FMA variant bit-for-bit (same table, same operation order, explicit
// fma() only where glibc's build fuses), and is verified against libm on the
// host by tests/test_math_exact.py.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define RM_MATH_HD __host__ __device__ __forceinline__
#else
#define RM_MATH_HD inline __attribute__((always_inline))
#endif

namespace rm {

RM_MATH_HD double rm_fabs(double x) { return __builtin_fabs(x); }
RM_MATH_HD double rm_trunc(double x) { return __builtin_trunc(x); }
RM_MATH_HD double rm_floor(double x) { return __builtin_floor(x); }
RM_MATH_HD double rm_sqrt(double x) { return __builtin_sqrt(x); }
RM_MATH_HD double rm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

}  // namespace rm

#include "rm_math_pow.h"
#include "rm_math_trig.h"
