// rm_math_trig.h -- sin, cos, acos, atan2, log of the path (catalog.py:277-293, :510-513).
//
// STATUS: PLATFORM (not yet exact).  These forward to the device math library
// (OCML) in the HIP build and to libm in the host check build; both are < 1 ulp
// but not bit-identical to glibc's, so rays that amplify ulp noise (Mandelbulb
// boundary crawlers) may differ from the reference.  Only Mandelbulb (graded) and
// Gyroid (next) reach these.  Exact restatements replace them one by one.
#pragma once

namespace rm {

RM_MATH_HD double rm_sin(double x) { return ::sin(x); }
RM_MATH_HD double rm_cos(double x) { return ::cos(x); }
RM_MATH_HD double rm_acos(double x) { return ::acos(x); }
RM_MATH_HD double rm_atan2(double y, double x) { return ::atan2(y, x); }
RM_MATH_HD double rm_log(double x) { return ::log(x); }

}  // namespace rm
