// rm_math_atan.h -- acos, atan2 (catalog.py:277-278).  STATUS: PLATFORM (not yet exact).
#pragma once

namespace rm {

RM_MATH_HD double rm_acos(double x) { return ::acos(x); }
RM_MATH_HD double rm_atan2(double y, double x) { return ::atan2(y, x); }

}  // namespace rm
