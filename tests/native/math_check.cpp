// Host-only check build of the product's math headers (tests only; the product
// never loads this).  Exposes each rm_* function so tests can diff it against libm.
#include <stddef.h>
#include "../../raymarch_algo_compare_amd/csrc/rm_math.h"

extern "C" {
void rmc_pow(const double* x, const double* y, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = rm::rm_pow(x[i], y[i]); }
// lane-local part of rm_pow_half: the rounded square root and whether the guard lets it stand for pow(x, 0.5)
void rmc_pow_half_guard(const double* x, size_t n, double* root, unsigned char* safe) { for (size_t i = 0; i < n; ++i) { bool ok; root[i] = rm::rm_pow_half_guard(x[i], &ok); safe[i] = ok; } }
void rmc_pow2(const double* x, size_t n, double ya, double yb, double* oa, double* ob) { for (size_t i = 0; i < n; ++i) rm::rm_pow2(x[i], ya, yb, &oa[i], &ob[i]); }
void rmc_sin(const double* x, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = rm::rm_sin(x[i]); }
void rmc_cos(const double* x, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = rm::rm_cos(x[i]); }
void rmc_acos(const double* x, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = rm::rm_acos(x[i]); }
void rmc_atan2(const double* y, const double* x, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = rm::rm_atan2(y[i], x[i]); }
void rmc_log(const double* x, size_t n, double* out) { for (size_t i = 0; i < n; ++i) out[i] = rm::rm_log(x[i]); }
// libm references, called through a volatile function pointer so nothing is folded
void rml_pow(const double* x, const double* y, size_t n, double* out) { double (*volatile f)(double, double) = ::pow; for (size_t i = 0; i < n; ++i) out[i] = f(x[i], y[i]); }
void rml_sin(const double* x, size_t n, double* out) { double (*volatile f)(double) = ::sin; for (size_t i = 0; i < n; ++i) out[i] = f(x[i]); }
void rml_cos(const double* x, size_t n, double* out) { double (*volatile f)(double) = ::cos; for (size_t i = 0; i < n; ++i) out[i] = f(x[i]); }
void rml_acos(const double* x, size_t n, double* out) { double (*volatile f)(double) = ::acos; for (size_t i = 0; i < n; ++i) out[i] = f(x[i]); }
void rml_atan2(const double* y, const double* x, size_t n, double* out) { double (*volatile f)(double, double) = ::atan2; for (size_t i = 0; i < n; ++i) out[i] = f(y[i], x[i]); }
void rml_log(const double* x, size_t n, double* out) { double (*volatile f)(double) = ::log; for (size_t i = 0; i < n; ++i) out[i] = f(x[i]); }
}
