// rm_tables.h -- access to the gathered libm tables (rm_libm_tables.h, generated): constant memory, or the LDS
// mirrors the render kernels fill (rm_load_tables, rm_kernels.h).
//
// The mirrors are RE-LAID-OUT for the LDS banks.  A wavefront gathers one row per lane; ds_read_b64 serves 32 lanes
// per cycle when they hit different banks, bank = (byte address / 4) mod 64.  glibc's rows are 16 or 32 bytes long
// (exp / log: two words, pow's log table and __sincostab: four), so rows i and i + 4 (resp. i + 8) would share their
// banks -- 8 to 16 distinct positions for 64 lanes, the 53 % conflict rate of profiles/r01_v5.  The mirrors use odd row
// strides (3 or 5 doubles: 32 distinct positions) and drop the word of pow's log table that pow never reads.  The
// 13- and 7-double rows of the acos / atan tables already have odd strides.  Values are untouched: same doubles,
// same operations, same bits.
#pragma once

#include "rm_libm_tables.h"

namespace rm {

#if defined(__HIP_DEVICE_COMPILE__) && defined(RM_TABLES_IN_LDS)
constexpr int kPowLogStride = 3, kExpStride = 3, kLogStride = 3, kSinCosStride = 5;
__shared__ double rm_s_pow_log_tab[128 * kPowLogStride];     // invc, logc, logctail        (glibc row: invc, pad, logc, logctail)
__shared__ uint64_t rm_s_exp_tab[128 * kExpStride];          // tail, sbits, (unused)
__shared__ double rm_s_log_tab[128 * kLogStride];            // invc, logc, (unused)
__shared__ double rm_s_sincostab[110 * kSinCosStride];       // sn, ssn, cs, ccs, (unused)
__shared__ double rm_s_asncs[2808];
__shared__ double rm_s_inroot[128];
__shared__ double rm_s_cij[1687];
#define rm_asncs rm_s_asncs
#define rm_inroot rm_s_inroot
#define rm_cij rm_s_cij
// field f of row i
__device__ __forceinline__ double tab_pow_log(int i, int f) { return rm_s_pow_log_tab[kPowLogStride * i + f]; }     // f: 0 invc, 1 logc, 2 logctail
__device__ __forceinline__ uint64_t tab_exp(int i, int f) { return rm_s_exp_tab[kExpStride * i + f]; }              // f: 0 tail, 1 sbits
__device__ __forceinline__ double tab_log(int i, int f) { return rm_s_log_tab[kLogStride * i + f]; }                // f: 0 invc, 1 logc
__device__ __forceinline__ double tab_sincos(int row, int f) { return rm_s_sincostab[kSinCosStride * row + f]; }    // f: 0 sn, 1 ssn, 2 cs, 3 ccs
#else
#define rm_asncs rm_g_asncs
#define rm_inroot rm_g_inroot
#define rm_cij rm_g_cij
#if defined(__HIPCC__)
#define RM_TAB_HD __host__ __device__ __forceinline__
#else
#define RM_TAB_HD inline __attribute__((always_inline))
#endif
RM_TAB_HD double tab_pow_log(int i, int f) { return rm_g_pow_log_tab[4 * i + (f ? f + 1 : 0)]; }
RM_TAB_HD uint64_t tab_exp(int i, int f) { return rm_g_exp_tab[2 * i + f]; }
RM_TAB_HD double tab_log(int i, int f) { return rm_g_log_tab[2 * i + f]; }
RM_TAB_HD double tab_sincos(int row, int f) { return rm_g_sincostab[4 * row + f]; }
#endif
#define rm_pow_log_head rm_g_pow_log_head
#define rm_exp_head rm_g_exp_head
#define rm_log_head rm_g_log_head

}  // namespace rm
