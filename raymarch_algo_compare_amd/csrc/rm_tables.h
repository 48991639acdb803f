// rm_tables.h -- access to the gathered libm tables (rm_libm_tables.h, generated): constant memory, or the LDS
// mirrors the render kernels fill (rm_load_tables, rm_kernels.h).
//
// Row strides of the mirrors (a wavefront gathers one row per lane; bank = (byte address / 4) mod 64):
//   * pow's log table, the exp table and the log table keep glibc's rows (4, 2 and 2 doubles): a row is fetched with
//     one aligned ds_read_b128.  Odd strides (3 doubles: 32 distinct bank positions instead of 8-16) were tried in
//     round 2: the conflict counter went down, but rows stop being 16-byte aligned, the fetch becomes ds_read2_b64
//     plus a multiply for the address, and the pow-bound scenes got SLOWER (Pillar Forest 1.82 -> 1.95-2.04 ms,
//     same box; Mandelbulb unchanged) -- reverted.
//   * __sincostab rows (4 doubles in glibc) are padded to 5: rows i and i + 8 no longer share their banks
//     (LDS bank conflicts of the Mandelbulb frame 53 % -> 43 % of LDS-active cycles, profiles/r02_v2).
//   * the 13- and 7-double rows of the acos / atan tables already have odd strides.
// Values are untouched: same doubles, same operations, same bits.
#pragma once

#include "rm_libm_tables.h"

namespace rm {

#if defined(__HIP_DEVICE_COMPILE__) && defined(RM_TABLES_IN_LDS)
constexpr int kPowLogStride = 4, kExpStride = 2, kLogStride = 2, kSinCosStride = 5;
__shared__ double rm_s_pow_log_tab[128 * kPowLogStride];     // invc, (pad), logc, logctail  (glibc's row)
__shared__ uint64_t rm_s_exp_tab[128 * kExpStride];          // tail, sbits
__shared__ double rm_s_log_tab[128 * kLogStride];            // invc, logc
__shared__ double rm_s_sincostab[110 * kSinCosStride];       // sn, ssn, cs, ccs, (unused)
__shared__ double rm_s_asncs[2808];
__shared__ double rm_s_inroot[128];
__shared__ double rm_s_cij[1687];
#define rm_asncs rm_s_asncs
#define rm_inroot rm_s_inroot
#define rm_cij rm_s_cij
// field f of row i
__device__ __forceinline__ double tab_pow_log(int i, int f) { return rm_s_pow_log_tab[kPowLogStride * i + (f ? f + 1 : 0)]; }     // f: 0 invc, 1 logc, 2 logctail
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
