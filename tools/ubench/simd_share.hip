// Micro-benchmark (developer tool): do the wavefronts of one workgroup, placed on different SIMDs of a compute unit,
// run independently?  Every wave runs the same loop (a dependent fp64 chain with scalar constant moves in between, the
// instruction mix of the libm restatements); the time of ONE wave is printed for 1, 2, 3, 4 waves per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/simd_share.hip -o tools/ubench/simd_share.exe
#include <hip/hip_runtime.h>
#include <cstdio>

// dense, INDEPENDENT instruction streams (4 chains per wave): does the rate of one wave depend on how many other waves
// of the workgroup -- on other SIMDs -- issue the same kind of instruction?  KIND 0 v_fma_f64, 1 v_add_f64, 2 v_mul_f64,
// 3 v_fma_f32, 4 v_cndmask_b32 (a select), 5 v_add_u32, 6 s_mov_b32 pairs
template <int KIND>
__global__ void dense(double* out, double a, double b, int n, long long* cyc, unsigned* where)
{
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float f0 = (float)x0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); }
            else if (KIND == 1) { x0 += b; x1 += b; x2 += b; x3 += b; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (KIND == 2) { x0 *= a; x1 *= a; x2 *= a; x3 *= a; asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (KIND == 3) { f0 = __builtin_fmaf(f0, (float)a, (float)b); f1 = __builtin_fmaf(f1, (float)a, (float)b); f2 = __builtin_fmaf(f2, (float)a, (float)b); f3 = __builtin_fmaf(f3, (float)a, (float)b); }
            else if (KIND == 4) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc"); }
            else if (KIND == 5) { asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); }
            else if (KIND == 7) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %2, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %3, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %0, s[20:21]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "s20", "s21"); }
            else if (KIND == 15) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n\tv_cndmask_b32_e64 %1, %1, %2, vcc\n\tv_cndmask_b32_e64 %2, %2, %3, vcc\n\tv_cndmask_b32_e64 %3, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc"); }
            else if (KIND == 16) { asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc\n\tv_addc_co_u32 %2, vcc, %2, %3, vcc\n\tv_addc_co_u32 %3, vcc, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc"); }
            else if (KIND == 17) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e32 %1, %1, %2, vcc\n\tv_cndmask_b32_e64 %2, %2, %3, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %0, s[20:21]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "vcc", "s20", "s21"); }
            else if (KIND == 8) { asm volatile("v_bfi_b32 %0, %4, %0, %1\n\tv_bfi_b32 %1, %4, %1, %2\n\tv_bfi_b32 %2, %4, %2, %3\n\tv_bfi_b32 %3, %4, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(threadIdx.x * 0x01010101u)); }
            else if (KIND == 9) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cmp_lt_u32 vcc, %1, %2\n\tv_cmp_lt_u32 vcc, %2, %3\n\tv_cmp_lt_u32 vcc, %3, %0" : : "v"(u0), "v"(u1), "v"(u2), "v"(u3) : "vcc"); u0 += u; }
            else if (KIND == 10) { asm volatile("v_cmp_lt_f64 s[20:21], %0, %1\n\tv_cmp_lt_f64 s[22:23], %1, %2\n\tv_cmp_lt_f64 s[24:25], %2, %3\n\tv_cmp_lt_f64 s[26:27], %3, %0" : : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); x0 += 1.0; }
            else if (KIND == 11) { asm volatile("v_and_or_b32 %0, %0, %4, %1\n\tv_and_or_b32 %1, %1, %4, %2\n\tv_and_or_b32 %2, %2, %4, %3\n\tv_and_or_b32 %3, %3, %4, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(threadIdx.x * 0x01010101u)); }
            else if (KIND == 12) { asm volatile("v_max_f64 %0, %0, %1\n\tv_min_f64 %1, %1, %2\n\tv_max_f64 %2, %2, %3\n\tv_min_f64 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); }
            else if (KIND == 13) { asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); }
            else if (KIND == 14) { asm volatile("v_ashrrev_i32 %0, 31, %1\n\tv_xor_b32 %1, %1, %2\n\tv_lshlrev_b32 %2, 1, %3\n\tv_sub_u32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)); }
            else { unsigned s0, s1, s2, s3; asm volatile("s_mov_b32 %0, 0x12345678\n\ts_mov_b32 %1, 0x3ff00000\n\ts_mov_b32 %2, 0x12345679\n\ts_mov_b32 %3, 0x3ff00001" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3)); u0 += s0 ^ s1 ^ s2 ^ s3; }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3;
    if ((threadIdx.x & 63) == 0) {
        cyc[threadIdx.x >> 6] = t1 - t0;
        where[threadIdx.x >> 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}
template <int KIND>
void run_dense(const char* name)
{
    double* out; long long* cyc; unsigned* where;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 64); hipMalloc(&where, 32);
    const int n = 8192;
    for (int waves = 1; waves <= 4; ++waves) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((dense<KIND>), dim3(1), dim3(64 * waves), 0, 0, out, 1.0000001, 1e-9, n, cyc, where);
        hipDeviceSynchronize();
        long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
        printf("dense %-24s %d wave(s):", name, waves);
        for (int i = 0; i < waves; ++i) printf("  %.2f", (double)h[i] * 10.0 * 2.4 / (n * 8.0 * 4.0));
        printf("   cycles (2.4 GHz) per instruction and wave\n");
    }
}

template <int MIX>
__global__ void k(double* out, double a, double b, int n, long long* cyc, unsigned* where)
{
    double x = a + threadIdx.x;
    __shared__ double tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = 1.0 + i * 1e-3;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MIX == 0) x = __builtin_fma(x, a, b);                       // VALU only (operands in registers)
            else if (MIX == 1) {                                            // + two scalar moves per step (a literal constant)
                unsigned lo, hi;
                asm volatile("s_mov_b32 %0, 0x12345678\n\ts_mov_b32 %1, 0x3ff00000" : "=s"(lo), "=s"(hi));
                x = __builtin_fma(x, __hiloint2double((int)hi, (int)lo), b);
            } else if (MIX == 2) {                                          // + a dependent LDS gather per step (a table row chosen by the value)
                const unsigned idx = ((unsigned)__double2hiint(x) >> 8) & 127u;
                x = __builtin_fma(tab[idx * 2 + 1], a, x * 1e-9 + b);
            } else if (MIX == 3) {                                          // a reciprocal (quarter-rate unit) per step
                x = __builtin_fma(__builtin_amdgcn_rcp(x), a, b + 1.0);
            } else if (MIX == 4) {                                          // a wave-uniform branch per step
                if (__builtin_amdgcn_readfirstlane(__double2hiint(x)) & 0x100000) x = __builtin_fma(x, a, b); else x = __builtin_fma(x, b, a);
                asm volatile("" : "+v"(x));
            } else {                                                        // an fp64 division per step
                x = b / x + a;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) {
        cyc[threadIdx.x >> 6] = t1 - t0;
        where[threadIdx.x >> 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

template <int MIX>
void run(const char* name)
{
    double* out; long long* cyc; unsigned* where;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 64); hipMalloc(&where, 32);
    const int n = 4096;
    for (int waves = 1; waves <= 4; ++waves) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MIX>), dim3(1), dim3(64 * waves), 0, 0, out, 1.0000001, 1e-9, n, cyc, where);
        hipDeviceSynchronize();
        long long h[4]; unsigned w[4];
        hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost); hipMemcpy(w, where, 16, hipMemcpyDeviceToHost);
        printf("%-34s %d wave(s):", name, waves);
        for (int i = 0; i < waves; ++i) printf("  simd %u %.2f ns/step", (w[i] >> 4) & 3, (double)h[i] * 10.0 / (n * 16.0));
        printf("\n");
    }
}

int main()
{
    run_dense<0>("v_fma_f64"); run_dense<1>("v_add_f64"); run_dense<2>("v_mul_f64"); run_dense<3>("v_fma_f32");
    run_dense<4>("v_cndmask_b32 (vcc)"); run_dense<15>("v_cndmask_b32_e64 (vcc)"); run_dense<17>("cndmask 1 of 4 e32 vcc"); run_dense<16>("v_addc_co_u32 vcc"); run_dense<7>("v_cndmask_b32_e64 (sgpr)"); run_dense<8>("v_bfi_b32"); run_dense<11>("v_and_or_b32");
    run_dense<9>("v_cmp_lt_u32 -> vcc"); run_dense<10>("v_cmp_lt_f64 -> sgpr"); run_dense<12>("v_max/min_f64"); run_dense<13>("v_mov_b32"); run_dense<14>("ashr/xor/shl/sub");
    run_dense<5>("v_add_u32"); run_dense<6>("s_mov_b32 x4 + v_add");
    run<0>("dependent v_fma_f64");
    run<1>("v_fma_f64 + 2 s_mov per step");
    run<2>("v_fma_f64 + dependent LDS gather");
    run<3>("v_rcp_f64 + v_fma_f64");
    run<4>("wave-uniform branch + v_fma_f64");
    run<5>("fp64 division");
    return 0;
}
