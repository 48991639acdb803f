// Micro-benchmark (developer tool): which piece of the exact libm restatements slows down when THREE wavefronts of a
// workgroup (on three SIMDs of one compute unit) run it at the same time?  Every wave runs the same independent loop on
// one live lane; the time of wave 0 is printed for 1 and 3 waves, several launches each (placement varies per launch).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-cgp-select2branch -I raymarch_algo_compare_amd/csrc ...
#include "rm_kernels.h"
#include <cstdio>
#include <vector>
using namespace rm;

template <int WHAT>
__global__ __launch_bounds__(256) void k(int n, double seed, double* out, long long* cyc)
{
    rm_load_tables<SceneMandelbulb>();
    const int lane = lane_id();
    double x = seed + 1e-3 * (threadIdx.x >> 6), acc = 0.0;
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        for (int i = 0; i < n; ++i) {
            const double a = x + acc * 1e-300;
            double r, s, c;
            if constexpr (WHAT == 0) r = rm_acos<true>(a * 0.7);
            else if constexpr (WHAT == 1) { rm_sincos<true>(a * 9.0, &s, &c); r = s + c; }
            else if constexpr (WHAT == 2) r = rm_atan2<true>(a, 0.37);
            else if constexpr (WHAT == 3) { rm_pow2(a, 7.0, 8.0, &s, &c); r = s + c; }
            else if constexpr (WHAT == 4) r = pow_half_a(a);
            else if constexpr (WHAT == 5) r = py_max(-1.0, py_min(1.0, 0.3 / py_max(a, 1e-12)));
            else if constexpr (WHAT == 6) r = rm_log(a);
            else if constexpr (WHAT == 7) r = rm_acos<false>(a * 0.7);
            else if constexpr (WHAT == 8) { rm_sincos<false>(a * 9.0, &s, &c); r = s + c; }
            else r = rm_pow(a, 0.5);
            acc += r;
        }
    }
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = r1 - r0;
}

template <int WHAT>
void run(const char* name)
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 64);
    const int n = 20000;
    printf("%-34s", name);
    for (int waves : { 1, 3, 4 }) {
        printf("  %d wave(s):", waves);
        for (int rep = 0; rep < 4; ++rep) {
            hipLaunchKernelGGL((k<WHAT>), dim3(1), dim3(64 * waves), 0, 0, n, 0.81, out, cyc);
            hipDeviceSynchronize();
            long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
            if (rep) printf(" %.0f", (double)h[0] * 10.0 / n);
        }
    }
    printf("   ns per call\n");
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("acos (band skipping)"); run<7>("acos (all bands)"); run<1>("sincos (band skipping)"); run<8>("sincos (all bands)");
    run<2>("atan2 (band skipping)"); run<3>("pow2 (r^7, r^8)"); run<4>("guarded sqrt (length_a)"); run<9>("pow(x, 0.5)");
    run<5>("division + clamp"); run<6>("log");
    return 0;
}
