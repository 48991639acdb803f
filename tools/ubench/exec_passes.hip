// Micro-benchmark (developer tool): does a quarter-rate fp64 instruction cost fewer issue cycles when few lanes are live?
// Twelve waves on one compute unit (three per SIMD) run four independent v_fma_f64 chains each; the live lanes are all 64,
// the first 16, every fourth lane, or one.  If the hardware skipped the 16-lane passes that have no live lane, a wave whose
// last rays were moved into one quarter of its lanes would issue its fp64 work four times faster.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/exec_passes.hip -o exec_passes.exe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(768) void run(int mode, int n, double a, double b, double* sink, long long* ticks)
{
    const int lane = threadIdx.x & 63;
    const bool live = mode == 0 ? true : mode == 1 ? lane < 16 : mode == 2 ? (lane & 3) == 0 : mode == 3 ? lane == 0 : lane < 32;
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if (live) {
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    if (x0 + x1 + x2 + x3 == 42.0) sink[0] = x0;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main()
{
    double* sink; long long* ticks;
    if (hipMalloc(&sink, 8) != hipSuccess || hipMalloc(&ticks, 8) != hipSuccess) return 1;
    const int n = 20000;
    const char* names[] = { "all 64 lanes", "lanes 0-15", "every 4th lane", "one lane", "lanes 0-31" };
    for (int rep = 0; rep < 2; ++rep)
        for (int waves : { 4, 12 })
            for (int mode = 0; mode < 5; ++mode) {
                hipLaunchKernelGGL(run, dim3(1), dim3(64 * waves), 0, 0, mode, n, 1.0000001, 1e-9, sink, ticks);
                (void)hipDeviceSynchronize();
                long long t;
                (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
                printf("%2d waves on one CU, live = %-15s: %.2f ns per v_fma_f64 of one wave (%.2f cycles at 2.4 GHz)\n", waves, names[mode],
                       t * 10.0 / (32.0 * n), t * 10.0 / (32.0 * n) * 2.4);
            }
    return 0;
}
