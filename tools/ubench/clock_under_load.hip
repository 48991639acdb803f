// Micro-benchmark (developer tool): does the shader clock drop while the whole chip computes?  One workgroup per compute
// unit (forced by its LDS size), four waves = four SIMDs.  Wave 0 times a DEPENDENT chain of v_add_u32 (a lone wave issues
// one per ~5.5 cycles whatever its neighbours on the other SIMDs do -- simd_share.hip), in `segs` segments, against the
// 100 MHz s_memrealtime; waves 1-3 either leave at once or run independent v_fma_f64 streams (the frame's dense phase).
// ns per chain instruction, idle chip against loaded chip = the clock ratio.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_under_load.hip -o tools/ubench/clock_under_load.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(int heavy, int segs, int n, double a, double b, float* ns_per_instr, double* sink)
{
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    if (wave == 0) {
        unsigned u = threadIdx.x;
        for (int sgm = 0; sgm < segs; ++sgm) {
            const long long t0 = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < n; ++i)
                asm volatile("v_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\t"
                             "v_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0\n\tv_add_u32 %0, %0, %0" : "+v"(u));
            const long long t1 = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) ns_per_instr[blockIdx.x * segs + sgm] = (float)((t1 - t0) * 10.0 / (8.0 * n));
        }
        if (u == 12345u) lds[0] = 1;
    } else if (heavy) {
        double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        // about as long as wave 0: the chain instruction takes ~5.5 cycles, four independent fp64 fmas ~16-17
        for (int i = 0; i < segs * n * 8 / 3; ++i) {
            x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
        }
        if (x0 + x1 + x2 + x3 == 42.0) sink[0] = x0;
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, segs = 40, n = 6000;      // a segment ~ 48 000 instructions ~ 0.11 ms at 2.4 GHz
    float* d;
    double* sink;
    hipMalloc(&d, sizeof(float) * cus * segs);
    hipMalloc(&sink, 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    std::vector<float> h(cus * segs);
    for (int rep = 0; rep < 2; ++rep)
        for (int heavy = 0; heavy < 2; ++heavy)
            for (int grid : { 1, cus }) {
                hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 100 * 1024, 0, heavy, segs, n, 1.0000001, 1e-9, d, sink);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), d, sizeof(float) * grid * segs, hipMemcpyDeviceToHost);
                printf("workgroups %3d  fp64 neighbours %d : ns per chain instruction, workgroup 0, segments 0 4 9 19 39 = %.3f %.3f %.3f %.3f %.3f",
                       grid, heavy, h[0], h[4], h[9], h[19], h[39]);
                double m = 0;
                for (int g = 0; g < grid; ++g) m += h[g * segs + segs - 1];
                printf("   mean of all workgroups, last segment %.3f\n", m / grid);
            }
    return 0;
}
