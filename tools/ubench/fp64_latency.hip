// Micro-benchmark (developer tool): issue vs dependent-chain cost of fp64 VALU instructions on gfx950,
// one wavefront on one SIMD.  Prints cycles per instruction (s_memtime) for chains of 1, 2, 3, 4
// independent accumulators.  build: hipcc --offload-arch=gfx950 -O3 fp64_latency.hip -o fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int OP>
__global__ void k(double* out, double a, double b, int n, long long* cyc)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) x[c] = __builtin_fma(x[c], a, b);
                else if (OP == 1) x[c] = x[c] * a;
                else if (OP == 2) x[c] = x[c] + b;
                else { float f = (float)x[c]; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"((float)a), "v"((float)b)); x[c] = f; }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, int OP>
void run(const char* name)
{
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int n = 4096;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CHAINS, OP>), dim3(1), dim3(64), 0, 0, out, 1.0000001, 1e-9, n, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s chains=%d: %.2f memtime-ticks per instruction (%.2f per chain step)\n", name, CHAINS, (double)h / (n * 16.0 * CHAINS), (double)h / (n * 16.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, 0>("v_fma_f64"); run<2, 0>("v_fma_f64"); run<3, 0>("v_fma_f64"); run<4, 0>("v_fma_f64"); run<8, 0>("v_fma_f64");
    run<1, 1>("v_mul_f64"); run<2, 1>("v_mul_f64"); run<4, 1>("v_mul_f64");
    run<1, 2>("v_add_f64"); run<2, 2>("v_add_f64"); run<4, 2>("v_add_f64");
    return 0;
}
