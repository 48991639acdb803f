// Micro-benchmark (developer tool): where do the waves of a persistent grid sit?  768 workgroups of 256 threads with
// render_kernel's LDS footprint (three per compute unit); every wave records HW_ID (SIMD, CU, shader engine) and XCC_ID.
// Prints, per compute unit, the workgroups it hosts and the SIMD of each of their four waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/placement.hip -o placement.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void where(unsigned* out, int spin)
{
    extern __shared__ char lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);      // keep the whole grid resident
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (spin < 0) lds[threadIdx.x] = 1;
}

int main()
{
    const int wgs = 768;
    unsigned* d;
    if (hipMalloc(&d, wgs * 4 * 2 * 4) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void*)where, hipFuncAttributeMaxDynamicSharedMemorySize, 33 * 1024);
    hipLaunchKernelGGL(where, dim3(wgs), dim3(256), 33 * 1024, 0, d, 20000);      // 200 us
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 8);
    (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> by_cu;
    int same_order = 0;
    for (int b = 0; b < wgs; ++b) {
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        bool ordered = true;
        for (int w = 0; w < 4; ++w) ordered = ordered && (((h[(b * 4 + w) * 2] >> 4) & 3) == (unsigned)((((h[b * 8] >> 4) & 3) + w) & 3));
        same_order += ordered;
    }
    printf("compute units seen: %zu; workgroups whose waves sit on SIMDs s, s+1, s+2, s+3 (mod 4): %d of %d\n", by_cu.size(), same_order, wgs);
    int shown = 0, stride_ok = 0, distinct0 = 0;
    for (auto& kv : by_cu) {
        const auto& v = kv.second;
        bool stride = v.size() == 3 && v[1] - v[0] == 256 && v[2] - v[1] == 256;
        stride_ok += stride;
        unsigned seen = 0;
        for (int b : v) seen |= 1u << ((h[b * 8] >> 4) & 3);
        distinct0 += __builtin_popcount(seen) == (int)v.size();
        if (shown++ < 12) {
            printf("xcc %u se %u cu %2u :", kv.first >> 12, (kv.first >> 8) & 0xf, kv.first & 0xf);
            for (int b : v) {
                printf("  wg %3d simds", b);
                for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
            }
            printf("\n");
        }
    }
    printf("compute units hosting workgroups b, b+256, b+512: %d; with wave 0 of its workgroups on distinct SIMDs: %d\n", stride_ok, distinct0);
    return 0;
}
