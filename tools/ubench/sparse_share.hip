// Micro-benchmark (developer tool): how much slower does a SPARSE wave march when it shares its SIMD with other sparse waves?
// One workgroup of W waves on one compute unit (W = 1, 4, 8, 12: at most one, one, two, three waves per SIMD); every wave
// carries `live` copies of one 352-iteration ray of the Sphere frame (pixel 771, 509 at 1920x1080) and marches it REP times with
// the production loop (march_one<SceneSphere, StratStandard>: exact pow behind its guard, tables in LDS).
// Prints microseconds per iteration of wave 0.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-cgp-select2branch \
//        -I raymarch_algo_compare_amd/csrc tools/ubench/sparse_share.hip -o sparse_share.exe
#include "rm_kernels.h"
#include <cstdio>
using namespace rm;

template <class Scene>
__global__ __launch_bounds__(768) void k(MarchCfg cfg, vec3 o, vec3 d, int live, int rep, double* out, long long* ticks)
{
    rm_load_tables<Scene>();
    __syncthreads();
    const int lane = lane_id();
    double acc = 0.0;
    int iters = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    if (lane < live) {
        for (int r = 0; r < rep; ++r) {
            const Result res = march_one<Scene, StratStandard>(o, v3(d.x + acc * 1e-300, d.y, d.z), cfg);
            acc += res.t;
            iters += res.iters;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = iters; }
}

template <class Scene>
void run(const char* name, vec3 o, vec3 d)
{
    MarchCfg cfg;
    cfg.hit_threshold = 1e-4; cfg.max_distance = 100.0; cfg.lipschitz = 1.0; cfg.max_iterations = 512; cfg.full = 0;
    cfg.prm = default_strat_params();
    double* out; long long* ticks;
    if (hipMalloc(&out, 8 * 768) != hipSuccess || hipMalloc(&ticks, 16) != hipSuccess) return;
    const double n = 1.0 / __builtin_sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    d = v3(d.x * n, d.y * n, d.z * n);
    for (int pass = 0; pass < 2; ++pass)
        for (int live : { 1, 16, 64 })
            for (int waves : { 1, 4, 8, 12 }) {
                hipLaunchKernelGGL(k<Scene>, dim3(1), dim3(64 * waves), 0, 0, cfg, o, d, live, 20, out, ticks);
                (void)hipDeviceSynchronize();
                long long h[2];
                (void)hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
                if (pass == 1)
                    printf("%-14s live lanes %2d, %2d waves on the CU: %.3f us per iteration (%lld iterations per ray)\n", name, live, waves,
                           h[0] / 100.0 / (double)h[1], h[1] / 20);
            }
}

// The frame in small: every wave first marches with all 64 lanes (`dense` rays one after the other: the dense phase), then only
// its lane 0 goes on (the tail); the pace of wave 0's tail is recorded ray by ray (a ray = 352 iterations ~ 0.12 ms).
template <class Scene>
__global__ __launch_bounds__(768) void phases(MarchCfg cfg, vec3 o, vec3 d, int dense, int segs, int filler, double* out, float* pace)
{
    __shared__ unsigned int s_done;
    if (threadIdx.x == 0) s_done = 0u;
    rm_load_tables<Scene>();
    __syncthreads();
    const int lane = lane_id();
    double acc = 0.0;
    for (int r = 0; r < dense; ++r) acc += march_one<Scene, StratStandard>(o, v3(d.x + acc * 1e-300, d.y, d.z), cfg).t;
    if (filler && threadIdx.x >= 64 * filler) {
        // filler waves: instead of idling through the tail they keep all 64 lanes of their SIMD's fp64 pipe busy until wave 0 is done
        double x0 = acc + lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
            for (int i = 0; i < 64; ++i) {
                x0 = __builtin_fma(x0, 1.0000001, 1e-9); x1 = __builtin_fma(x1, 1.0000001, 1e-9);
                x2 = __builtin_fma(x2, 1.0000001, 1e-9); x3 = __builtin_fma(x3, 1.0000001, 1e-9);
            }
        }
        acc += x0 + x1 + x2 + x3;
    } else if (lane == 0) {
        for (int sgm = 0; sgm < segs; ++sgm) {
            const long long t0 = __builtin_amdgcn_s_memrealtime();
            const Result res = march_one<Scene, StratStandard>(o, v3(d.x + acc * 1e-300, d.y, d.z), cfg);
            const long long t1 = __builtin_amdgcn_s_memrealtime();
            acc += res.t;
            if (threadIdx.x == 0 && blockIdx.x == 0) pace[sgm] = (float)((t1 - t0) / 100.0 / res.iters);
        }
        if (threadIdx.x == 0) __hip_atomic_store(&s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    out[threadIdx.x] = acc;
}

template <class Scene>
void run_phases(vec3 o, vec3 d)
{
    MarchCfg cfg;
    cfg.hit_threshold = 1e-4; cfg.max_distance = 100.0; cfg.lipschitz = 1.0; cfg.max_iterations = 512; cfg.full = 0;
    cfg.prm = default_strat_params();
    double* out; float* pace;
    const int segs = SceneIterative<Scene>::value ? 6 : 24;
    if (hipMalloc(&out, 8 * 768) != hipSuccess || hipMalloc(&pace, 4 * 24) != hipSuccess) return;
    const double n = 1.0 / __builtin_sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    d = v3(d.x * n, d.y * n, d.z * n);
    const bool bulb = SceneIterative<Scene>::value;
    for (int pass = 0; pass < 2; ++pass)
        for (int grid : { 1, 256 })
            for (int filler : { 0, 4, 1 })
            for (int dense : { 0, bulb ? 1 : 2, bulb ? 2 : 20 }) {
                hipLaunchKernelGGL(phases<Scene>, dim3(grid), dim3(768), 0, 0, cfg, o, d, dense, segs, filler, out, pace);
                (void)hipDeviceSynchronize();
                float h[24];
                (void)hipMemcpy(h, pace, 4 * segs, hipMemcpyDeviceToHost);
                if (pass == 1) {
                    printf("%3d workgroups x 12 waves, %2d dense rays first (%.1f ms), then lane 0 of %s (fp64 filler in the others: %s), us per iteration ray by ray:",
                           grid, dense, dense * (SceneIterative<Scene>::value ? 512 * 3.0e-3 : 352 * 0.334e-3), filler == 1 ? "wave 0" : filler == 4 ? "waves 0-3" : "every wave", filler ? "yes" : "no");
                    for (int i = 0; i < segs; ++i) printf(" %.2f", h[i]);
                    printf("\n");
                }
            }
}

int main(int argc, char** argv)
{
    if (argc > 1) {      // any argument: the Mandelbulb's 512-iteration ray of pixel (240, 128) at 480x270 instead (one wave: whole evaluations)
        run_phases<SceneMandelbulb>(v3(0.0, 0.0, 3.0), v3(0.0021383343303320534, 0.02779834629431532, -1.0));
        return 0;
    }
    run_phases<SceneSphere>(v3(0.0, 0.0, 5.0), v3(-0.20153801063378607, 0.032609598537562186, -1.0));
    run<SceneSphere>("Sphere", v3(0.0, 0.0, 5.0), v3(-0.20153801063378607, 0.032609598537562186, -1.0));
    return 0;
}
